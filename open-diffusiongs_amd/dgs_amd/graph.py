"""`DGSDenoiser.forward` at fixed shapes as ONE hipGraph (the sampling loop calls it 30 times with the same shapes,
gaussian_diffusion.py:560-603 -> :350).

What makes the step capturable: the DiT forward is one C call with no allocation and no synchronisation (csrc/dit_forward.hip), and
the rasterizer's planned mode (dgs_amd/raster.py `_AsyncPlan`) neither reads `num_rendered` back nor allocates by callback after the
first call of a shape -- the reference's per-forward `cudaMemcpy(&num_rendered, ..., DeviceToHost)` (rasterizer_impl.cu:281) would end
any capture.  A replay is one host call (~10 us) instead of ~190 kernel launches and a few dozen tensor allocations enqueued from
Python; the device executes the same kernels in the same order (hipGraph keeps a dependency barrier between consecutive kernel nodes),
so the gain is on the host side: the step's cost no longer depends on how fast the host enqueues.

Static-buffer semantics (as with any captured graph): inputs are copied into the graph's own tensors before a replay, outputs are
the graph's own tensors and are overwritten by the next replay -- `clone()` what has to survive.
"""
import torch

_INPUT_KEYS = ("image", "ray_o", "ray_d", "c2w", "fxfycxcy")


class GraphedForward:
    def __init__(self, model, input_batch, timesteps, warmup=2):
        dev = input_batch["image"].device
        if dev.type != "cuda":
            raise RuntimeError("GraphedForward needs a GPU stream to capture (the CPU emulation build has none)")
        self.model = model
        self.static = {k: input_batch[k].detach().clone() for k in _INPUT_KEYS}
        self.t = timesteps.detach().clone()
        self.key = self.shape_key(input_batch, timesteps)
        backend = model.gs_renderer.backend()
        # warm-up on a side stream: the first render of a shape synchronises (it learns the binning capacity), workspaces are
        # allocated, per-kernel attributes are set -- none of which may happen inside the capture
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):
                model(self.static, self.t)
        torch.cuda.current_stream(dev).wait_stream(side)
        backend.check_async(wait=True)                 # the warm-up renders' statistics: capacity + ordering form of the capture
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (RCCL's watchdog polls its events) must not invalidate the capture
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.rendered, self.gaussians = model(self.static, self.t)
        B, V, _, H, W = self.static["image"].shape
        Vr = int(self.static["c2w"].shape[1])
        self._plan = backend.plan_for(model.cfg.n_gaussians + V * H * W, W, H, B * Vr, Vr, dev)
        self._stats = getattr(self._plan, "graph_stats", None)     # pinned int32[4], rewritten by every replay
        self._event = None
        self._backend = backend
        self.replays = 0

    @staticmethod
    def shape_key(input_batch, timesteps):
        return tuple((k, tuple(input_batch[k].shape), input_batch[k].dtype) for k in _INPUT_KEYS) + (tuple(timesteps.shape), timesteps.dtype)

    def check(self, wait=True):
        """Raise if the latest replay failed on the device (wait=True: block until it has finished)."""
        self._check_previous(wait=wait)

    def _check_previous(self, wait=False):
        """The previous replay's instance statistics (in pinned memory once its event has passed): a scene that outgrew the captured
        binning capacity rendered NaN -- raise, as the eager path does on its next call."""
        if self._event is None or self._stats is None:
            return
        if wait:
            self._event.synchronize()
        elif not self._event.query():
            return
        self._event = None
        n, status = int(self._stats[0]) & 0xFFFFFFFF, int(self._stats[1])
        if status != 0:
            from . import _native
            raise RuntimeError(f"GraphedForward: the previous replay failed on the device ({_native.status_string(self._backend.lib, status)}; "
                               f"{n} instances, captured capacity {self._plan.capacity}): its image is NaN.  Drop this graph "
                               f"(DGSDenoiser.drop_graphs()) and run the step eagerly once to re-learn the capacity")

    def replay(self):
        self._check_previous()
        self.graph.replay()
        self._event = torch.cuda.Event()
        self._event.record()
        self.replays += 1
        return self.rendered, self.gaussians

    def __call__(self, input_batch, timesteps):
        for k, dst in self.static.items():
            src = input_batch[k]
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        if timesteps.data_ptr() != self.t.data_ptr():
            self.t.copy_(timesteps)
        return self.replay()
