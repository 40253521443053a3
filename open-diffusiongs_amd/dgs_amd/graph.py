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


_ROW_POOL = []          # page-locked int32[4] rows, allocated in chunks and NEVER freed (a graph may still store into its rows while it is
                        # being collected; freeing page-locked memory is a device-wide synchronisation)


def _pinned_rows(n):
    """n page-locked statistics rows for one capture, from a process-wide pool."""
    while len(_ROW_POOL) < n:
        chunk = torch.zeros(64, 4, dtype=torch.int32).pin_memory()
        _ROW_POOL.extend(chunk[i] for i in range(64))
    rows, _ROW_POOL[:] = _ROW_POOL[:n], _ROW_POOL[n:]
    for r in rows:
        r.zero_()
    return rows


class GraphedForward:
    MAX_RASTER_CALLS = 8          # pinned statistics rows handed to the rasterizer calls of one captured step

    def __init__(self, model, input_batch, timesteps, warmup=2):
        dev = input_batch["image"].device
        if dev.type != "cuda":
            raise RuntimeError("GraphedForward needs a GPU stream to capture (the CPU emulation build has none)")
        self.model = model
        self.static = {k: input_batch[k].detach().clone() for k in _INPUT_KEYS}
        self.t = timesteps.detach().clone()
        self.key = self.shape_key(input_batch, timesteps)
        self._backend = model.gs_renderer.backend()
        self.replays = self.recaptures = self.healed = 0
        self._capture(warmup)

    def _capture(self, warmup):
        """Warm-up + capture.  The rasterizer calls of the captured step report into pinned rows of THIS graph (allocated here, outside
        the capture), and the backend tells which plans they ran from (`_watch`): a plan at risk -- binning capacity below the worst
        case of its shape (dgs_amd/raster.py `_AsyncPlan`) -- makes every replay a verified one."""
        dev, model, backend = self.t.device, self.model, self._backend
        # warm-up on a side stream: the first render of a shape synchronises (it learns the binning capacity), workspaces are
        # allocated, per-kernel attributes are set -- none of which may happen inside the capture
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):
                model(self.static, self.t)
        torch.cuda.current_stream(dev).wait_stream(side)
        backend.check_async(wait=True)                 # the warm-up renders' statistics: capacity + ordering form of the capture
        if getattr(self, "_rows", None):
            _ROW_POOL.extend(self._rows)               # a re-capture: the graph that stored into them is replaced below
        self._rows = _pinned_rows(self.MAX_RASTER_CALLS)
        backend.begin_capture_log(self._rows)
        self.graph = torch.cuda.CUDAGraph()
        # No cyclic garbage collection while the stream is capturing: a finalizer that frees page-locked memory or destroys a graph of
        # an object that died earlier is not a capturable operation (seen as a bare abort() in the GPU suite when graph tests ran behind
        # tests that leave trainers and backends for the collector).  torch.cuda.graph() collects once on entry; this keeps it at that.
        import gc
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            # thread_local: other threads of the process (RCCL's watchdog polls its events) must not invalidate the capture
            with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.rendered, self.gaussians = model(self.static, self.t)
        finally:
            if gc_was_on:
                gc.enable()
            self._watch = backend.end_capture_log()    # [(plan, pinned row, capacity, (P, W, H, V))] of the captured rasterizer calls
        if not self._watch:
            raise RuntimeError("GraphedForward: the captured step made no planned rasterizer call -- nothing to verify replays against")
        self._verify = any(plan.at_risk(cap) for plan, _row, cap, _shape in self._watch)
        self._event = None

    def __del__(self):
        rows = getattr(self, "_rows", None)
        if rows and _ROW_POOL is not None:
            _ROW_POOL.extend(rows)                     # back to the pool (never freed)

    @staticmethod
    def shape_key(input_batch, timesteps):
        return tuple((k, tuple(input_batch[k].shape), input_batch[k].dtype) for k in _INPUT_KEYS) + (tuple(timesteps.shape), timesteps.dtype)

    def _failed(self):
        """The latest replay's statistics rows (valid once its event has passed): raises for a device-side failure other than an
        outgrown buffer, returns True if a call outgrew its buffer (after telling the plans what the scene needs)."""
        from . import _native
        over = False
        for plan, row, cap, shape in self._watch:
            n, status, longest = int(row[0]) & 0xFFFFFFFF, int(row[1]), int(row[2]) & 0xFFFFFFFF
            if status == _native.DGS_ERR_BINNING_OVERFLOW:
                plan.note(self._backend.lib, n, longest, *shape)
                over = True
            elif status != 0:
                raise RuntimeError(f"GraphedForward: a replay failed on the device ({_native.status_string(self._backend.lib, status)})")
        return over

    def check(self, wait=True):
        """Unverified graphs (no plan at risk: their buffers hold the worst case, a replay cannot outgrow them): raise if the latest
        replay reported any other device-side failure.  Verified graphs have nothing pending -- `replay` looked already."""
        if self._event is None:
            return
        if wait:
            self._event.synchronize()
        elif not self._event.query():
            return
        self._event = None
        if self._failed():
            raise RuntimeError("GraphedForward: a replay outgrew a binning buffer that was sized for the worst case of its shape")

    def replay(self):
        """One step.  With a plan at risk the host waits for the replay and looks at what its rasterizer calls reported; a scene
        that outgrew the captured capacity is healed here -- the graph is captured again with buffers sized for it (the warm-up
        of the capture renders the scene eagerly, which raises the plan's capacity) and replayed -- so the tensors this returns
        are always a complete render, as the reference's are (it resizes on every call: rasterize_points.cu:27-33)."""
        for _ in range(4):
            self.graph.replay()
            self.replays += 1
            self._event = torch.cuda.Event()
            self._event.record()
            if not self._verify:
                return self.rendered, self.gaussians
            self._event.synchronize()
            self._event = None
            if not self._failed():
                return self.rendered, self.gaussians
            self.healed += 1
            self.recaptures += 1
            self._capture(warmup=1)
        raise RuntimeError("GraphedForward: a step kept outgrowing its binning buffers")

    def __call__(self, input_batch, timesteps):
        for k, dst in self.static.items():
            src = input_batch[k]
            if src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        if timesteps.data_ptr() != self.t.data_ptr():
            self.t.copy_(timesteps)
        return self.replay()
