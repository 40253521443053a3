"""One data-parallel training step of the hot path (what PointDiffusionSystem.training_step does around the model,
systems/diffusion_gs_system.py:71-128, minus the parts that are out of scope: noise schedule, LPIPS, logging):

    gaussians = model.image_to_gaussians(noisy views)          DiT forward, activations saved / per-block recompute   (HIP)
    renders   = model.render_gaussians(gaussians, cameras)     all b*v views, one launch sequence                     (HIP)
    loss      = mean((renders - target)^2)                     the reference's lambda_mse term                        (HIP: dgs_amd.losses)
    loss.backward()                                            rasterizer backward + DiT backward                     (HIP)
      `- per finished gradient group: all-reduce of the buckets it completes, on RCCL's stream, WHILE the backward of the
         earlier blocks is still being enqueued / executed                                                           (dgs_amd.parallel)
    optimizer step on the fp32 master parameters               torch.optim (AdamW in the reference configs)
    refresh of the engine's bf16 / transposed weight copies    in place

Lightning DDP in the reference (`strategy: ddp_find_unused_parameters_true`, configs/diffusionGS_rel.yaml:80;
scripts/train_obj_stage1.sh:5-7) reduces ~74 buckets of 25 MB behind autograd hooks.  Here dgs_dit_backward writes every
gradient into ONE flat fp32 buffer in backward-completion order and calls back after each block (DgsDitBackwardArgs.block_done);
the callback launches every bucket that just became final.  The parameters' .grad tensors ARE views of that buffer: the
collective reduces them in place and nothing is copied in or out.  Averaging is folded into the loss scale.
"""
import torch

from . import losses
from .parallel import BucketedAllReduce, GradNorm, broadcast_parameters


class DataParallelTrainer:
    """Owns the model's training hooks while it lives: `.grad` of every parameter is a view of the engine's flat gradient buffer,
    the backward writes there in place and reports finished groups to the bucketed all-reduce.  `close()` (or leaving the `with`
    block) hands the model back: a later plain `loss.backward()` then returns gradients through autograd again."""

    def __init__(self, model, optimizer, bucket_bytes=None, accumulate_grad_batches=1, group=None, compress=None, force_collectives=False,
                 max_grad_norm=None, broadcast_from=0, deterministic=None):
        """deterministic (default on; DGS_RASTER_DETERMINISTIC=0 or False turns it off): the rasterizer backward without floating-point
        atomics (dgs_raster.h `scratch`), which makes the WHOLE step bit-reproducible -- the DiT backward, the bucketed all-reduce
        order, the norm and the AdamW launch already are -- for +0.10 of 1.07 ms of rasterizer backward at 4 views of 256^2
        (profiles/r04_raster_deterministic_ab.txt), ~0.1 % of the step, in trained-like scenes, and +1.6 % of the step (90.9 -> 92.4 ms,
        profiles/r05_train_det_alloc.txt) in the densest case the bench has: 4 x 10 views of random-init Gaussians, 634 M instances,
        45 GB of scratch (36 bytes per instance slot; past `backend.deterministic_budget` = 64 GiB the atomic form runs and
        `backend.last_backward_deterministic` says so).  Two runs from the same state then hold identical parameters,
        and so do the ranks of a data-parallel job after every step (tests/test_optim.py, tests/test_parallel_gloo.py).
        max_grad_norm: global-norm gradient clip (Lightning `gradient_clip_val`; the reference trains with 0.5,
        configs/diffusionGS_rel.yaml:76-77) -- the norm's partial sums are taken bucket by bucket behind each bucket's all-reduce and
        the scale is applied inside the optimizer launch (FusedAdamW) or as one in-place scale of the flat buffer (any other optimizer).
        broadcast_from: the rank whose parameters every rank starts from (DDP's init-time broadcast); None skips it."""
        self.model, self.opt = model, optimizer
        import os
        if deterministic is None:
            deterministic = os.environ.get("DGS_RASTER_DETERMINISTIC", "1") not in ("", "0")
        self.deterministic = bool(deterministic)
        renderer = getattr(model, "gs_renderer", None)
        self._renderer, self._private_backend = renderer, False
        self._raster_backend = renderer.backend() if renderer is not None and hasattr(renderer, "backend") else None
        if renderer is not None and getattr(renderer, "_backend", "absent") is None and self._raster_backend is not None:
            # the model renders through the process-wide default backend: the trainer's choice (deterministic backward) must not leak to
            # its other users, so the model gets a backend of its own for as long as the trainer lives (same library, same exponential;
            # its plans learn their shapes again on the first step)
            from .raster import RasterBackend
            own = RasterBackend(lib=self._raster_backend.lib, exact_exp=self._raster_backend.exact_exp)
            own.deterministic_budget = self._raster_backend.deterministic_budget
            renderer._backend, self._raster_backend, self._private_backend = own, own, True
        self._raster_was_deterministic = getattr(self._raster_backend, "deterministic", None)
        self._raster_was_budget = getattr(self._raster_backend, "deterministic_budget", None)
        if self._raster_backend is not None:
            self._raster_backend.deterministic = self.deterministic
        self.accumulate = int(accumulate_grad_batches)
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm else None
        self.broadcast_bytes = 0
        if broadcast_from is not None:
            self.broadcast_bytes = broadcast_parameters(model, src=broadcast_from, group=group)
            if self.broadcast_bytes:
                model.refresh_engine_weights()
        eng = model.engine()
        self.fg = eng._train_state()["fg"]
        if next(model.parameters()).device != self.fg.flat.device:
            raise RuntimeError("DataParallelTrainer: the module's parameters must live on the engine's device (model.to(device))")
        self.norm = GradNorm(self.fg.flat, getattr(model, "_lib", None) or eng.lib) if self.max_grad_norm else None
        self.last_grad_sumsq = None      # device float[1] of the last step (its square root is the norm BEFORE clipping, as clip_grad_norm_ returns)
        self.reducer = BucketedAllReduce(self.fg.flat, bucket_bytes, group=group, compress=compress, force_collectives=force_collectives,
                                         norm=self.norm)
        self.world = self.reducer.world
        self._accum = torch.zeros_like(self.fg.flat) if self.accumulate > 1 else None      # allocated lazily too when a batch needs splitting
        self._summed_to = 0
        self._use_accum = False
        self._last_micro = True
        self._layers = eng.layers
        model._grads_in_place = True
        model._block_hook = self._on_gradients_final
        self._attach_grads()
        self.last_psnr = None
        self.lead_probe = None           # a list: every block_done callback appends (stage, host time, event recorded on the compute stream)

    def close(self):
        """Give the model back: no in-place gradients, no per-block hook; the parameters keep COPIES of their last gradients."""
        m = self.model
        if getattr(m, "_block_hook", None) == self._on_gradients_final:
            m._block_hook = None
            m._grads_in_place = False
            for p in m.parameters():
                if p.grad is not None:
                    p.grad = p.grad.clone()
            if self._private_backend:
                self._renderer._backend = None                 # back to the process-wide default backend, which was never touched
            elif self._raster_backend is not None and self._raster_was_deterministic is not None:
                self._raster_backend.deterministic = self._raster_was_deterministic
                self._raster_backend.deterministic_budget = self._raster_was_budget

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def _attach_grads(self):
        """.grad of every parameter = its slice of the flat buffer (adaLN slices included): written by the backward, reduced
        in place by the collectives, read by the optimizer."""
        views = self.model.engine().grad_views()
        for n, p in self.model.named_parameters():
            v = views[n].reshape(p.shape)
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def _end_of_stage(self, stage):
        if stage < 0:
            return self.fg.flat.numel()
        if stage >= self._layers:
            return self.fg.end_of("head_ada_b")
        return self.fg.end_of(f"{stage}.ada_b")

    def _on_gradients_final(self, stage):
        """Host callback from inside dgs_dit_backward: flat[:end] is final once the kernels enqueued so far have run."""
        if not self._last_micro:
            return
        if self.lead_probe is not None:                          # measurement (bench.py): how far the host runs ahead of the device here
            import time
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.lead_probe.append((stage, time.perf_counter(), ev))
        end = self._end_of_stage(stage)
        if self._use_accum and end > self._summed_to:             # fold in the earlier micro-batches, slice by slice
            self.fg.flat[self._summed_to:end] += self._accum[self._summed_to:end]
            self._summed_to = end
        self.reducer.ready_up_to(end, tag=stage)

    def step(self, batch, t, target, render_c2w=None, render_fxfycxcy=None):
        """batch: dict(image, ray_o, ray_d, c2w, fxfycxcy) like the reference's input_batch (leading dim = accumulate *
        per-micro-batch size when accumulate_grad_batches > 1); target [b, v, 3, H, W].  Returns the (local) mean loss."""
        m, K = self.model, self.accumulate
        if m._block_hook != self._on_gradients_final:
            raise RuntimeError("DataParallelTrainer.step after close() (or another trainer took the model over)")
        self._attach_grads()
        c2w = batch["c2w"] if render_c2w is None else render_c2w
        k = batch["fxfycxcy"] if render_fxfycxcy is None else render_fxfycxcy
        H, W = batch["image"].shape[3], batch["image"].shape[4]
        nb = batch["image"].shape[0]
        assert nb % K == 0, "batch size must be a multiple of accumulate_grad_batches"
        mb = nb // K
        limit = getattr(m, "MAX_DIFFERENTIABLE_BATCH", 4)
        if mb > limit:
            # one dgs_dit_forward_train / dgs_dit_backward call takes <= 4 samples: a larger per-rank batch (the reference's scene
            # configurations: 12 at 512^2, 24 at 256^2; configs/diffusionGS_scene_512.yaml:16) runs as micro-batches inside THIS
            # optimizer step -- same gradient (mean over the whole batch), one all-reduce, one clip, one update
            c = -(-mb // limit)
            while mb % c:
                c += 1
            K, mb = K * c, mb // c
            if self._accum is None:
                self._accum = torch.zeros_like(self.fg.flat)
        total = 0.0
        self._summed_to = 0
        self._use_accum = K > 1
        for j in range(K):
            sl = slice(j * mb, (j + 1) * mb)
            self._last_micro = j == K - 1
            params, _ = m.image_to_gaussians(batch["image"][sl], batch["ray_o"][sl], batch["ray_d"][sl], t[sl])
            rendered = m.render_gaussians(params, c2w[sl], k[sl], H, W)
            loss, _l2, self.last_psnr = losses.mse_psnr(rendered, target[sl].to(rendered.dtype), lib=getattr(m, "_lib", None))
            (loss * (1.0 / (K * self.world))).backward()       # mean over micro-batches and ranks folded into the scale
            total = total + loss.detach()
            if not self._last_micro:
                if j == 0:
                    self._accum.copy_(self.fg.flat)
                else:
                    self._accum += self.fg.flat
        self.reducer.finish(average=False)
        if self.norm is not None:
            self.last_grad_sumsq = self.norm.total()
            if getattr(self.opt, "refreshes_engine", False):          # FusedAdamW: the scale rides in the update launch
                self.opt.step(grad_sumsq=self.last_grad_sumsq, max_grad_norm=self.max_grad_norm)
            else:                                                      # torch.nn.utils.clip_grad_norm_ on the flat buffer (the .grad views)
                # a non-finite norm (overflow / NaN upstream): the update is skipped, as FusedAdamW's launch and the reference's GradScaler
                # do -- a NaN coefficient multiplied into the gradients would corrupt every parameter (one host read: this branch is not
                # the no-sync path)
                if bool(torch.isfinite(self.last_grad_sumsq).all()):
                    coef = (self.max_grad_norm / (self.last_grad_sumsq.sqrt() + 1e-6)).clamp(max=1.0)
                    self.fg.flat.mul_(coef)
                    self.opt.step()
        else:
            self.opt.step()
        # weights changed: the engine's bf16 / transposed copies have to follow.  dgs_amd.optim.FusedAdamW writes them in the same launch
        # as the update; for any other optimizer they are refreshed EXPLICITLY -- DGSDenoiser.engine() only notices parameter version
        # counters, and an optimizer is free not to move them (foreach / fused implementations, `p.data` updates, EMA swaps)
        if getattr(self.opt, "refreshes_engine", False):
            m.engine()
        else:
            m.refresh_engine_weights()
        return total / K
