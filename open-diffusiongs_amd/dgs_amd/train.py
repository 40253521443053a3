"""One data-parallel training step of the hot path (what PointDiffusionSystem.training_step does around the model,
systems/diffusion_gs_system.py:71-128, minus the parts that are out of scope: noise schedule, LPIPS, logging):

    gaussians = model.image_to_gaussians(noisy views)          DiT forward, activations saved        (HIP)
    renders   = model.render_gaussians(gaussians, cameras)     all b*v views, one launch sequence    (HIP)
    loss      = mean((renders - target)^2)                     the reference's lambda_mse term       (HIP: dgs_amd.losses, one pass)
    loss.backward()                                            rasterizer backward + DiT backward    (HIP)
    gradient all-reduce over the ranks                         RCCL over xGMI, a few large buckets   (dgs_amd.parallel)
    optimizer step on the fp32 master parameters               torch.optim (AdamW in the reference configs)

Lightning DDP in the reference (`strategy: ddp_find_unused_parameters_true`) reduces ~74 buckets of 25 MB; here the
gradients already live in one flat buffer in backward-completion order, so the exchange is a handful of large
collectives that start while earlier blocks are still in backward.
"""
import torch

from . import losses
from .parallel import BucketedAllReduce


class DataParallelTrainer:
    def __init__(self, model, optimizer, bucket_bytes=256 << 20):
        self.model, self.opt = model, optimizer
        self.bucket_bytes = bucket_bytes
        self._reducer = None

    def _reduce_gradients(self):
        """All-reduce (mean) of the engine's flat gradient buffer; parameter .grad tensors are then refreshed from it."""
        eng = self.model.engine()
        fg = eng._train["fg"]
        if self._reducer is None or self._reducer.flat.data_ptr() != fg.flat.data_ptr():
            self._reducer = BucketedAllReduce(fg.flat, self.bucket_bytes)
        if self._reducer.world > 1:
            self._reducer.finish()
            views = eng.grad_views()
            with torch.no_grad():
                for n, p in self.model.named_parameters():
                    p.grad.copy_(views[n].reshape(p.shape))

    def step(self, batch, t, target, render_c2w=None, render_fxfycxcy=None):
        """batch: dict(image, ray_o, ray_d, c2w, fxfycxcy) like the reference's input_batch; target [b, v, 3, H, W]."""
        m = self.model
        self.opt.zero_grad(set_to_none=True)
        params, _ = m.image_to_gaussians(batch["image"], batch["ray_o"], batch["ray_d"], t)
        c2w = batch["c2w"] if render_c2w is None else render_c2w
        k = batch["fxfycxcy"] if render_fxfycxcy is None else render_fxfycxcy
        H, W = batch["image"].shape[3], batch["image"].shape[4]
        rendered = m.render_gaussians(params, c2w, k, H, W)
        loss, _l2, self.last_psnr = losses.mse_psnr(rendered, target.to(rendered.dtype), lib=getattr(m, "_lib", None))
        loss.backward()
        self._reduce_gradients()
        self.opt.step()
        return loss.detach()
