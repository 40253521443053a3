"""Training-step timing at the reference's obj stage-1 shape (BASELINE.json configs[3]): batch 4 x 4 input views at 256^2,
10 rendered views per sample, DiT forward (activations saved) + batched rasterizer forward + MSE + rasterizer backward + DiT
backward.  Development tool (1 GPU; the gradient all-reduce is exercised by tests/test_parallel_gloo.py)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import numpy as np
import torch

from bench import dit_flops, synth_batch
from dgs_amd import cameras, denoiser as dn

DEV = torch.device("cuda:0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--render-views", type=int, default=10)
    ap.add_argument("--iters", type=int, default=5)
    a = ap.parse_args()
    m = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=24), device=DEV)
    m.reset_parameters(seed=0)
    batch, t = synth_batch(a.batch, 4, a.res, DEV, 0)
    rc2w = torch.tensor(np.stack([cameras.ring_cameras(a.render_views, phase_deg=7.0 * b) for b in range(a.batch)])).to(DEV)
    rk = torch.tensor(cameras.default_fxfycxcy(a.res)).expand(a.batch, a.render_views, 4).contiguous().to(DEV)
    target = torch.rand(a.batch, a.render_views, 3, a.res, a.res, device=DEV)
    eng = m.engine()
    L = eng.num_tokens(4, a.res, a.res)
    ev = lambda: torch.cuda.Event(enable_timing=True)

    def step(record=None):
        e = [ev() for _ in range(5)] if record is not None else None
        if e: e[0].record()
        out, _ = eng.forward_train(batch["image"], batch["ray_o"], batch["ray_d"], t)
        if e: e[1].record()
        leaves = [out[k].requires_grad_(True) for k in ("xyz", "features", "scaling", "rotation", "opacity")]
        from dgs_amd.raster import default_backend, render_views_autograd
        img = render_views_autograd(default_backend(), *leaves, a.res, a.res, rc2w, rk)
        loss = ((img - target) ** 2).mean()
        if e: e[2].record()
        loss.backward()
        if e: e[3].record()
        eng.backward(*(x.grad for x in leaves))
        if e: e[4].record()
        if record is not None:
            record.append(e)
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    rec = []
    t0 = ev(); t1 = ev()
    t0.record()
    for _ in range(a.iters):
        step(rec)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / a.iters
    parts = np.mean([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in rec], axis=0)
    fl = 3 * dit_flops(L) * a.batch
    print(f"train step B={a.batch} res={a.res} L={L} render_views={a.render_views}: {ms:.2f} ms/step")
    print(f"  DiT forward(train) {parts[0]:.2f} ms | raster fwd + loss {parts[1]:.2f} ms | raster bwd {parts[2]:.2f} ms | DiT backward {parts[3]:.2f} ms")
    print(f"  DiT fwd+bwd = {parts[0] + parts[3]:.2f} ms -> {fl / ((parts[0] + parts[3]) * 1e-3) / 1e12:.1f} TFLOP/s (3 x fwd FLOPs, no recompute)")
    print(f"  samples/s = {a.batch / ms * 1e3:.2f}, rendered views/s = {a.batch * a.render_views / ms * 1e3:.1f}")
    print(f"  saved activations {eng._train['saved'].numel() / 2**30:.2f} GiB, backward workspace {eng._train['bws'].numel() / 2**30:.2f} GiB, "
          f"gradient buffer {eng._train['fg'].flat.numel() * 4 / 2**30:.2f} GiB")


if __name__ == "__main__":
    main()
