"""Rasterizer forward micro-benchmark (development tool; bench.py is the contract benchmark)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

from dgs_amd import synth
from dgs_amd.raster import default_backend


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--regime", default="trained")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    be = default_backend()
    sc = synth.gaussian_scene(a.res, regime=a.regime, seed=0)
    cams, _, _ = synth.render_cameras(a.res, a.views, phase_deg=10)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
    xyz, shs, sca, rot, op = (t(sc[k]) for k in ("xyz", "shs", "scales", "rotations", "opacities"))
    vm = t(np.stack([c["viewmatrix"] for c in cams])); pm = t(np.stack([c["projmatrix"] for c in cams]))
    cp = t(np.stack([c["campos"] for c in cams])); bg = t(np.ones(3, np.float32))

    def run(cap=0):
        return be.forward_views(bg, xyz[None], None, op, sca, rot, 1.0, None, vm, pm, cp, None, cams[0]["tanfovx"],
                                cams[0]["tanfovy"], a.res, a.res, shs, 0, False, False, views_per_set=a.views,
                                binning_capacity=cap)

    n = run()[0]
    P = xyz.shape[0]
    print(f"res {a.res} views {a.views} regime {a.regime}: P={P} N_total={n} N/P/view={n / P / a.views:.2f}")
    for label, cap in (("sync (reference-style num_rendered readback)", 0), ("async (preallocated binning)", int(n * 1.2))):
        for _ in range(3):
            run(cap)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(a.iters):
            run(cap)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.iters * 1e3
        print(f"  {label}: {e0.elapsed_time(e1) / a.iters:.3f} ms/call (gpu events), {wall:.3f} ms wall "
              f"-> {a.views / (e0.elapsed_time(e1) / a.iters) * 1e3:.0f} views/s")


def bench_backward(a):
    """forward + backward through the batched autograd entry (raw parameters), views/s."""
    from dgs_amd import cameras
    from dgs_amd.raster import render_views_autograd
    dev = torch.device("cuda:0")
    be = default_backend()
    sc = synth.gaussian_scene(a.res, regime=a.regime, seed=0, activated=False)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
    leaves = [t(sc[k])[None].requires_grad_(True) for k in ("xyz", "shs", "scales", "rotations", "opacities")]
    c2w = t(cameras.ring_cameras(a.views, phase_deg=10))[None]
    k = t(cameras.default_fxfycxcy(a.res)).expand(1, a.views, 4).contiguous()
    w = torch.randn(1, a.views, 3, a.res, a.res, device=dev) / (3 * a.res * a.res)

    def step():
        for x in leaves:
            x.grad = None
        img = render_views_autograd(be, *leaves, a.res, a.res, c2w, k)
        img.backward(w)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    print(f"  forward+backward (batched autograd entry): {ms:.3f} ms/call -> {a.views / ms * 1e3:.0f} views/s")


if __name__ == "__main__":
    main()
    import argparse as _ap
    _p = _ap.ArgumentParser(); _p.add_argument("--res", type=int, default=256); _p.add_argument("--views", type=int, default=4)
    _p.add_argument("--regime", default="trained"); _p.add_argument("--iters", type=int, default=20)
    bench_backward(_p.parse_args())
