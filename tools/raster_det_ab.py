"""Rasterizer backward: the deterministic form (per-instance slots + ordered gather, dgs_raster.h `scratch`) against the atomic form, in
ONE process on one box: forward + backward time of 4 views at 256^2 (and 512^2 trained-like) in both regimes, alternating, plus
bit-reproducibility of the deterministic form (two runs: identical gradients) and its distance from the atomic form's result.

    python tools/raster_det_ab.py  > gpurun_out/raster_det_ab.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))

import numpy as np
import torch

from dgs_amd import cameras, synth
from dgs_amd.raster import RasterBackend, render_views_autograd


def case(res, regime, views, iters=10):
    dev = torch.device("cuda:0")
    sc = synth.gaussian_scene(res, regime=regime, seed=0, activated=False)
    t = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
    raw = [t(sc[k])[None] for k in ("xyz", "shs", "scales", "rotations", "opacities")]
    c2w = t(cameras.ring_cameras(views, phase_deg=10))[None]
    k = t(cameras.default_fxfycxcy(res)).expand(1, views, 4).contiguous()
    w = torch.randn(1, views, 3, res, res, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) / (3 * res * res)
    bes = {"atomic": RasterBackend(), "deterministic": RasterBackend()}
    bes["atomic"].deterministic = False
    bes["deterministic"].deterministic = True

    def step(be):
        leaves = [x.clone().requires_grad_(True) for x in raw]
        render_views_autograd(be, *leaves, res, res, c2w, k).backward(w)
        return [x.grad for x in leaves]

    grads = {n: [step(be) for _ in range(3)] for n, be in bes.items()}      # also the warm-up (call 1 of a shape is synchronous)
    assert bes["deterministic"].last_backward_deterministic and not bes["atomic"].last_backward_deterministic
    times = {n: [] for n in bes}
    for rnd in range(3):                                                    # alternate: clocks drift within a call
        for n, be in bes.items():
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                step(be)
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / iters)
    names = ("xyz", "features", "scaling", "rotation", "opacity")
    same = {n: all(torch.equal(a, b) for a, b in zip(grads[n][1], grads[n][2])) for n in grads}
    dist = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(grads["deterministic"][2], grads["atomic"][2]))
    P = raw[0].shape[1]
    print(f"{res}^2, {views} views, {regime}: P = {P}")
    for n in bes:
        print(f"  {n:14s} forward + backward {min(times[n]):.3f} ms (rounds: {', '.join('%.3f' % x for x in times[n])})   two runs bit-identical: {same[n]}")
    print(f"  deterministic vs atomic, max |diff| / max |gradient| over {names}: {dist:.2e}")


if __name__ == "__main__":
    case(256, "trained", 4)
    case(256, "init", 4)
    case(512, "trained", 4, iters=5)
