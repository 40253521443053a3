"""A/B of two builds of the library on dgs_dit_layernorm_backward at the training shape (4 samples x 4224 rows x 1024, bf16 dh, fp32
residual gradient in and out): the outputs side by side, then alternating timed calls (HIP events).  Algorithmic bytes per call:
rows x width x (4 x + 2 dh + 4 dx_in + 4 dx_out).          python tools/ln_bwd_ab.py <base.so> [<new.so>]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import _p, _stream

DEV = "cuda:0"
libs = [_native.open_library(os.path.abspath(sys.argv[1])), _native.open_library(os.path.abspath(sys.argv[2])) if len(sys.argv) > 2 else _native.open_library(_native.LIB_PATH)]
names = ["base", "new"]
g = torch.Generator(device=DEV).manual_seed(0)
for B, rows in ((4, 4224), (1, 4224), (4, 4352)):
    W = 1024
    M = B * rows
    x = torch.randn(M, W, generator=g, device=DEV) * 2 + 0.5
    dh = torch.randn(M, W, generator=g, device=DEV).to(torch.bfloat16)
    w = 1 + 0.2 * torch.randn(W, generator=g, device=DEV)
    mod = torch.randn(B, 3 * W, generator=g, device=DEV)
    dx_in = torch.randn(M, W, generator=g, device=DEV)
    calls, outs = [], []
    for lib in libs:
        dx, dmod, dw = torch.zeros(M, W, device=DEV), torch.zeros(B, 3 * W, device=DEV), torch.zeros(W, device=DEV)
        nb = lib.dgs_dit_layernorm_backward_scratch_bytes(M, W, rows)
        scratch = torch.zeros(nb // 4, device=DEV)
        a = _native.DgsDitLayerNormBackwardArgs()
        a.rows, a.width, a.x, a.dh, a.dh_f32, a.weight, a.scale, a.mod_stride, a.rows_per_batch, a.eps = M, W, _p(x), _p(dh), 0, _p(w), _p(mod[:, W:]), 3 * W, rows, 1e-6
        a.dx_in, a.dx_out, a.dshift, a.dscale, a.dweight, a.scratch, a.scratch_bytes = _p(dx_in), _p(dx), _p(dmod), _p(dmod[:, W:]), _p(dw), _p(scratch), nb
        st = _stream(x.device)
        f = (lambda lib=lib, a=a, st=st: lib.dgs_dit_layernorm_backward(ctypes.byref(a), st))
        assert f() == 0
        calls.append(f); outs.append((dx, dmod, dw, scratch, a))
    torch.cuda.synchronize()
    rel = lambda u, v: float((u - v).norm() / v.norm())
    print(f"B={B} rows={rows}: slab rows base {outs[0][3].numel() // (3 * W)} new {outs[1][3].numel() // (3 * W)}; dx identical "
          f"{torch.equal(outs[0][0], outs[1][0])}; column sums rel diff shift|scale {rel(outs[1][1], outs[0][1]):.2e} weight {rel(outs[1][2], outs[0][2]):.2e}", flush=True)
    n = 30
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)] for _ in libs]
    for i in range(n):
        for k, f in enumerate(calls):
            e0, e1 = ev[k][i]
            e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    med = [sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in v)[n // 2] for v in ev]
    nbytes = M * W * 14.0
    print(f"timing B={B} rows={rows}: " + "  ".join(f"{nm} {m:.1f} us ({nbytes / m / 1e6:.2f} TB/s)" for nm, m in zip(names, med)) + f"  ratio {med[1] / med[0]:.3f}  (incl. the column reduce launch)", flush=True)
