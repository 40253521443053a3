"""A/B of two builds of the library on dgs_dit_layernorm at the DiT shapes (fp32 residual stream in, bf16 GEMM operand out, adaLN
shift / scale per sample; the input LayerNorm with its weight and fp32 output): outputs bit-compared, then alternating timed calls
(HIP events around 8 back-to-back launches).  Algorithmic bytes per call: rows x width x (4 + 2).
    python tools/ln_ab.py <base.so> [<new.so>]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps

DEV = "cuda:0"
ops = [DitOps(_native.open_library(os.path.abspath(sys.argv[1]))), DitOps(_native.open_library(os.path.abspath(sys.argv[2])) if len(sys.argv) > 2 else None)]
g = torch.Generator(device=DEV).manual_seed(0)
W = 1024
for B, rows, kind in ((1, 4352, "mod"), (4, 4352, "mod"), (1, 4352, "weight_f32"), (1, 16512, "mod")):
    M = B * rows
    x = torch.randn(M, W, generator=g, device=DEV) * 2 + 0.5
    mod = torch.randn(B, 6 * W, generator=g, device=DEV)
    w = 1 + 0.2 * torch.randn(W, generator=g, device=DEV)
    if kind == "mod":
        call = lambda o: o.layernorm(x, None, mod[:, :W], mod[:, W:2 * W], rows_per_batch=rows)
    else:
        call = lambda o: o.layernorm(x, w, None, None, rows_per_batch=rows, out_f32=True)
    a, b = call(ops[0]), call(ops[1])
    torch.cuda.synchronize()
    print(f"B={B} rows={rows} {kind}: outputs bit-identical: {torch.equal(a, b)}; finite {bool(torch.isfinite(b.float()).all())}", flush=True)
    n, reps = 30, 8
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)] for _ in ops]
    for i in range(n):
        for k, o in enumerate(ops):
            e0, e1 = ev[k][i]
            e0.record()
            for _ in range(reps):
                call(o)
            e1.record()
    torch.cuda.synchronize()
    med = [sorted(e0.elapsed_time(e1) * 1e3 / reps for e0, e1 in v)[n // 2] for v in ev]
    nbytes = M * W * (8.0 if kind != "mod" else 6.0)
    print(f"timing B={B} rows={rows} {kind}: base {med[0]:.2f} us ({nbytes / med[0] / 1e6:.2f} TB/s)  new {med[1]:.2f} us ({nbytes / med[1] / 1e6:.2f} TB/s)  "
          f"ratio {med[1] / med[0]:.3f}  (per launch, {reps} back to back incl. the output allocation)", flush=True)
