"""Times the DiT GEMM shapes per kernel family (HIP events around back-to-back launches); DGS_GEMM_DBG=1 prints in-kernel
cycle stamps of the sliced kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps

DEV = "cuda:0"
ops = DitOps()
L, lpad, W = int(os.environ.get("GEMM_VALID", "4098")), 4352, 1024   # GEMM_VALID=4096: full tiles only (no learned-token rows)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
g = torch.Generator(device=DEV).manual_seed(0)
bf = lambda *s: torch.randn(*s, generator=g, device=DEV).to(torch.bfloat16)
xn, h = bf(B * lpad, W), bf(B * lpad, 4 * W)
w1, w2, wq, wp = bf(4 * W, W) * 0.02, bf(W, 4 * W) * 0.02, bf(3 * W, W) * 0.02, bf(W, W) * 0.02
x = torch.randn(B * lpad, W, device=DEV)
gate = torch.randn(B, W, device=DEV)
algos = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "0,4").split(",")]
cases = {
    "qkv": lambda al: ops.gemm(xn, wq, None, _native.EPI_QKV, rows_per_batch=lpad, valid_rows=L, algo=al),
    "fc1": lambda al: ops.gemm(xn, w1, None, _native.EPI_GELU_BF16, rows_per_batch=lpad, valid_rows=L, algo=al),
    "fc2": lambda al: ops.gemm(h, w2, None, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, rows_per_batch=lpad, valid_rows=L, algo=al),
    "f32": lambda al: ops.gemm(xn, w1, None, _native.EPI_F32, rows_per_batch=lpad, valid_rows=L, algo=al),
    "proj": lambda al: ops.gemm(xn, wp, None, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, rows_per_batch=lpad, valid_rows=L, algo=al),
}
flops = {"f32": 2 * B * L * 4 * W * W, "qkv": 2 * B * L * 3 * W * W, "fc1": 2 * B * L * 4 * W * W, "fc2": 2 * B * L * 4 * W * W, "proj": 2 * B * L * W * W}
only = os.environ.get("GEMM_CASES")
for name, fn in cases.items():
    if only and name not in only.split(","):
        continue
    for al in algos:
        for _ in range(2):
            fn(al)
        n = 1 if os.environ.get("DGS_GEMM_DBG") else 10
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn(al)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print(f"{name} algo {al}: {us:.1f} us  {flops[name] / us / 1e6:.0f} TFLOP/s", flush=True)
