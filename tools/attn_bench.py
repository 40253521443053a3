"""Runs the attention / GEMM kernels a few times at production shape (target of tools/pmc_run.py and rocprofv3)."""
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
L, lpad, W, heads = 4098, 4352, 1024, 16
g = torch.Generator(device=DEV).manual_seed(0)
bf = lambda *s: torch.randn(*s, generator=g, device=DEV).to(torch.bfloat16)
qk, vt = bf(lpad, 2 * W), bf(1, W, lpad)
xn, h = bf(lpad, W), bf(lpad, 4 * W)
w1, w2, wq = bf(4 * W, W) * 0.02, bf(W, 4 * W) * 0.02, bf(3 * W, W) * 0.02
x = torch.randn(lpad, W, device=DEV)
gate = torch.randn(1, W, device=DEV)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(iters):
    ops.attention(qk, vt, L, heads)
    ops.gemm(xn, w1, None, _native.EPI_GELU_BF16, rows_per_batch=lpad, valid_rows=L)
    ops.gemm(h, w2, None, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, rows_per_batch=lpad, valid_rows=L)
    ops.gemm(xn, wq, None, _native.EPI_QKV, rows_per_batch=lpad, valid_rows=L)
torch.cuda.synchronize()
print("done")
