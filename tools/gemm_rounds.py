"""fc1-shaped GEMM (N = 4096, K = 1024) on the sliced kernels as a function of the tile count and the padding layout: finds
grid-quantisation effects (development tool)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps

DEV = "cuda:0"
ops = DitOps()
W = 1024
algos = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "4").split(",")]
for algo in algos:
    for M, valid in ((4096, 0), (4352, 4096), (4352, 4098), (4352, 4128)):
        xn = torch.randn(M, W, device=DEV).to(torch.bfloat16)
        w1 = (torch.randn(4 * W, W, device=DEV) * 0.02).to(torch.bfloat16)
        out = torch.empty(M, 4 * W, device=DEV, dtype=torch.bfloat16)
        fn = lambda: ops.gemm(xn, w1, None, _native.EPI_GELU_BF16, out=out, rows_per_batch=M, valid_rows=valid, algo=algo)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"algo {algo} M={M} valid={valid}: {us:.1f} us  {2 * (valid or M) * 4 * W * W / us / 1e6:.0f} TF/s", flush=True)
