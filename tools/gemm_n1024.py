"""proj / fc2-shaped GEMMs (N = 1024: few 256-wide tiles) by kernel family, tile width and padding layout (development tool)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps

DEV = "cuda:0"
ops = DitOps()
W = 1024
algos = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "0,1,4").split(",")]
splitk = len(sys.argv) > 2 and sys.argv[2] == "splitk"
for name, K in (("proj", W), ("fc2", 4 * W)):
    for algo in algos:
        for M, valid in ((4352, 4096), (4352, 4098)):
            a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
            w = (torch.randn(W, K, device=DEV) * 0.02).to(torch.bfloat16)
            x = torch.randn(M, W, device=DEV)
            gate = torch.randn(1, W, device=DEV)
            bias = torch.randn(W, device=DEV)
            fn = lambda: ops.gemm(a, w, bias, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, rows_per_batch=M, valid_rows=valid, algo=algo, splitk=splitk)
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            print(f"{name} algo {algo} BN={os.environ.get('DGS_GEMM_BN', 'auto')} splitk={splitk} valid={valid}: {us:.1f} us  {2 * valid * W * K / us / 1e6:.0f} TF/s", flush=True)
