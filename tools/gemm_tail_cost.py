"""What the two learned-token rows cost each DiT GEMM at one sample: the same launch with valid_rows = 4096 (full tiles only) and
4098 (the shipped shape).  Development tool."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps

DEV = "cuda:0"
ops = DitOps()
M = 4352
for name, N, K, epi in (("qkv", 3072, 1024, _native.EPI_QKV), ("proj", 1024, 1024, _native.EPI_GATE_RESIDUAL),
                        ("fc1", 4096, 1024, _native.EPI_GELU_BF16), ("fc2", 1024, 4096, _native.EPI_GATE_RESIDUAL)):
    a = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    w = (torch.randn(N, K, device=DEV) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    x = torch.randn(M, N, device=DEV)
    gate = torch.randn(1, N, device=DEV)
    res = []
    for valid in (4096, 4098):
        kw = dict(rows_per_batch=M, valid_rows=valid)
        if epi == _native.EPI_GATE_RESIDUAL:
            kw.update(out=x, gate=gate)
        fn = lambda: ops.gemm(a, w, bias, epi, **kw)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"{name:5s} N={N} K={K}: full tiles {res[0]:6.1f} us   with the 2 learned-token rows {res[1]:6.1f} us   (+{res[1] - res[0]:.1f})")
