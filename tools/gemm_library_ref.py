"""Reference point, not product code: what the vendor GEMM library (hipBLASLt behind torch.matmul) reaches on the four GEMM shapes
of a DiT block at one sample (M = 4352 padded rows), plain bf16 GEMM without any epilogue -- next to the product's fused kernels
(which also apply bias / GELU / gate + residual / the V^T transposed copy).  Tells how much of the gap to the MFMA peak is the
shape (one round of tiles on 256 CUs, K = 1024) rather than the kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps

DEV, M, L, W = "cuda:0", 4352, 4098, 1024
ops = DitOps()
g = torch.Generator(device=DEV).manual_seed(0)
bf = lambda *s: torch.randn(*s, generator=g, device=DEV).to(torch.bfloat16)
iters = 50
for name, N, K, epi in (("qkv", 3 * W, W, _native.EPI_QKV), ("proj", W, W, _native.EPI_GATE_RESIDUAL), ("fc1", 4 * W, W, _native.EPI_GELU_BF16),
                        ("fc2", W, 4 * W, _native.EPI_GATE_RESIDUAL)):
    A, Wt, bias = bf(M, K), bf(N, K) * 0.05, torch.randn(N, generator=g, device=DEV)
    x0, gate = torch.randn(M, N, generator=g, device=DEV), torch.randn(1, N, generator=g, device=DEV)
    out_lib = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    qk = torch.empty(M, 2 * N // 3, dtype=torch.bfloat16, device=DEV) if epi == _native.EPI_QKV else None
    vt = torch.empty(1, N // 3, M, dtype=torch.bfloat16, device=DEV) if epi == _native.EPI_QKV else None
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)

    def lib():
        torch.matmul(A, Wt.t(), out=out_lib)

    def mine():
        kw = dict(rows_per_batch=M, valid_rows=L)
        if epi == _native.EPI_GATE_RESIDUAL:
            ops.gemm(A, Wt, bias, epi, out=x0, gate=gate, **kw)
        elif epi == _native.EPI_QKV:
            ops.gemm(A, Wt, bias, epi, out=qk, vt=vt, **kw)
        else:
            ops.gemm(A, Wt, bias, epi, out=out, **kw)

    res = {}
    for tag, fn in (("library", lib), ("product", mine)):
        for _ in range(5):
            fn()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for e0, e1 in ev:
            e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        res[tag] = sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in ev)[iters // 2]
    fl = 2.0 * M * N * K
    print(f"{name} [{M} x {N} x {K}]: vendor library (plain GEMM) {res['library']:.1f} us = {fl / res['library'] / 1e6:.0f} TFLOP/s on the padded rows; "
          f"product (fused epilogue) {res['product']:.1f} us = {fl / res['product'] / 1e6:.0f} TFLOP/s", flush=True)
