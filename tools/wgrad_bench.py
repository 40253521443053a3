"""Weight-gradient GEMM timing at the training shapes (B samples x lpad tokens): single-pass 128-wide kernel vs split-K on the
sliced 256 x 256 kernel.  Development tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps, DgsDitGemmArgs, _p, _stream

DEV = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    ops = DitOps()
    T = 4224
    for B in (4, 1):
        for N, K in ((3072, 1024), (1024, 4096), (4096, 1024), (1024, 1024)):
            dyT = (torch.randn(B, N, T, device=DEV) * 0.1).to(torch.bfloat16)
            xT = torch.randn(B, K, T, device=DEV).to(torch.bfloat16)
            out = torch.empty(N, K, device=DEV)
            nbytes = ops.lib.dgs_dit_gemm_splitk_bytes(N, K, B * T, T)
            ws = torch.empty(max(nbytes, 4) // 4, device=DEV)

            def call(split):
                a = DgsDitGemmArgs()
                a.M, a.N, a.K = N, K, B * T
                a.A, a.lda, a.W, a.ldw = _p(dyT), T, _p(xT), T
                a.k_per_batch, a.a_batch_stride, a.w_batch_stride = T, N * T, K * T
                a.epilogue, a.out, a.ldo = _native.EPI_F32, _p(out), K
                a.splitk_ws = _p(ws) if split and nbytes else None
                ops._check(ops.lib.dgs_dit_gemm(a, _stream(DEV)))

            fl = 2.0 * N * K * B * T
            t0, t1 = timeit(lambda: call(False)), timeit(lambda: call(True))
            print(f"B={B} dW[{N},{K}]: single-pass {t0 * 1e6:7.1f} us ({fl / t0 / 1e12:6.1f} TF/s) | split-K ({nbytes // (N * K * 4)} planes) "
                  f"{t1 * 1e6:7.1f} us ({fl / t1 / 1e12:6.1f} TF/s)")


if __name__ == "__main__":
    main()
