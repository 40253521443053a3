"""Per-kernel timing of the DiT path at production shapes (development tool; bench.py is the contract benchmark)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))

import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps

DEV = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--batch", type=int, default=1)
    a = ap.parse_args()
    ops = DitOps()
    W, heads = 1024, 16
    L = 2 + 4 * (a.res // 8) ** 2
    lpad = (L + 255) // 256 * 256
    M = a.batch * lpad
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g, device=DEV)
    bf = lambda t: t.to(torch.bfloat16)
    print(f"res {a.res} batch {a.batch}: L={L} lpad={lpad} M={M}")
    x = rnd(M, W)
    mod = rnd(a.batch, 6 * W)
    xn = bf(rnd(M, W)); h = bf(rnd(M, 4 * W))
    ws = {"qkv": bf(rnd(3 * W, W) * 0.02), "proj": bf(rnd(W, W) * 0.02), "fc1": bf(rnd(4 * W, W) * 0.02), "fc2": bf(rnd(W, 4 * W) * 0.02)}
    bias = {k: rnd(v.shape[0]) for k, v in ws.items()}
    qk = torch.empty(M, 2 * W, dtype=torch.bfloat16, device=DEV); vt = torch.empty(a.batch, W, lpad, dtype=torch.bfloat16, device=DEV)
    ob = torch.empty(M, 4 * W, dtype=torch.bfloat16, device=DEV)
    rows = []
    t = timeit(lambda: ops.layernorm(x, None, mod[:, :W], mod[:, W:2 * W], rows_per_batch=lpad))
    rows.append(("layernorm+modulate", t, M * W * 6 / t / 1e9, "GB/s"))
    t = timeit(lambda: ops.gemm(xn, ws["qkv"], bias["qkv"], _native.EPI_QKV, out=qk, vt=vt, rows_per_batch=lpad, valid_rows=L))
    rows.append(("gemm qkv  [M,1024]x[3072,1024]", t, 2 * M * 3 * W * W / t / 1e12, "TFLOP/s"))
    qkr = bf(rnd(M, 2 * W)); vtr = bf(rnd(a.batch, W, lpad))
    t = timeit(lambda: ops.attention(qkr, vtr, L, heads))
    rows.append((f"attention L={L}", t, 4 * L * L * W * a.batch / t / 1e12, "TFLOP/s"))
    t = timeit(lambda: ops.gemm(xn, ws["proj"], bias["proj"], _native.EPI_GATE_RESIDUAL, out=x, gate=mod[:, 2 * W:3 * W], rows_per_batch=lpad, valid_rows=L))
    rows.append(("gemm proj [M,1024]x[1024,1024] +gate+res", t, 2 * M * W * W / t / 1e12, "TFLOP/s"))
    t = timeit(lambda: ops.gemm(xn, ws["fc1"], bias["fc1"], _native.EPI_GELU_BF16, out=ob, rows_per_batch=lpad, valid_rows=L))
    rows.append(("gemm fc1  [M,1024]x[4096,1024] +gelu", t, 2 * M * 4 * W * W / t / 1e12, "TFLOP/s"))
    t = timeit(lambda: ops.gemm(h, ws["fc2"], bias["fc2"], _native.EPI_GATE_RESIDUAL, out=x, gate=mod[:, 2 * W:3 * W], rows_per_batch=lpad, valid_rows=L))
    rows.append(("gemm fc2  [M,4096]x[1024,4096] +gate+res", t, 2 * M * 4 * W * W / t / 1e12, "TFLOP/s"))
    cvec = rnd(a.batch, W); adaw = bf(rnd(148 * W, W) * 0.02)
    t = timeit(lambda: ops.rowlinear(cvec, adaw, None, silu_input=True))
    rows.append(("adaLN GEMV [B,1024]x[151552,1024]", t, 148 * W * W * 2 / t / 1e9, "GB/s"))
    tot = 0.0
    for name, t, rate, unit in rows:
        print(f"  {name:44s} {t * 1e6:9.1f} us   {rate:9.1f} {unit}")
    per_layer = rows[0][1] * 2 + rows[1][1] + rows[2][1] + rows[3][1] + rows[4][1] + rows[5][1]
    print(f"  per DiT block (sum of isolated kernels): {per_layer * 1e6:.1f} us -> 24 blocks {per_layer * 24 * 1e3:.2f} ms")

    # whole model
    from dgs_amd import denoiser as dn
    sys.path.insert(0, ROOT)
    from bench import dit_flops
    from dgs_amd import synth
    m = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=24), device=DEV)
    m.reset_parameters(seed=0)
    batch, tt = synth.make_batch(a.batch, a.res, V=4, device=torch.device(DEV), seed=0, with_t=True)
    eng = m.engine()
    t = timeit(lambda: eng.image_to_gaussians(batch["image"], batch["ray_o"], batch["ray_d"], tt), iters=10)
    print(f"  image_to_gaussians (whole DiT): {t * 1e3:.3f} ms = {dit_flops(L) * a.batch / t / 1e12:.1f} TFLOP/s ({dit_flops(L) * a.batch / t / 2.5e15 * 100:.1f}% of bf16 MFMA peak)")
    params, _ = eng.image_to_gaussians(batch["image"], batch["ray_o"], batch["ray_d"], tt)
    p = dn.AttrDict(params)
    t2 = timeit(lambda: m.render_gaussians(p, batch["c2w"], batch["fxfycxcy"], a.res, a.res), iters=10)
    print(f"  render_gaussians ({a.batch * 4} views, random-init Gaussians): {t2 * 1e3:.3f} ms")
    t3 = timeit(lambda: m(batch, tt), iters=10)
    print(f"  DGSDenoiser.forward: {t3 * 1e3:.3f} ms -> {a.batch * 4 / t3:.1f} renders/s")


if __name__ == "__main__":
    main()
