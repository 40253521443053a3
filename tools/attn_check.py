"""Attention forward on the GPU vs an fp64 reference at production shapes: prints rel-L2 / max-abs for the rows handled by
the matrix-pipe path and for the L % 32 tail rows separately, then times the kernel with HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd.dit import DitOps

DEV = "cuda:0"
ops = DitOps()
bfr = lambda x: x.to(torch.bfloat16)
for L, B in ((4098, 1), (258, 2), (1026, 1), (16386, 1)):
    heads = 16
    lpad = (L + 255) // 256 * 256
    g = torch.Generator(device=DEV).manual_seed(L)
    q, k, v = (torch.randn(B, heads, lpad, 64, generator=g, device=DEV) for _ in range(3))
    q[0, 3, 5] *= 8.0
    k[0, 3, L - 1] *= 8.0
    qb, kb, vb = bfr(q), bfr(k), bfr(v)
    qk = torch.cat([qb.permute(0, 2, 1, 3).reshape(B * lpad, heads * 64), kb.permute(0, 2, 1, 3).reshape(B * lpad, heads * 64)], 1).contiguous()
    vt = vb.permute(0, 1, 3, 2).reshape(B, heads * 64, lpad).contiguous()
    lse2 = torch.zeros(B, heads, lpad, device=DEV)
    out = ops.attention(qk, vt, L, heads, lse2=lse2).float().reshape(B, lpad, heads, 64).permute(0, 2, 1, 3)[:, :, :L]
    if L <= 4098:
        s = (qb.double()[:, :, :L] @ kb.double()[:, :, :L].transpose(-1, -2)) * 0.125
        ref = (s.softmax(-1) @ vb.double()[:, :, :L])[:, :, :L]
        lref = torch.logsumexp(s, -1) * 1.4426950408889634
        nf = L // 32 * 32
        rel = lambda a, b: float((a.double() - b).norm() / b.norm())
        print(f"L={L} B={B}: main rel {rel(out[:, :, :nf], ref[:, :, :nf]):.2e} max {float((out[:, :, :nf].double() - ref[:, :, :nf]).abs().max()):.2e}"
              f" | tail rel {rel(out[:, :, nf:], ref[:, :, nf:]):.2e} max {float((out[:, :, nf:].double() - ref[:, :, nf:]).abs().max()):.2e}"
              f" | lse2 max err {float((lse2[:, :, :L].double() - lref).abs().max()):.2e}", flush=True)
    for _ in range(3):
        ops.attention(qk, vt, L, heads)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.attention(qk, vt, L, heads)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"L={L} B={B}: {us:.1f} us/launch  {4 * L * L * 64 * heads * B / us / 1e6:.0f} TFLOP/s", flush=True)
