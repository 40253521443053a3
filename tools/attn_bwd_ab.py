"""A/B of builds of the library on the attention BACKWARD kernels (dQ + dK/dV launches of dgs_dit_attention_backward): bit-compare
of dqkv against the first library given (the builds are expected to execute the same arithmetic in the same order), then
alternating timed calls (HIP events around the pair of launches).
    python tools/attn_bwd_ab.py <base.so> <variant.so> [<variant.so> ...]        (AB_TIMING_ONLY=1: knock-out builds, no comparison)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps, _p, _stream

DEV = "cuda:0"
names = [os.path.basename(p) for p in sys.argv[1:]]
libs = [DitOps(_native.open_library(os.path.abspath(p))) for p in sys.argv[1:]]
heads, W = 16, 1024
g = torch.Generator(device=DEV).manual_seed(0)


def case(L, B):
    lpad = (L + 255) // 256 * 256
    qkv = torch.randn(B, lpad, 3 * W, generator=g, device=DEV).to(torch.bfloat16)
    dO = torch.zeros(B, lpad, W, device=DEV)
    dO[:, :L] = torch.randn(B, L, W, generator=g, device=DEV)
    dO = dO.to(torch.bfloat16)
    qkv2, qkvT = qkv.reshape(B * lpad, 3 * W).contiguous(), qkv.transpose(1, 2).contiguous()
    lse2 = torch.zeros(B, heads, lpad, device=DEV)
    o = libs[0].attention(qkv2, qkvT, L, heads, qkv_layout=True, lse2=lse2)
    return qkv2, qkvT, o, dO.reshape(B * lpad, W).contiguous(), dO.transpose(1, 2).contiguous(), lse2, lpad


def prepared(ops, c, L):
    """the call with its outputs allocated once: what is timed is the two launches"""
    qkv2, qkvT, o, dO, dOT, lse2, lpad = c
    dqkv, D = torch.zeros_like(qkv2), torch.zeros_like(lse2)
    a = _native.DgsDitAttentionBackwardArgs()
    a.B, a.heads, a.L, a.lpad = qkvT.shape[0], heads, L, lpad
    a.qkv, a.qkvT, a.o, a.dO, a.dOT, a.lse2, a.D, a.dqkv = (_p(t) for t in (qkv2, qkvT, o, dO, dOT, lse2, D, dqkv))
    a.scale = 0.125
    st = _stream(qkv2.device)
    return (lambda: ops._check(ops.lib.dgs_dit_attention_backward(ctypes.byref(a), st))), (dqkv, D)


CHECKS = () if os.environ.get("AB_TIMING_ONLY") else ((4098, 1), (4130, 1), (290, 2), (18, 2), (64, 1), (128, 1), (1026, 1), (600, 1), (264, 2), (257, 1))
for L, B in CHECKS:
    c = case(L, B)
    outs = []
    for ops in libs:
        ops.poison_lds()
        outs.append(ops.attention_backward(*c[:6], L, heads))
    torch.cuda.synchronize()
    lpad = c[6]
    ref = outs[0].view(B, lpad, 3 * W)
    for n, o in zip(names[1:], outs[1:]):
        v = o.view(B, lpad, 3 * W)
        full = L // 256 * 256 if L >= 256 else L          # rows of the MFMA workgroups; behind them the tail-token workgroups (<= 8 rows)
        tail = ""
        if full < L:
            d, r = v[:, full:L].float() - ref[:, full:L].float(), ref[:, full:L].float()
            tail = f"; rows {full}..{L - 1}: rel L2 diff {float(d.norm() / r.norm()):.3g}"
        print(f"L={L} B={B} {n}: rows < {full} bit-identical to {names[0]}: {torch.equal(v[:, :full].view(torch.int16), ref[:, :full].view(torch.int16))}"
              f"{tail}; finite {bool(torch.isfinite(v.float()).all())}; padding rows zero: {float(v[:, L:].float().abs().max()) == 0.0}", flush=True)

for L, B, n in ((4098, 1, 30), (4098, 4, 12), (16386, 1, 4)):
    c = case(L, B)
    keep = [prepared(ops, c, L) for ops in libs]
    calls = [k[0] for k in keep]
    ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)] for _ in libs]
    for _ in range(3):
        for f in calls:
            f()
    for i in range(n):
        for k, f in enumerate(calls):
            e0, e1 = ev[k][i]
            e0.record(); f(); e1.record()
    torch.cuda.synchronize()
    med = [sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in v)[n // 2] for v in ev]
    flops = 2.0 * 7 * L * L * W * B           # the seven L x L x 64 products per head the two kernels execute
    print(f"timing L={L} B={B}: " + "  ".join(f"{nm} {m:.1f} us ({flops / m / 1e6:.0f} TFLOP/s executed, x{m / med[0]:.3f})" for nm, m in zip(names, med)),
          flush=True)
