"""The attention backward (dQ + dK/dV launches) of the product library at one shape, n calls: the command rocprofv3 / tools/pmc_run.py
wrap.      python tools/attn_bwd_run.py L B n"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps, _p, _stream

L, B, n = (int(v) for v in sys.argv[1:4])
DEV, heads, W = "cuda:0", 16, 1024
ops = DitOps()
g = torch.Generator(device=DEV).manual_seed(0)
lpad = (L + 255) // 256 * 256
qkv = torch.randn(B, lpad, 3 * W, generator=g, device=DEV).to(torch.bfloat16)
dO = torch.zeros(B, lpad, W, device=DEV)
dO[:, :L] = torch.randn(B, L, W, generator=g, device=DEV)
dO = dO.to(torch.bfloat16)
qkv2, qkvT = qkv.reshape(B * lpad, 3 * W).contiguous(), qkv.transpose(1, 2).contiguous()
lse2 = torch.zeros(B, heads, lpad, device=DEV)
o = ops.attention(qkv2, qkvT, L, heads, qkv_layout=True, lse2=lse2)
dOr, dOT = dO.reshape(B * lpad, W).contiguous(), dO.transpose(1, 2).contiguous()
dqkv, D = torch.zeros_like(qkv2), torch.zeros_like(lse2)
a = _native.DgsDitAttentionBackwardArgs()
a.B, a.heads, a.L, a.lpad = B, heads, L, lpad
a.qkv, a.qkvT, a.o, a.dO, a.dOT, a.lse2, a.D, a.dqkv = (_p(t) for t in (qkv2, qkvT, o, dOr, dOT, lse2, D, dqkv))
a.scale = 0.125
st = _stream(qkv2.device)
for _ in range(n):
    ops._check(ops.lib.dgs_dit_attention_backward(ctypes.byref(a), st))
torch.cuda.synchronize()
