"""The attention forward launch's timeline of every workgroup (tools' library, DGS_ATTN_DBG=16: start / loop end / tail tile + fold /
output stores issued / arrived / end on the constant 100 MHz clock), at L = 4,098 (two tail queries) and L = 4,096 (none).
    DGS_AMD_LIBRARY=open-diffusiongs_amd/lib/libdgs_hip_instr.so DGS_ATTN_DBG=28 python tools/attn_timeline.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd.dit import DitOps

DEV = "cuda:0"
ops = DitOps()
W, heads = 1024, 16
g = torch.Generator(device=DEV).manual_seed(0)
bf = lambda *s: torch.randn(*s, generator=g, device=DEV).to(torch.bfloat16)
for L in (4098, 4096):
    lpad = (L + 255) // 256 * 256
    qk, vt = bf(lpad, 2 * W), bf(1, W, lpad)
    out = torch.zeros(lpad, W, dtype=torch.bfloat16, device=DEV)
    for _ in range(4):
        ops.attention(qk, vt, L, heads, q_prescaled=True, out=out)
        torch.cuda.synchronize()
