"""What the L mod 32 = 2 learned-token queries cost the attention kernel: the same launch at L = 4096 and L = 4098.  (Event times include the wrapper's
scratch allocation: for the kernel alone run one L per process under tools/prof.sh.)  Development tool."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open-diffusiongs_amd"))
import torch

from dgs_amd.dit import DitOps

DEV = "cuda:0"
ops = DitOps()
lpad, W, heads = 4352, 1024, 16
g = torch.Generator(device=DEV).manual_seed(0)
bf = lambda *s: torch.randn(*s, generator=g, device=DEV).to(torch.bfloat16)
qk, vt = bf(lpad, 2 * W), bf(1, W, lpad)
for L in ([int(a) for a in sys.argv[1:]] or [4096, 4098, 4128]):
    for _ in range(3):
        ops.attention(qk, vt, L, heads)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.attention(qk, vt, L, heads)
    e1.record()
    torch.cuda.synchronize()
    print(f"L = {L}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us")
