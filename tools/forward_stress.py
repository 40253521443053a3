"""Repeats the DiT forward on fixed inputs and compares every result with the first, bit for bit (the forward is deterministic):
a race shows up as a mismatch or NaN.  usage: python tools/forward_stress.py [res] [iters] [batch]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "open-diffusiongs_amd"))
import torch

from dgs_amd import denoiser as dn, synth

res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
model = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="relative_plk"), device=dev)
model.reset_parameters(seed=0)
batch, t = synth.make_batch(B, res, V=4, device=dev, seed=0, with_t=True)
eng = model.engine()
first = None
bad = 0
for i in range(iters):
    with torch.no_grad():
        params, _ = eng.image_to_gaussians(batch["image"], batch["ray_o"], batch["ray_d"], t)
    cur = torch.cat([v.float().flatten() for k, v in sorted(params.items()) if torch.is_tensor(v)])
    if first is None:
        first = cur.clone()
        print("first: finite", bool(torch.isfinite(first).all()), "norm", float(first.norm()))
    elif not torch.equal(cur, first):
        bad += 1
        print("iteration", i, "differs: finite", bool(torch.isfinite(cur).all()), "max abs diff", float((cur - first).abs().nan_to_num(1e30).max()))
print("res", res, "B", B, "iters", iters, "mismatches", bad, "env", {k: v for k, v in os.environ.items() if k.startswith("DGS_")})
