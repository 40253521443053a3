"""Development tool: which Python call sites fill large tensors during one training step (torch.zeros / zeros_like / zero_ / fill_ /
full / ones above 32 MiB), with the calling line -- rocprof shows ~4 FillFunctor launches of ~1 GB per step and not who asks."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import numpy as np
import torch

LOG = []


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        out = orig(*a, **k)
        t = out if isinstance(out, torch.Tensor) else None
        if t is not None and t.is_cuda and t.numel() * t.element_size() >= (int(os.environ.get("FILL_MIN_MIB", "32")) << 20):
            st = [s for s in traceback.extract_stack()[:-1] if "find_fills" not in s.filename][-3:]
            LOG.append((name, t.numel() * t.element_size() / 2 ** 20, " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st))))
        return out
    setattr(owner, name, f)


for n in ("zeros", "zeros_like", "full", "ones", "ones_like"):
    wrap(torch, n)
for n in ("zero_", "fill_", "new_zeros"):
    wrap(torch.Tensor, n)

from dgs_amd import cameras, denoiser as dn, synth
from dgs_amd.train import DataParallelTrainer

dev = torch.device("cuda:0")
B, V, res, RV = 4, 4, 256, 10
model = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="relative_plk"), device=dev)
model.reset_parameters(seed=0)
model = model.to(dev)
model.train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-5, weight_decay=0.05, fused=True)
tr = DataParallelTrainer(model, opt)
batch, t = synth.make_batch(B, res, V=V, device=dev, seed=100, with_t=True)
rc2w = torch.tensor(np.stack([cameras.ring_cameras(RV, phase_deg=5.0 + 7 * b) for b in range(B)])).to(dev)
rk = torch.tensor(cameras.default_fxfycxcy(res)).expand(B, RV, 4).contiguous().to(dev)
target = torch.rand(B, RV, 3, res, res, device=dev)
for i in range(3):
    LOG.clear()
    tr.step(batch, t, target, rc2w, rk)
    torch.cuda.synchronize()
import collections
print("fills in the third step (>= FILL_MIN_MIB MiB), grouped by call site:")
cnt = collections.Counter((n, w) for n, _, w in LOG)
for (n, w), c in cnt.most_common(25):
    print(f"  {c:4d} x {n:12s} {w}")
LOG = [x for x in LOG if x[1] >= 32]
print("large fills in the third step:")
for name, mib, where in LOG:
    print(f"  {name:12s} {mib:9.1f} MiB  {where}")
print("(none)" if not LOG else "")
