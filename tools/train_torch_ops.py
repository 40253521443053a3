"""Which torch ops (and hipMemcpy launches) a steady training step still issues around the library's own kernels, grouped by the Python
line that asks for them (torch.profiler with stacks): rocprof shows ~125 `copyBuffer` and ~60 `at::native` element-wise launches per step
and not who asks.
    python tools/train_torch_ops.py > gpurun_out/train_torch_ops.txt"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile

from dgs_amd import cameras, denoiser as dn, synth
from dgs_amd.optim import FusedAdamW
from dgs_amd.train import DataParallelTrainer

dev = torch.device("cuda:0")
B, V, res, RV = 4, 4, 256, 10
model = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="relative_plk"), device=dev)
model.reset_parameters(seed=0)
model = model.to(dev)
model.train()
tr = DataParallelTrainer(model, FusedAdamW(model, lr=1e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05), max_grad_norm=0.5)
batch, t = synth.make_batch(B, res, V=V, device=dev, seed=100, with_t=True)
rc2w = torch.tensor(np.stack([cameras.ring_cameras(RV, phase_deg=5.0 + 7 * b) for b in range(B)])).to(dev)
rk = torch.tensor(cameras.default_fxfycxcy(res)).expand(B, RV, 4).contiguous().to(dev)
target = torch.rand(B, RV, 3, res, res, device=dev)
for _ in range(3):
    tr.step(batch, t, target, rc2w, rk)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(batch, t, target, rc2w, rk)
    torch.cuda.synchronize()
by_site = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue                                           # top-level aten ops only
    kern = [k for k in ev.kernels] if hasattr(ev, "kernels") else []
    if not kern:
        continue
    site = next((s for s in ev.stack if "open-diffusiongs_amd" in s or "bench.py" in s or "tools/" in s), ev.stack[0] if ev.stack else "?")
    by_site[(ev.name, site.strip()[-110:], len(kern))] += 1
print("# top-level aten ops of ONE steady training step that launch device work: count x (op, device launches per call, innermost dgs_amd frame)")
for (name, site, nk), c in sorted(by_site.items(), key=lambda kv: -kv[1] * kv[0][2]):
    print(f"{c:5d} x {name:28s} launches/call {nk}   {site}")
print("total device launches from torch ops:", sum(c * k[2] for k, c in by_site.items()))
