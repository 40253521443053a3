"""A fixed synthetic batch fitted by the full-size model (24 blocks, 460 M parameters) through `DataParallelTrainer.step` for a few hundred
steps, twice from the same seed: the loss has to fall (gradients of the rasterizer backward, the DiT backward, the clip and the fused AdamW all
point the same way over many steps -- what the per-kernel gradient parity tests cannot say), and the two runs have to agree bit for bit in
every logged loss and in a checksum of all parameters (the deterministic training default over a horizon 50 x longer than the test's).
Inputs: smooth colour patterns as the 4 input views; targets: the same patterns at the input cameras + 6 further ring cameras.
    python tools/train_soak.py [steps, default 150] [batch, default 4] [scene]
`scene`: the scene model at 512^2 (BASELINE configs[4]'s shapes: 4 input + 7 rendered views, P = 1,048,578, per-block recompute) with the render
cameras turned AWAY from the scene (180 degrees about their up axis) for the first third of the run: next to nothing is in view, the
rasterizer's plan sizes its buffers for that, and when the cameras turn back the instance count outgrows them by orders of magnitude --
in a training step, where the plan's capacity is below the worst case -- and the planned render has to notice, re-issue itself and keep the
step's result (dgs_amd/raster.py `_AsyncPlan`: calls["healed"]).
Development tool (profiles/r05_train_soak.txt, r05_train_soak_scene512.txt)."""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import numpy as np
import torch

from dgs_amd import cameras, denoiser as dn, synth
from dgs_amd.optim import FusedAdamW
from dgs_amd.train import DataParallelTrainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
SCENE = len(sys.argv) > 3 and sys.argv[3] == "scene"
dev = torch.device("cuda:0")
res, V, RV = (512, 4, 7) if SCENE else (256, 4, 10)


def pattern(n, seed):
    """n smooth RGB images in [0.1, 0.9]: two low-frequency plane waves per channel."""
    g = torch.Generator().manual_seed(seed)
    y, x = torch.meshgrid(torch.linspace(0, 1, res), torch.linspace(0, 1, res), indexing="ij")
    f = torch.rand(n, 3, 2, 2, generator=g) * 3.0 + 0.5
    ph = torch.rand(n, 3, 2, generator=g) * 2 * math.pi
    img = sum(torch.sin(2 * math.pi * (f[:, :, k, 0, None, None] * x + f[:, :, k, 1, None, None] * y) + ph[:, :, k, None, None]) for k in range(2))
    return (0.5 + 0.2 * img).clamp(0.1, 0.9)


batch, t = synth.make_batch(B, res, V=V, device=dev, seed=100, with_t=True)
imgs = pattern(B * RV, 7).reshape(B, RV, 3, res, res).to(dev)
batch["image"] = imgs[:, :V].contiguous()
extra = torch.tensor(np.stack([cameras.ring_cameras(RV - V, phase_deg=5.0 + 7 * b) for b in range(B)])).to(dev)
rc2w = torch.cat([batch["c2w"], extra.to(batch["c2w"].dtype)], 1).contiguous()
rk = torch.tensor(cameras.default_fxfycxcy(res)).expand(B, RV, 4).contiguous().to(dev)
target = imgs.contiguous()

runs = []
for run in range(2):
    if SCENE:
        m = dn.DGSDenoiserScene(dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="plk", use_checkpoint=True), device=dev)
        m.activation_budget_bytes = 0                   # every block recomputed in the backward, as the reference's use_checkpoint does
    else:
        m = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="relative_plk"), device=dev)
    m.reset_parameters(seed=0)
    m = m.to(dev)
    m.train()
    log = []
    away = rc2w.clone()
    away[..., :3, 0] *= -1.0                             # x and z axes flipped: the camera looks the other way
    away[..., :3, 2] *= -1.0
    with DataParallelTrainer(m, FusedAdamW(m, lr=1e-4, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05), max_grad_norm=0.5) as tr:
        t0 = time.time()
        for i in range(steps):
            close = SCENE and i < steps // 3
            loss = tr.step(batch, t, target, away if close else rc2w, rk)
            if i % (5 if SCENE else 10) == 0 or i == steps - 1:
                healed = sum(p.calls["healed"] for p in m.gs_renderer.backend()._plans.values())
                log.append((i, float(loss), float(tr.last_grad_sumsq) ** 0.5, healed, close))
        torch.cuda.synchronize()
        dt = time.time() - t0
        det = m.gs_renderer.backend().last_backward_deterministic
        plans = [dict(capacity=p.capacity, seen_max=p.seen_max, calls=p.calls) for p in m.gs_renderer.backend()._plans.values()] if hasattr(m.gs_renderer.backend(), "_plans") else None
    chk = sum(float(p.detach().double().sum()) for p in m.parameters())
    absmax = max(float(p.detach().abs().max()) for p in m.parameters())
    finite = all(bool(torch.isfinite(p).all()) for p in m.parameters())
    runs.append((log, chk))
    print(f"[train soak] run {run}: {steps} steps, B = {B}, {RV} rendered views at {res}^2, {dt / steps * 1e3:.1f} ms per step (incl. the logged losses' "
          f"synchronisations); deterministic backward {det}; parameters finite {finite}, |max| {absmax:.3f}, sum {chk!r}", flush=True)
    if run == 0:
        for i, l, g, h, c in log:
            print(f"    step {i:4d}  loss {l:.6f}  grad norm {g:.4f}" + (f"  renders healed so far {h}{'  (cameras turned away)' if c else ''}" if SCENE else ""), flush=True)
        if plans:
            print("    raster plans:", plans, flush=True)
    del m
    torch.cuda.empty_cache()
# the rasterizer backend (and its plans) is the process's: the second run starts with the first run's capacity and has nothing to heal --
# the heal counts differ by construction, everything computed must not
same = [x[:3] for x in runs[0][0]] == [x[:3] for x in runs[1][0]] and runs[0][1] == runs[1][1]
if SCENE:
    print(f"[train soak] renders healed: run 0 {runs[0][0][-1][3] - 0}, run 1 {runs[1][0][-1][3] - runs[0][0][-1][3]} (it inherits run 0's plan: same bits with and without the repeat)", flush=True)
l0, l1 = runs[0][0][0][1], runs[0][0][-1][1]
print(f"[train soak] loss {l0:.6f} -> {l1:.6f} ({l1 / l0:.3f} x); the two runs agree bit for bit in every logged loss, gradient norm and the parameter sum: {same}", flush=True)
sys.exit(0 if same and l1 < (1.0 if SCENE else 0.8) * l0 else 1)
