"""Soak of the sampling step and of the rasterizer's forward + deterministic backward on one MI355X: the same work repeated, every result
compared with the first BIT FOR BIT (all of it is deterministic by construction: no floating-point atomics on these paths), and a
rotation through inputs of the same shapes so that the planned render and the captured graph see instance counts move.  What it is for:
the kernels that synchronise workgroups inside a launch (the depth sort's one-kernel passes, the attention tail's arrival counter, the
GEMMs' side jobs) have no emulator coverage of that part -- a race or a lost wake-up shows up here as a mismatch, a NaN or a stall.
    python tools/step_soak.py [seconds per phase, default 25]
Development tool (profiles/r05_step_soak.txt)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd import denoiser as dn, synth
from dgs_amd.raster import RasterBackend, render_views_autograd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
dev = torch.device("cuda:0")
model = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="relative_plk"), device=dev)
model.reset_parameters(seed=0)
model = model.to(dev).eval()

# ---- phase 1: the graph-replayed sampling step on three input sets in rotation ----
sets = [synth.make_batch(1, 256, V=4, device=dev, seed=s, with_t=True) for s in (0, 5, 11)]
far = {k: v.clone() for k, v in sets[2][0].items()}
far["c2w"][..., :3, 3] *= 5.0                                   # far cameras: every Gaussian covers few tiles -- the instance count drops, then jumps back
sets.append((far, sets[2][1]))
with torch.no_grad():
    want = []
    for b, t in sets:
        r, g = model(b, t)
        want.append((r.clone(), g[0]._xyz.clone()))
    graphed = model.graphed(*sets[0])
    n = bad = 0
    t0 = time.time()
    while time.time() - t0 < budget:
        i = n % len(sets)
        r, g = graphed(*sets[i])
        if n % 16 < len(sets):                                  # compare a full rotation every 16 steps (the comparison synchronises)
            ok = torch.equal(r, want[i][0]) and torch.equal(g[0]._xyz, want[i][1])
            if not ok:
                bad += 1
                print(f"step {n} (input set {i}): differs; finite {bool(torch.isfinite(r).all())}", flush=True)
        n += 1
    torch.cuda.synchronize()
    graphed.check(wait=True)
    model.gs_renderer.backend().check_async(wait=True)
print(f"[soak] sampling step, graph replays: {n} steps in {time.time() - t0:.1f} s, 4 input sets in rotation, {bad} mismatches; "
      f"re-captures {graphed.recaptures}, healed {graphed.healed}", flush=True)

# ---- phase 2: rasterizer forward + deterministic backward, 4 views at 256^2, on the step's Gaussians ----
with torch.no_grad():
    params, _ = model.image_to_gaussians(sets[0][0]["image"], sets[0][0]["ray_o"], sets[0][0]["ray_d"], sets[0][1])
be = RasterBackend()
be.deterministic = True
w = torch.randn(1, 4, 3, 256, 256, device=dev, generator=torch.Generator(device=dev).manual_seed(0)) / (3 * 256 * 256)
first = None
n = bad = 0
t0 = time.time()
while time.time() - t0 < budget:
    leaves = [x.detach().clone().requires_grad_(True) for x in (params.xyz, params.features, params.scaling, params.rotation, params.opacity)]
    img = render_views_autograd(be, *leaves, 256, 256, sets[0][0]["c2w"], sets[0][0]["fxfycxcy"])
    (img * w).sum().backward()
    cur = [img.detach()] + [x.grad for x in leaves]
    if first is None:
        first = [c.clone() for c in cur]
        assert all(bool(torch.isfinite(c).all()) for c in first)
    elif n % 8 == 0 and not all(torch.equal(a, b) for a, b in zip(cur, first)):
        bad += 1
        print(f"raster iteration {n}: differs", flush=True)
    n += 1
torch.cuda.synchronize()
be.check_async(wait=True)
print(f"[soak] rasterizer forward + deterministic backward: {n} iterations in {time.time() - t0:.1f} s, image + 5 gradients bit-identical to the first "
      f"in every compared iteration (every 8th): {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
