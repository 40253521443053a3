"""Development tool: per-tile distribution of the scan form's work in the random-init regime (4 views at 256^2): depth ranks a tile
tested (tile_scanned), entries it listed and walked.  Is the blend kernel's time the tiles that never saturate and test every rank?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import numpy as np
import torch

from dgs_amd import cameras, synth
from dgs_amd.raster import default_backend

dev = torch.device("cuda:0")
res, V = 256, 4
be = default_backend()
tt = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
T = ((res + 15) // 16) ** 2
for regime in ("init", "trained"):
    sc = synth.gaussian_scene(res, regime=regime, seed=0, activated=False)
    det = [tt(sc[k])[None] for k in ("xyz", "shs", "scales", "rotations", "opacities")]
    c2w = tt(cameras.ring_cameras(V, phase_deg=10))[None]
    k = tt(cameras.default_fxfycxcy(res)).expand(1, V, 4).contiguous()
    P = int(det[0].shape[1])
    view, proj, campos, tanfov = be.cameras_from_c2w(c2w, k, res, res)
    N, _c, radii, geom, binning, img = be.forward_views(torch.ones(3, device=dev), det[0], None, det[4].reshape(1, -1), det[2], det[3], 1.0, None, view, proj,
                                                        campos, tanfov, 0.0, 0.0, res, res, det[1], 0, False, False, views_per_set=V, raw_activations=True, planned=False)
    N = int(N)
    rd = lambda name, cnt: be.state_read(name, P, res, res, V, N, geom, binning, img, torch.int32, cnt).long().cpu().numpy()
    scanned, listed, walked = rd("tile_scanned", V * T), rd("list_len", V * T), rd("tile_work", V * T)
    pct = lambda a: " ".join(f"p{q}={int(np.percentile(a, q))}" for q in (50, 90, 99, 100))
    print(f"{regime}: P = {P}, {V * T} tiles")
    print(f"  depth ranks tested per tile   {pct(scanned)}   tiles that tested >= P/2: {int((scanned >= P // 2).sum())}, every rank: {int((scanned >= P).sum())}")
    print(f"  entries listed per tile       {pct(listed)}")
    print(f"  entries walked per tile       {pct(walked)}")
    if scanned.max() > 0:
        worst = np.argsort(-scanned)[:8]
        print("  the eight tiles that tested most: (view, tile x, tile y): ranks tested / listed / walked")
        for i in worst:
            print(f"    ({i // T}, {i % T % 16}, {i % T // 16}): {scanned[i]} / {listed[i]} / {walked[i]}")
