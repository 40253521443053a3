"""Exploration / evidence: restatement (oracle) vs the reference's own code (oracle/_ref) vs the HIP product path on the
comparison scenes of tests/ref_util.py; writes gpurun_out/ref_compare.json.  Needs a GPU (the reference runs there)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open-diffusiongs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch

import ref_util as U
from oracle.raster_oracle import RasterOracle
from oracle.raster_ref import RasterRef
from dgs_amd.raster import default_backend

out = {}
dev = torch.device("cuda:0")
only = sys.argv[1:] or None
for name, res, kw in U.scenes():
    if only and name not in only:
        continue
    deg = kw.get("sh_degree", 0)
    sc, cam, extra = U.make_scene(res, **kw)
    dpix = U.dpix_for(res)
    t0 = time.time()
    o0, o1 = RasterOracle(), RasterOracle()
    U.run(o0, sc, cam, res, deg, exp_mode=0, dpix=dpix, accum64=True, **extra)
    U.run(o1, sc, cam, res, deg, exp_mode=1, dpix=None, **extra)
    t_or = time.time() - t0
    rs, rf = RasterRef("strict"), RasterRef("fast")
    U.run(rs, sc, cam, res, deg, dpix=dpix, **extra)
    U.run(rf, sc, cam, res, deg, dpix=dpix, **extra)
    hip = U.hip_state(default_backend(), sc, cam, res, dev, deg, dpix=dpix, **extra)
    rec = {"P": int(sc["xyz"].shape[0]), "oracle_s": t_or,
           "oracle_libm_vs_ref_strict": U.compare(o0.get, rs.get, grads=True),
           "oracle_detexp_vs_ref_strict": U.compare(o1.get, rs.get),
           "ref_fast_vs_ref_strict": U.compare(rf.get, rs.get, grads=True),
           "hip_vs_ref_strict": U.compare(hip, rs.get, grads=True),
           "hip_vs_ref_fast": U.compare(hip, rf.get, grads=True)}
    # the reference's own kernels on this GPU, timed (context for the product's numbers)
    rs2 = RasterRef("fast")
    args = dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    if "colors_precomp" in extra:
        args.update(shs=None, colors_precomp=extra["colors_precomp"])
    if "cov3D_precomp" in extra:
        args.update(scales=None, rotations=None, cov3D_precomp=extra["cov3D_precomp"])
    rs2.forward(np.ones(3, np.float32), sc["xyz"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                cam["tanfovx"], cam["tanfovy"], res, res, sh_degree=deg, repeat=11, **args)
    rs2.backward(dpix, repeat=11)
    rec["ref_fast_ms"] = {"forward": rs2.time_ms("forward"), "backward": rs2.time_ms("backward")}
    out[name] = rec
    print(name, json.dumps(rec), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ref_compare.json"), "w"), indent=1)
