"""Generates tests/golden/raster_ref_*.npz from the REFERENCE rasterizer's own code (oracle/_ref, strict build) running
on an MI355X:   gpurun -- python oracle/make_raster_ref_golden.py   (writes into gpurun_out/golden/, copy to tests/golden/).

The fixtures pin oracle/raster_oracle.cpp on the CPU (tests/test_oracle_ref_golden.py): fields the restatement must
reproduce BIT-exactly are stored as SHA-256 digests of the full arrays (plus a strided sample to debug a mismatch with),
tolerance-compared fields (colour, final_T, gradients) as arrays / strided samples.  TEST INFRASTRUCTURE ONLY.
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open-diffusiongs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np

import ref_util as U

EXACT_INT = ("radii", "tiles_touched", "ranges", "point_list")
EXACT_FLOAT = ("depths", "means2D", "conic_opacity", "rgb", "cov3D")      # compared on visible Gaussians only
GRAD_STRIDE = 32
STATE_STRIDE = 8


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def exact_fields(get, precomp):
    """name -> array whose bytes must match between the reference and the restatement."""
    vis = get("radii") > 0
    out = {k: get(k) for k in EXACT_INT}
    for k in EXACT_FLOAT:
        if precomp and k in ("rgb", "cov3D"):     # the reference never writes them when colours / covariances are inputs
            continue
        a = get(k)
        out[k] = a.reshape(a.shape[0], -1)[vis]
    return out


def main():
    from oracle.raster_ref import RasterRef
    out_dir = os.path.join(ROOT, "gpurun_out", "golden")
    os.makedirs(out_dir, exist_ok=True)
    for name, res, kw in U.scenes():
        deg = kw.get("sh_degree", 0)
        sc, cam, extra = U.make_scene(res, **kw)
        dpix = U.dpix_for(res)
        r = RasterRef("strict")
        n = U.run(r, sc, cam, res, deg, dpix=dpix, **extra)
        precomp = bool(extra)
        rec = {"num_rendered": np.int64(n), "res": np.int64(res), "sh_degree": np.int64(deg)}
        small = res <= 64
        for k, a in exact_fields(r.get, precomp).items():
            rec["sha_" + k] = np.array(digest(a))
            if small:                                   # a sample to debug a digest mismatch with
                rec["sample_" + k] = a[::STATE_STRIDE].copy()
        px = 1 if small else 4
        rec["px_stride"] = np.int64(px)
        rec["out_color"] = r.get("out_color")[:, ::px, ::px].copy()
        rec["final_T"] = r.get("final_T")[::px, ::px].copy()
        rec["n_contrib"] = r.get("n_contrib")[::px, ::px].astype(np.uint16 if r.get("n_contrib").max() < 65536 else np.uint32)
        rec["sha_n_contrib"] = np.array(digest(r.get("n_contrib")))
        for k in U.GRADS:
            g = r.get(k)
            if g.size == 0:
                continue
            rec["grad_" + k] = g[::GRAD_STRIDE].copy()
            rec["gradsum_" + k] = g.astype(np.float64).sum(axis=0)
            rec["gradmax_" + k] = np.float64(np.abs(g).max())
        path = os.path.join(out_dir, f"raster_ref_{name}.npz")
        np.savez_compressed(path, **rec)
        print(name, n, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
