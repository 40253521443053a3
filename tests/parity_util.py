"""Shared parity checks: HIP path (or its CPU-emulated build) vs the CPU oracle.  exact=True (dgs_raster.h `exact_exp` = 1): every
integer AND float artefact bit-exact; exact=False (the product default: hardware v_exp_f32 in the blend loops): everything that does
not depend on alpha bit-exact, colour / final_T within 1e-5, n_contrib equal on all but <= 1e-5 of the pixels."""
import contextlib

import numpy as np
import torch

from oracle.raster_oracle import RasterOracle
from util_scene import oracle_forward


@contextlib.contextmanager
def exp_mode(backend, exact):
    """The blend exponential of `backend` for the duration of a forward (+ backward) pair; None leaves the backend as it is."""
    old = backend.exact_exp
    if exact is not None:
        backend.exact_exp = bool(exact)
    try:
        yield backend
    finally:
        backend.exact_exp = old


def run_backend_forward(backend, sc, cams, H, W, device, bg=(1.0, 1.0, 1.0), sh_degree=0, colors_precomp=None,
                        cov3D_precomp=None, views_per_set=None, debug=True, exact=None):
    """Renders all `cams` in ONE batched call.  sc arrays are [P,...] (one set) or [S,P,...]."""
    with exp_mode(backend, exact):
        return _run_backend_forward(backend, sc, cams, H, W, device, bg, sh_degree, colors_precomp, cov3D_precomp, views_per_set, debug)


def _run_backend_forward(backend, sc, cams, H, W, device, bg, sh_degree, colors_precomp, cov3D_precomp, views_per_set, debug):
    t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    xyz = t(sc["xyz"])
    if xyz.dim() == 2:
        xyz = xyz[None]
    V = len(cams)
    vm = t(np.stack([c["viewmatrix"] for c in cams]))
    pm = t(np.stack([c["projmatrix"] for c in cams]))
    cam = t(np.stack([c["campos"] for c in cams]))
    same = all(c["tanfovx"] == cams[0]["tanfovx"] and c["tanfovy"] == cams[0]["tanfovy"] for c in cams)
    tanfov = None if same else t(np.array([[c["tanfovx"], c["tanfovy"]] for c in cams], np.float32))
    use_sh = colors_precomp is None
    use_sr = cov3D_precomp is None
    out = backend.forward_views(
        t(bg), xyz, t(colors_precomp), t(sc["opacities"]), t(sc["scales"]) if use_sr else None,
        t(sc["rotations"]) if use_sr else None, 1.0, t(cov3D_precomp), vm, pm, cam, tanfov, cams[0]["tanfovx"],
        cams[0]["tanfovy"], H, W, t(sc["shs"]) if use_sh else None, sh_degree, False, debug,
        views_per_set=views_per_set or V)
    return out


def assert_forward_parity(backend, sc, cams, H, W, device, bg=(1.0, 1.0, 1.0), sh_degree=0, views_per_set=None,
                          check_state=True, colors_precomp=None, cov3D_precomp=None, exact=True):
    """Comparison of every integer and float artefact of the forward pass, per view (see the module docstring for `exact`)."""
    out = run_backend_forward(backend, sc, cams, H, W, device, bg, sh_degree, colors_precomp, cov3D_precomp, views_per_set, exact=exact)
    n_total, color, radii, geom, binning, img = out
    V = len(cams)
    xyz = np.asarray(sc["xyz"])
    multi = xyz.ndim == 3
    P = xyz.shape[-2]
    vps = views_per_set or V
    T = ((W + 15) // 16) * ((H + 15) // 16)
    rd = lambda name, dt, cnt: backend.state_read(name, P, W, H, V, n_total, geom, binning, img, dt, cnt).cpu().numpy()
    if check_state:
        depths = rd("depths", torch.float32, V * P).reshape(V, P)
        means2D = rd("means2D", torch.float32, V * P * 2).reshape(V, P, 2)
        conic = rd("conic_opacity", torch.float32, V * P * 4).reshape(V, P, 4)
        rgb = rd("rgb", torch.float32, V * P * 4).reshape(V, P, 4)[:, :, :3]
        tiles = rd("tiles_touched", torch.int32, V * P).reshape(V, P)
        ranges = rd("ranges", torch.int32, V * T * 2).reshape(V, T, 2)
        ncon = rd("n_contrib", torch.int32, V * H * W).reshape(V, H, W)
        fT = rd("final_T", torch.float32, V * H * W).reshape(V, H, W)
        plist = rd("point_list", torch.int32, max(int(n_total), 1))[: int(n_total)]
        have = rd("list_len", torch.int32, V * T).reshape(V, T)
    total = 0
    for v, cam in enumerate(cams):
        s = v // vps
        scv = {k: (np.asarray(a)[s] if multi else np.asarray(a)) for k, a in sc.items()}
        o = RasterOracle()
        kw = {}
        if colors_precomp is not None:
            kw.update(colors_precomp=np.asarray(colors_precomp)[s] if multi else colors_precomp, shs=None)
        if cov3D_precomp is not None:
            kw.update(cov3D_precomp=np.asarray(cov3D_precomp)[s] if multi else cov3D_precomp, scales=None, rotations=None)
        n = oracle_forward(o, scv, cam, H, W, bg=bg, sh_degree=sh_degree, exp_mode=1 if exact else 0, **kw)
        total += n
        np.testing.assert_array_equal(radii[v].cpu().numpy(), o.get("radii"), err_msg=f"radii view {v}")
        if check_state:
            vis = o.get("radii") > 0
            np.testing.assert_array_equal(tiles[v], o.get("tiles_touched").astype(np.int32), err_msg="tiles_touched")
            np.testing.assert_array_equal(depths[v][vis].view(np.uint32), o.get("depths")[vis].view(np.uint32), err_msg="depth bits")
            np.testing.assert_array_equal(means2D[v][vis].view(np.uint32), o.get("means2D")[vis].view(np.uint32), err_msg="means2D bits")
            np.testing.assert_array_equal(conic[v][vis].view(np.uint32), o.get("conic_opacity")[vis].view(np.uint32), err_msg="conic bits")
            if colors_precomp is None:
                np.testing.assert_array_equal(rgb[v][vis].view(np.uint32), o.get("rgb")[vis].view(np.uint32), err_msg="rgb bits")
            org, opl = o.get("ranges"), o.get("point_list")
            for t in range(T):
                a, b = ranges[v, t]
                oa, ob = org[t]
                assert b - a == ob - oa, f"tile {t} of view {v}: length {b - a} vs oracle {ob - oa}"
                # the scan form produces a tile's list on demand: a prefix, at least as long as the blend walked
                n_have = int(have[v, t])
                gx = (W + 15) // 16
                walked = int(ncon[v, 16 * (t // gx):16 * (t // gx) + 16, 16 * (t % gx):16 * (t % gx) + 16].max())
                assert walked <= n_have <= b - a, f"tile {t} of view {v}: {n_have} list entries present, walked {walked}, length {b - a}"
                np.testing.assert_array_equal(plist[a:a + n_have], opl[oa:oa + n_have].astype(np.int32), err_msg=f"sorted list tile {t} view {v}")
            if exact:
                np.testing.assert_array_equal(ncon[v], o.get("n_contrib").astype(np.int32).reshape(H, W), err_msg="n_contrib")
                np.testing.assert_array_equal(fT[v].view(np.uint32), o.get("final_T").view(np.uint32).reshape(H, W), err_msg="final_T bits")
        if exact:
            np.testing.assert_array_equal(color[v].cpu().numpy().view(np.uint32), o.get("out_color").view(np.uint32),
                                          err_msg=f"colour bits view {v}")
        else:
            # the product exponential takes the oracle's side of the 1/255 alpha cut-off for every pair (guard band, csrc/dgs_device.h);
            # what remains is a pixel whose T crosses 1e-4 within an ulp (forward.cu:344): its last pair may be counted by one side only,
            # n_contrib differs by one there and the colour by <= 1e-4 x the pair's weight; every other pixel agrees to 1e-5
            allowed = max(2, int(1e-5 * H * W))
            if check_state:
                bad = int((ncon[v] != o.get("n_contrib").astype(np.int32).reshape(H, W)).sum())
                assert bad <= allowed, f"n_contrib differs on {bad} of {H * W} pixels (view {v})"    # the bar of tests/test_raster_ref_gpu.py
                dT = np.abs(fT[v] - o.get("final_T").reshape(H, W))
                assert int((dT > 1e-5).sum()) <= allowed and float(dT.max()) <= 5e-3, (v, int((dT > 1e-5).sum()), float(dT.max()))
            diff = np.abs(color[v].cpu().numpy() - o.get("out_color")).max(axis=0)
            off = int((diff > 1e-5).sum())
            assert off <= allowed and float(diff.max()) <= 5e-3, f"view {v}: {off} pixels differ by more than 1e-5 (max {float(diff.max()):.3g})"
    assert int(n_total) == total, (n_total, total)
    return out
