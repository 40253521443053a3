"""Field-by-field comparison of two rasterizer states (restatement `RasterOracle`, reference `RasterRef`, or a dict of
arrays read back from the HIP product path).  Used by tests/test_raster_ref_gpu.py, oracle/make_raster_ref_golden.py and
tools/ref_compare.py."""
import numpy as np

INT_FIELDS = ("radii", "tiles_touched", "n_contrib")
FLOAT_STATE = ("depths", "means2D", "conic_opacity", "rgb", "cov3D")
GRADS = ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
         "dL_drotations")


def scenes():
    """(name, res, kwargs) of the comparison scenes: SURVEY 8d regimes at 64^2 / 256^2, SH degree 0..3, precomputed
    colour / covariance, a non-square frame.  Sizes are chosen so that the CPU restatement finishes in seconds."""
    return [
        ("init64", 64, dict(regime="init")),
        ("trained64", 64, dict(regime="trained")),
        ("small64", 64, dict(regime="small")),
        ("trained64_sh1", 64, dict(regime="trained", sh_degree=1)),
        ("trained64_sh2", 64, dict(regime="trained", sh_degree=2)),
        ("trained64_sh3", 64, dict(regime="trained", sh_degree=3)),
        ("trained64_precomp", 64, dict(regime="trained", precomp=True)),
        ("trained256", 256, dict(regime="trained")),
        ("init256", 256, dict(regime="init")),
    ]


def make_scene(res, regime="trained", sh_degree=0, precomp=False, seed=0, view=1, n_views=4):
    from dgs_amd import synth
    sc = synth.gaussian_scene(res, regime=regime, seed=seed, sh_degree=sh_degree)
    cams, _, _ = synth.render_cameras(res, n_views, phase_deg=10)
    kw = {}
    if precomp:
        rng = np.random.default_rng(seed + 7)
        P = sc["xyz"].shape[0]
        kw["colors_precomp"] = rng.uniform(0, 1, size=(P, 3)).astype(np.float32)
        s, q = sc["scales"].astype(np.float64), sc["rotations"].astype(np.float64)
        r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                      2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                      2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
        Sg = R @ (s[:, :, None] ** 2 * np.transpose(R, (0, 2, 1)))
        kw["cov3D_precomp"] = np.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], -1).astype(np.float32)
    return sc, cams[view], kw


def run(o, sc, cam, res, sh_degree=0, exp_mode=0, bg=(1.0, 1.0, 1.0), dpix=None, accum64=False, **kw):
    """Forward (+ backward when dpix is given) on any RasterOracle-shaped object."""
    args = dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    if "colors_precomp" in kw:
        args.update(shs=None, colors_precomp=kw["colors_precomp"])
    if "cov3D_precomp" in kw:
        args.update(scales=None, rotations=None, cov3D_precomp=kw["cov3D_precomp"])
    n = o.forward(np.asarray(bg, np.float32), sc["xyz"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
                  cam["tanfovx"], cam["tanfovy"], res, res, sh_degree=sh_degree, exp_mode=exp_mode, **args)
    if dpix is not None:
        o.backward(dpix, accum64=accum64)
    return n


def dpix_for(res, seed=0):
    return (np.random.default_rng(seed).normal(size=(3, res, res)) / (3 * res * res)).astype(np.float32)


def per_tile_lists(get):
    """Per-tile lists as far as they exist: the HIP path's scan form produces a tile's list on demand (a prefix, `list_len`)."""
    rng, pl = get("ranges").astype(np.int64), get("point_list")
    try:
        have = get("list_len").astype(np.int64)
    except KeyError:
        have = rng[:, 1] - rng[:, 0]
    return [pl[a:a + h] for (a, b), h in zip(rng, have)]


def compare(a, b, grads=False):
    """a, b: callables name -> array.  Returns a flat dict of mismatch statistics."""
    st = {}
    for k in INT_FIELDS:
        st[f"{k}_mismatch"] = int(np.count_nonzero(a(k).astype(np.int64).ravel() != b(k).astype(np.int64).ravel()))
    vis = (a("radii") > 0) & (b("radii") > 0)
    for k in FLOAT_STATE:
        try:
            x, y = a(k), b(k)
        except KeyError:      # the HIP path keeps no cov3D copy
            continue
        x, y = x.reshape(x.shape[0], -1)[vis], y.reshape(y.shape[0], -1)[vis]
        st[f"{k}_bitdiff"] = int(np.count_nonzero(x.view(np.uint32) != y.view(np.uint32)))
        st[f"{k}_maxrel"] = float(np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-20), initial=0.0))
    la, lb = per_tile_lists(a), per_tile_lists(b)
    ra, rb = a("ranges").astype(np.int64), b("ranges").astype(np.int64)
    st["num_rendered"] = (int((ra[:, 1] - ra[:, 0]).sum()), int((rb[:, 1] - rb[:, 0]).sum()))
    st["tiles_len_mismatch"] = int(np.count_nonzero((ra[:, 1] - ra[:, 0]) != (rb[:, 1] - rb[:, 0])))
    st["tiles_order_mismatch"] = sum(1 for x, y in zip(la, lb) if not np.array_equal(x[:min(len(x), len(y))], y[:min(len(x), len(y))]))
    ca, cb = a("out_color").astype(np.float64), b("out_color").astype(np.float64)
    st["color_maxabs"] = float(np.abs(ca - cb).max())
    mse = float(np.mean((np.clip(ca, 0, 1) - np.clip(cb, 0, 1)) ** 2))
    st["color_psnr_db"] = float(200.0 if mse == 0 else -10 * np.log10(mse))
    st["final_T_maxabs"] = float(np.abs(a("final_T").astype(np.float64) - b("final_T")).max())
    if grads:
        for k in GRADS:
            x, y = a(k).astype(np.float64), b(k).astype(np.float64)
            if x.size == 0 or not (np.any(x) or np.any(y)):
                continue
            st[f"{k}_relmax"] = float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-30))
    return st


def hip_state(backend, sc, cam, res, device, sh_degree=0, dpix=None, bg=(1.0, 1.0, 1.0), **kw):
    """Runs the HIP product path (one view) and returns a name -> array callable with the RasterOracle field names."""
    import torch
    from parity_util import run_backend_forward
    out = run_backend_forward(backend, sc, [cam], res, res, device, bg, sh_degree, kw.get("colors_precomp"),
                              kw.get("cov3D_precomp"), 1)
    n, color, radii, geom, binning, img = out
    P = sc["xyz"].shape[0]
    T = ((res + 15) // 16) ** 2
    rd = lambda name, dt, cnt: backend.state_read(name, P, res, res, 1, n, geom, binning, img, dt, cnt).cpu().numpy()
    vals = {"out_color": color[0].cpu().numpy(), "radii": radii[0].cpu().numpy(),
            "depths": rd("depths", torch.float32, P), "means2D": rd("means2D", torch.float32, 2 * P).reshape(P, 2),
            "conic_opacity": rd("conic_opacity", torch.float32, 4 * P).reshape(P, 4),
            "rgb": rd("rgb", torch.float32, 4 * P).reshape(P, 4)[:, :3].copy(),
            "tiles_touched": rd("tiles_touched", torch.int32, P), "ranges": rd("ranges", torch.int32, 2 * T).reshape(T, 2),
            "n_contrib": rd("n_contrib", torch.int32, res * res).reshape(res, res),
            "final_T": rd("final_T", torch.float32, res * res).reshape(res, res),
            "point_list": rd("point_list", torch.int32, max(int(n), 1))[: int(n)], "list_len": rd("list_len", torch.int32, T)}
    if dpix is not None:
        t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
        use_sh, use_sr = "colors_precomp" not in kw, "cov3D_precomp" not in kw
        g = backend.backward_views(
            t(bg), t(sc["xyz"])[None], radii, None if use_sh else t(kw["colors_precomp"])[None], t(sc["opacities"]),
            t(sc["scales"])[None] if use_sr else None, t(sc["rotations"])[None] if use_sr else None, 1.0,
            None if use_sr else t(kw["cov3D_precomp"])[None], t(cam["viewmatrix"])[None], t(cam["projmatrix"])[None],
            t(cam["campos"])[None], None, cam["tanfovx"], cam["tanfovy"], t(dpix)[None], t(sc["shs"])[None] if use_sh else None,
            sh_degree, geom, n, binning, img, True, views_per_set=1)
        M = sc["shs"].shape[1] if use_sh else 0
        vals.update({"dL_dmeans2D": g["means2D"][0].cpu().numpy(), "dL_dconic": g["conic"][0].cpu().numpy().reshape(P, 2, 2),
                     "dL_dopacity": g["opacity"].cpu().numpy().reshape(P, 1), "dL_dmeans3D": g["means3D"][0].cpu().numpy(),
                     "dL_dcov3D": g["cov3D"][0].cpu().numpy(),
                     "dL_dcolors": g["colors"][0].cpu().numpy() if not use_sh else np.zeros((P, 3), np.float32),
                     "dL_dsh": g["sh"][0].cpu().numpy() if use_sh else np.zeros((P, 0, 3), np.float32),
                     "dL_dscales": g["scales"][0].cpu().numpy() if use_sr else np.zeros((P, 3), np.float32),
                     "dL_drotations": g["rotations"][0].cpu().numpy() if use_sr else np.zeros((P, 4), np.float32)})
    return vals.__getitem__
