"""Backward parity helper: HIP path (or its CPU-emulated build) vs the CPU oracle.  Gradients are sums of fp32 terms in an
order the reference leaves unspecified (global atomics), so the bar is relative, against the oracle's fp64-accumulated
sums: |g - g_ref| <= rtol * max|g_ref| + atol per tensor."""
import numpy as np
import torch

from oracle.raster_oracle import RasterOracle
from parity_util import exp_mode, run_backend_forward
from util_scene import oracle_forward

NAMES = {"means2D": "dL_dmeans2D", "colors": "dL_dcolors", "opacity": "dL_dopacity", "means3D": "dL_dmeans3D",
         "cov3D": "dL_dcov3D", "sh": "dL_dsh", "scales": "dL_dscales", "rotations": "dL_drotations"}


OBSERVED = []          # (what, max |got - ref| / max |ref|) of every comparison: tools/raster_grad_error.py reads it


def close(got, ref, rtol=2e-4, what=""):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    scale = max(float(np.abs(ref).max()), 1e-12)
    err = float(np.abs(got - ref).max())
    OBSERVED.append((what, err / scale))
    assert err <= rtol * scale + 1e-9, f"{what}: max err {err:.3e} vs scale {scale:.3e}"


def assert_backward_parity(backend, sc, cams, H, W, device, sh_degree=0, bg=(1.0, 1.0, 1.0), seed=0, colors_precomp=None,
                           cov3D_precomp=None, exact=True, rtol=2e-4):
    """One Gaussian set rendered into len(cams) views in ONE batched call; oracle: per view, gradients summed over views.
    exact=True: the oracle's own exponential in the blend loops (every float bit-identical with the oracle; the bar measures summation
    order only); exact=False / None: the product default (compensated v_exp_f32 + the cut-off guard band, csrc/dgs_device.h): the same
    bar holds (profiles/r06_raster_grad_error.txt)."""
    with exp_mode(backend, exact):
        return _assert_backward_parity(backend, sc, cams, H, W, device, sh_degree, bg, seed, colors_precomp, cov3D_precomp, rtol)


def _assert_backward_parity(backend, sc, cams, H, W, device, sh_degree, bg, seed, colors_precomp, cov3D_precomp, rtol):
    global close
    _close = close
    close = lambda got, ref, what="": _close(got, ref, rtol=rtol, what=what)
    try:
        return _assert_backward_parity_body(backend, sc, cams, H, W, device, sh_degree, bg, seed, colors_precomp, cov3D_precomp)
    finally:
        close = _close


def _assert_backward_parity_body(backend, sc, cams, H, W, device, sh_degree, bg, seed, colors_precomp, cov3D_precomp):
    t = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    V = len(cams)
    out = run_backend_forward(backend, sc, cams, H, W, device, bg, sh_degree, colors_precomp, cov3D_precomp, V, debug=True)
    n_total, color, radii, geom, binning, img = out
    rng = np.random.default_rng(seed)
    dpix = rng.normal(size=(V, 3, H, W)).astype(np.float32) / (3 * H * W)
    use_sh, use_sr = colors_precomp is None, cov3D_precomp is None
    vm = t(np.stack([c["viewmatrix"] for c in cams])); pm = t(np.stack([c["projmatrix"] for c in cams]))
    cam = t(np.stack([c["campos"] for c in cams]))
    g = backend.backward_views(t(bg), t(sc["xyz"])[None], radii, t(colors_precomp)[None] if not use_sh else None, t(sc["opacities"]),
                               t(sc["scales"])[None] if use_sr else None, t(sc["rotations"])[None] if use_sr else None, 1.0,
                               t(cov3D_precomp)[None] if not use_sr else None, vm, pm, cam, None, cams[0]["tanfovx"],
                               cams[0]["tanfovy"], t(dpix), t(sc["shs"])[None] if use_sh else None, sh_degree, geom, n_total,
                               binning, img, True, views_per_set=V)
    P = sc["xyz"].shape[0]
    ref = {k: 0.0 for k in ("colors", "opacity", "means3D", "sh", "scales", "rotations")}
    for v, c in enumerate(cams):
        o = RasterOracle()
        kw = {}
        if not use_sh:
            kw.update(colors_precomp=colors_precomp, shs=None)
        if not use_sr:
            kw.update(cov3D_precomp=cov3D_precomp, scales=None, rotations=None)
        oracle_forward(o, sc, c, H, W, bg=bg, sh_degree=sh_degree, exp_mode=1, **kw)
        o.backward(dpix[v], accum64=True)
        close(g["means2D"][v].cpu().numpy(), o.get("dL_dmeans2D"), what=f"means2D view {v}")
        close(g["cov3D"][v].cpu().numpy(), o.get("dL_dcov3D"), what=f"cov3D view {v}")
        for k in ref:
            if k == "colors" and use_sh:
                continue
            ref[k] = ref[k] + o.get(NAMES[k]).astype(np.float64)
    close(g["opacity"].cpu().numpy().reshape(P, 1), ref["opacity"], what="opacity")
    close(g["means3D"][0].cpu().numpy(), ref["means3D"], what="means3D")
    if use_sh:
        close(g["sh"][0].cpu().numpy(), ref["sh"], what="sh")
    else:
        close(g["colors"][0].cpu().numpy(), ref["colors"], what="colors")
    if use_sr:
        close(g["scales"][0].cpu().numpy(), ref["scales"], what="scales")
        close(g["rotations"][0].cpu().numpy(), ref["rotations"], what="rotations")
    return g
