"""Pins the oracle's backward (backward.cu restatement) against float64 autograd of the independent
PyTorch restatement, and against a central finite difference of the oracle's own forward."""
import numpy as np
import pytest
import torch

from oracle.raster_oracle import RasterOracle
import ref_torch_raster as R
from util_scene import small_scene, oracle_forward


def _autograd(sc, cam, H, W, deg, gpix, bg):
    t = lambda a, g=False: torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=g)
    xyz, shs, scales, rot, op = (t(sc[k], True) for k in ("xyz", "shs", "scales", "rotations", "opacities"))
    out = R.render(xyz, shs, scales, rot, op, t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"]),
                   cam["tanfovx"], cam["tanfovy"], H, W, t(bg), sh_degree=deg)
    (out["color"] * t(gpix)).sum().backward()
    return dict(dL_dmeans3D=xyz.grad.numpy(), dL_dsh=shs.grad.numpy(), dL_dscales=scales.grad.numpy(),
                dL_drotations=rot.grad.numpy(), dL_dopacity=op.grad.numpy())


def _relerr(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


@pytest.mark.parametrize("deg,seed", [(0, 11), (2, 12), (3, 13)])
def test_backward_matches_float64_autograd(deg, seed):
    H, W = 40, 56
    sc, cams = small_scene(40, W, H, seed=seed, sh_degree=deg)
    cam = cams[0]
    rng = np.random.default_rng(seed)
    gpix = rng.normal(size=(3, H, W)).astype(np.float32)
    bg = (0.3, 0.6, 0.9)
    o = RasterOracle()
    oracle_forward(o, sc, cam, H, W, bg=bg, sh_degree=deg)
    o.backward(gpix, accum64=True)
    ref = _autograd(sc, cam, H, W, deg, gpix, bg)
    for k, tol in [("dL_dsh", 2e-4), ("dL_dopacity", 2e-4), ("dL_dmeans3D", 2e-3), ("dL_dscales", 2e-3),
                   ("dL_drotations", 2e-3)]:
        assert _relerr(o.get(k), ref[k]) < tol, (k, _relerr(o.get(k), ref[k]))
    # float accumulation (reference behaviour) stays within float noise of the double accumulation
    o32 = RasterOracle()
    oracle_forward(o32, sc, cam, H, W, bg=bg, sh_degree=deg)
    o32.backward(gpix, accum64=False)
    for k in ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors", "dL_dmeans3D", "dL_dscales", "dL_drotations"):
        assert _relerr(o32.get(k), o.get(k)) < 1e-4, k


def test_backward_precomputed_cov_and_colors():
    H, W = 32, 48
    sc, cams = small_scene(24, W, H, seed=5)
    cam = cams[0]
    o = RasterOracle()
    oracle_forward(o, sc, cam, H, W)
    cov = o.get("cov3D").copy()
    rgb = o.get("rgb").copy()
    gpix = np.random.default_rng(1).normal(size=(3, H, W)).astype(np.float32)
    o.backward(gpix, accum64=True)
    o2 = RasterOracle()
    o2.forward(np.ones(3, np.float32), sc["xyz"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"], cam["campos"],
               cam["tanfovx"], cam["tanfovy"], H, W, colors_precomp=rgb, cov3D_precomp=cov)
    np.testing.assert_array_equal(o2.get("out_color"), o.get("out_color"))
    o2.backward(gpix, accum64=True)
    np.testing.assert_array_equal(o2.get("dL_dcolors"), o.get("dL_dcolors"))
    np.testing.assert_array_equal(o2.get("dL_dcov3D"), o.get("dL_dcov3D"))
    assert (o2.get("dL_dscales") == 0).all() and (o2.get("dL_dsh").size == 0)


def test_backward_finite_difference_opacity_and_color():
    # central differences through the oracle's own float32 forward (loose: float32 forward noise)
    H, W = 32, 32
    sc, cams = small_scene(12, W, H, seed=21, log_scale=-2.2)
    cam = cams[0]
    gpix = np.random.default_rng(2).normal(size=(3, H, W)).astype(np.float32)
    o = RasterOracle()
    oracle_forward(o, sc, cam, H, W)
    o.backward(gpix, accum64=True)
    g_op, g_sh = o.get("dL_dopacity"), o.get("dL_dsh")

    def loss(scn):
        oo = RasterOracle()
        oracle_forward(oo, scn, cam, H, W)
        return float((oo.get("out_color").astype(np.float64) * gpix).sum())

    eps = 2e-3
    for i in range(4):
        for key, grad, idx in (("opacities", g_op, (i, 0)), ("shs", g_sh, (i, 0, 1))):
            p, m = {k: v.copy() for k, v in sc.items()}, {k: v.copy() for k, v in sc.items()}
            p[key][idx] += eps
            m[key][idx] -= eps
            fd = (loss(p) - loss(m)) / (2 * eps)
            assert abs(fd - grad[idx]) < 2e-2 * max(1.0, abs(fd)), (key, idx, fd, grad[idx])
