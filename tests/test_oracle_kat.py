"""Known-answer tests that pin the CPU oracle (the reference ships no
golden vectors for this path, so the oracle is pinned by closed forms and by an independent
float64 PyTorch restatement)."""
import math

import numpy as np
import pytest
import torch

from dgs_amd import cameras
from oracle.raster_oracle import RasterOracle, det_expf, mark_visible
import ref_torch_raster as R
from util_scene import small_scene, oracle_forward


def _front_cam(res):
    c2w = cameras.look_at_c2w((3.0, 0.0, 0.0))
    return cameras.camera_from_c2w(c2w, cameras.default_fxfycxcy(res), res, res)


def test_det_expf_accuracy_and_edges():
    xs = np.concatenate([np.linspace(-87, 20, 20001), -np.logspace(-8, 1.9, 500)]).astype(np.float32)
    worst = 0.0
    for x in xs:
        ref = math.exp(float(x))
        worst = max(worst, abs(det_expf(float(x)) - ref) / ref)
    assert worst < 2.0 * 2 ** -24          # < 2 ulp
    assert det_expf(0.0) == 1.0
    assert det_expf(-100.0) == 0.0
    assert det_expf(89.0) == float("inf")


def test_single_isotropic_gaussian_closed_form():
    # (i) one isotropic Gaussian at the origin seen from distance 3 with the principal point at the
    # image centre: mean2D = (cx-0.5, cy-0.5); cov2D = (fx*s/z)^2 + 0.3 on the diagonal.
    res, s, op = 64, 0.05, 0.8
    cam = _front_cam(res)
    sc = dict(xyz=np.zeros((1, 3), np.float32), shs=np.array([[[1.0, 0.2, -0.4]]], np.float32),
              scales=np.full((1, 3), s, np.float32), rotations=np.array([[1, 0, 0, 0]], np.float32),
              opacities=np.array([[op]], np.float32))
    o = RasterOracle()
    n = oracle_forward(o, sc, cam, res, res)
    img = o.get("out_color")
    fx = 1422.222 / 1024 * res
    var = (fx * s / 3.0) ** 2 + 0.3
    rgb = np.maximum(0.28209479177387814 * sc["shs"][0, 0] + 0.5, 0)
    # eigenvalue uses max(0.1, mid^2-det) (forward.cu:229-232): isotropic -> lambda = var + sqrt(0.1)
    radius = math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))
    assert o.get("radii")[0] == radius
    np.testing.assert_allclose(o.get("means2D")[0], [res / 2 - 0.5, res / 2 - 0.5], atol=1e-4)
    np.testing.assert_allclose(o.get("depths")[0], 3.0, rtol=1e-6)
    for (x, y) in [(31, 31), (32, 32), (35, 30), (28, 36), (40, 40)]:
        d2 = (x - (res / 2 - 0.5)) ** 2 + (y - (res / 2 - 0.5)) ** 2
        a = min(0.99, op * math.exp(-0.5 * d2 / var))
        exp_c = rgb * a + (1 - a) * 1.0 if a >= 1 / 255 else np.ones(3)
        np.testing.assert_allclose(img[:, y, x], exp_c, rtol=2e-5, atol=2e-6)
    # rect: centre 31.5 +- radius covers tiles 1..2 in both axes for this radius
    x0, x1 = int((31.5 - radius) / 16), int((31.5 + radius + 15) / 16)
    assert n == (x1 - x0) ** 2 == o.get("tiles_touched")[0]


def test_depth_order_swap_two_gaussians():
    # (ii) two overlapping Gaussians on the optical axis; swapping depths swaps compositing order.
    res = 32
    cam = _front_cam(res)

    def run(z_red, z_blue):
        sc = dict(xyz=np.array([[z_red, 0, 0], [z_blue, 0, 0]], np.float32),
                  shs=np.array([[[1.77, -1.77, -1.77]], [[-1.77, -1.77, 1.77]]], np.float32),
                  scales=np.full((2, 3), 0.08, np.float32), rotations=np.tile([1, 0, 0, 0], (2, 1)).astype(np.float32),
                  opacities=np.full((2, 1), 0.9, np.float32))
        o = RasterOracle()
        oracle_forward(o, sc, cam, res, res, bg=(0, 0, 0))
        return o.get("out_color")[:, 16, 16], o.get("point_list"), o.get("depths")

    c1, pl1, d1 = run(0.5, -0.5)     # camera at x=+3: larger x is nearer -> red in front
    c2, pl2, d2 = run(-0.5, 0.5)
    assert d1[0] < d1[1] and d2[0] > d2[1]
    assert c1[0] > c1[2] and c2[2] > c2[0]
    assert set(pl1[:2]) == {0, 1} and pl1[0] == 0 and pl2[0] == 1


def test_equal_depth_ties_broken_by_index():
    res = 32
    cam = _front_cam(res)
    P = 6
    sc = dict(xyz=np.zeros((P, 3), np.float32), shs=np.zeros((P, 1, 3), np.float32),
              scales=np.full((P, 3), 0.05, np.float32), rotations=np.tile([1, 0, 0, 0], (P, 1)).astype(np.float32),
              opacities=np.full((P, 1), 0.3, np.float32))
    o = RasterOracle()
    oracle_forward(o, sc, cam, res, res)
    pl, rg = o.get("point_list"), o.get("ranges")
    for t in range(rg.shape[0]):
        seg = pl[rg[t, 0]:rg[t, 1]]
        assert list(seg) == sorted(seg)


def test_empty_tile_is_background_and_p0():
    # (iii) + (iv)
    res = 48
    cam = _front_cam(res)
    sc = dict(xyz=np.array([[0, 0.9, 0.9]], np.float32), shs=np.zeros((1, 1, 3), np.float32),
              scales=np.full((1, 3), 0.01, np.float32), rotations=np.array([[1, 0, 0, 0]], np.float32),
              opacities=np.array([[0.9]], np.float32))
    o = RasterOracle()
    oracle_forward(o, sc, cam, res, res, bg=(0.25, 0.5, 0.75))
    img = o.get("out_color")
    rg = o.get("ranges")
    empty = [t for t in range(rg.shape[0]) if rg[t, 0] == rg[t, 1]]
    assert empty
    t = empty[-1]
    ty, tx = divmod(t, 3)
    np.testing.assert_array_equal(img[:, ty * 16, tx * 16], np.array([0.25, 0.5, 0.75], np.float32))
    assert (o.get("n_contrib").reshape(res, res)[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] == 0).all()
    o2 = RasterOracle()
    sc0 = dict(xyz=np.zeros((0, 3), np.float32), shs=np.zeros((0, 1, 3), np.float32), scales=np.zeros((0, 3), np.float32),
               rotations=np.zeros((0, 4), np.float32), opacities=np.zeros((0, 1), np.float32))
    assert oracle_forward(o2, sc0, cam, res, res) == 0
    assert (o2.get("out_color") == 0).all()      # rasterize_points.cu:68: zeros, not background


def test_near_plane_cull():
    # (vi) p_view.z <= 0.2 is culled (auxiliary.h:154)
    res = 32
    cam = _front_cam(res)
    xyz = np.array([[2.79, 0, 0], [2.81, 0, 0], [3.5, 0, 0], [2.0, 0, 0]], np.float32)   # depth .21 / .19 / -.5 / 1
    sc = dict(xyz=xyz, shs=np.zeros((4, 1, 3), np.float32), scales=np.full((4, 3), 0.01, np.float32),
              rotations=np.tile([1, 0, 0, 0], (4, 1)).astype(np.float32), opacities=np.full((4, 1), 0.5, np.float32))
    o = RasterOracle()
    oracle_forward(o, sc, cam, res, res)
    rad = o.get("radii")
    assert rad[0] > 0 and rad[1] == 0 and rad[2] == 0 and rad[3] > 0
    vis = mark_visible(xyz, cam["viewmatrix"], cam["projmatrix"])
    assert list(vis) == [True, False, False, True]


@pytest.mark.parametrize("deg,seed", [(0, 1), (1, 2), (3, 3)])
def test_forward_matches_float64_torch_restatement(deg, seed):
    H, W = 40, 56            # non-multiple of 16 in both axes, non-square
    sc, cams = small_scene(48, W, H, seed=seed, sh_degree=deg)
    cam = cams[0]
    o = RasterOracle()
    oracle_forward(o, sc, cam, H, W, bg=(0.3, 0.6, 0.9), sh_degree=deg)
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    ref = R.render(t(sc["xyz"]), t(sc["shs"]), t(sc["scales"]), t(sc["rotations"]), t(sc["opacities"]),
                   t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"]), cam["tanfovx"], cam["tanfovy"],
                   H, W, t([0.3, 0.6, 0.9]), sh_degree=deg)
    np.testing.assert_array_equal(o.get("radii"), ref["radii"].numpy())
    np.testing.assert_allclose(o.get("out_color"), ref["color"].numpy(), atol=3e-5)
    assert (o.get("n_contrib").reshape(H, W) != ref["n_contrib"].numpy()).mean() < 0.002
    # exp_mode 1 (deterministic exp) stays within float noise of libm
    o1 = RasterOracle()
    oracle_forward(o1, sc, cam, H, W, bg=(0.3, 0.6, 0.9), sh_degree=deg, exp_mode=1)
    np.testing.assert_allclose(o1.get("out_color"), o.get("out_color"), atol=2e-6)
    np.testing.assert_array_equal(o1.get("point_list"), o.get("point_list"))


def test_binning_invariants_ragged():
    H, W = 50, 70
    sc, cams = small_scene(300, W, H, seed=7, log_scale=-2.0)
    o = RasterOracle()
    n = oracle_forward(o, sc, cams[0], H, W)
    tt, po, pl, rg, keys = (o.get(k) for k in ("tiles_touched", "point_offsets", "point_list", "ranges", "keys"))
    assert n == tt.sum() == po[-1] == len(pl)
    np.testing.assert_array_equal(np.cumsum(tt), po)
    assert (np.diff(keys.astype(np.uint64)) >= 0).all()
    gx = (W + 15) // 16
    depths = o.get("depths")
    for t in range(rg.shape[0]):
        a, b = rg[t]
        assert ((keys[a:b] >> np.uint64(32)) == t).all()
        dd = depths[pl[a:b]]
        assert (np.diff(dd) >= 0).all()
    assert rg.shape[0] == gx * ((H + 15) // 16)
