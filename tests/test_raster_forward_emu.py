"""Kernel-logic parity on CPU: csrc/raster_forward.hip compiled against tests/hipemu and compared bit for bit
with the oracle.  (The same assertions run on the real gfx950 build in tests/test_raster_forward_gpu.py.)"""
import numpy as np
import pytest
import torch

from dgs_amd import synth
from emu_util import emu_backend
from parity_util import assert_forward_parity
from util_scene import small_scene

CPU = torch.device("cpu")


@pytest.fixture(params=["scan", "sort", "bitonic"], autouse=True)
def binning_form(request, monkeypatch):
    """Both binning forms of raster_forward.hip (per-tile scan of the depth-ordered Gaussians / instance list + per-tile bitmap
    sort / instance list + per-tile bitonic sort in LDS) must give the reference's per-tile lists: every test runs with each forced
    (DGS_RASTER_BIN -> DgsRasterForwardArgs.binning_form)."""
    monkeypatch.setenv("DGS_RASTER_BIN", {"sort": "1", "scan": "2", "bitonic": "3"}[request.param])


@pytest.mark.parametrize("deg,seed,H,W", [(0, 1, 40, 56), (3, 3, 33, 17), (1, 5, 64, 64)])
def test_small_scenes_bit_exact(deg, seed, H, W):
    sc, cams = small_scene(200, W, H, seed=seed, sh_degree=deg, n_views=2)
    assert_forward_parity(emu_backend(), sc, cams, H, W, CPU, bg=(0.3, 0.6, 0.9), sh_degree=deg)


def test_product_default_hardware_exp_on_the_emulator():
    """`exact_exp` = 0 (what the product runs): integer artefacts and per-Gaussian state bit-exact, colours within 1e-5 (on the
    emulator the `hardware` exponential is libm's exp2f: this checks the plumbing of the mode, the GPU test checks v_exp_f32)."""
    res = 64
    sc = synth.gaussian_scene(res, regime="trained", seed=0)
    cams, _, _ = synth.render_cameras(res, 2, phase_deg=10)
    assert_forward_parity(emu_backend(), sc, cams, res, res, CPU, exact=False)


def test_diffusiongs_shaped_64_multi_view():
    res = 64
    sc = synth.gaussian_scene(res, regime="trained", seed=0)
    cams, _, _ = synth.render_cameras(res, 3, phase_deg=10)
    scn = dict(xyz=sc["xyz"], shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], opacities=sc["opacities"])
    assert_forward_parity(emu_backend(), scn, cams, res, res, CPU)


def test_many_instances_per_tile_and_ties():
    # big Gaussians: every one touches every tile; duplicated depths exercise the stable tie-break
    H, W = 48, 48
    sc, cams = small_scene(5000, W, H, seed=9, log_scale=-1.2)
    sc["xyz"][100:200] = sc["xyz"][100]        # 100 identical positions -> identical depth keys
    assert_forward_parity(emu_backend(), sc, cams, H, W, CPU)


def test_ties_inside_a_bucket_below_the_network_limit():
    """30 identical depth keys land in ONE bucket of the per-tile LDS sort and stay below its limit (40: above it the tile takes the
    bitonic network, which the test above exercises with 100): the place of a key inside its bucket comes from counting the
    bucket's smaller 64-bit keys, so equal depth bits are ordered by Gaussian index like the reference's stable sort."""
    H, W = 48, 48
    sc, cams = small_scene(3000, W, H, seed=11, log_scale=-1.2)
    sc["xyz"][300:330] = sc["xyz"][300]
    assert_forward_parity(emu_backend(), sc, cams, H, W, CPU)


def clustered_scene(P, n_front, res):
    """n_front faint Gaussians stacked on the first tile, nearest to the camera, the rest tiny and elsewhere: the first tile's
    list is a fraction of P but its entries are the first depth ranks (the scan form's window is sized from the average
    density and has to shrink: scan_more), and no pixel saturates, so the whole list is walked."""
    from dgs_amd import cameras
    sc, cams = small_scene(P, res, res, seed=21, log_scale=-6.5, spread=0.5)
    c2w = cameras.ring_cameras(1, phase_deg=25.0)[0]
    ro, rd = cameras.pixel_rays(c2w, cameras.default_fxfycxcy(res, res), res, res)
    o, d = np.asarray(ro).reshape(res, res, 3)[8, 8], np.asarray(rd).reshape(res, res, 3)[8, 8]
    rng = np.random.default_rng(5)
    sc["xyz"][:n_front] = (o + d * (1.6 + 0.2 * rng.uniform(size=(n_front, 1))) + rng.normal(0, 0.01, size=(n_front, 3))).astype(np.float32)
    sc["scales"][:n_front] = np.exp(-2.6).astype(np.float32)
    sc["opacities"][:n_front] = 0.02
    return sc, cams


def test_scan_window_shrinks_on_depth_clusters():
    res = 48
    sc, cams = clustered_scene(20000, 3000, res)
    assert_forward_parity(emu_backend(), sc, cams, res, res, CPU)


def test_precomputed_inputs_and_two_sets():
    H, W = 32, 48
    a, cams = small_scene(64, W, H, seed=2, n_views=4)
    b, _ = small_scene(64, W, H, seed=3)
    sc = {k: np.stack([a[k], b[k]]) for k in a}
    assert_forward_parity(emu_backend(), sc, cams, H, W, CPU, views_per_set=2)
    rng = np.random.default_rng(0)
    cols = rng.uniform(0, 1, size=(2, 64, 3)).astype(np.float32)
    assert_forward_parity(emu_backend(), sc, cams, H, W, CPU, views_per_set=2, colors_precomp=cols)


def test_empty_and_culled():
    H, W = 32, 32
    sc, cams = small_scene(8, W, H, seed=4)
    sc["xyz"][:] = np.array([10.0, 10.0, 10.0], np.float32)      # everything behind / outside
    out = assert_forward_parity(emu_backend(), sc, cams, H, W, CPU, bg=(0.1, 0.2, 0.3))
    assert out[0] == 0
    be = emu_backend()
    z = torch.zeros
    n, color, radii, *_ = be.rasterize_gaussians(z(3), z(0, 3), z(0), z(0, 1), z(0, 3), z(0, 4), 1.0, z(0), torch.eye(4),
                                                 torch.eye(4), 1.0, 1.0, H, W, z(0, 1, 3), 0, z(3), False, False)
    assert n == 0 and color.shape == (3, H, W) and float(color.abs().sum()) == 0.0


def test_more_gaussians_than_one_bitmap_window():
    # P > 16128*32 ranks -> tile_sort_kernel walks two bitmap windows; also > 1 radix block column
    H, W = 32, 32
    P = 16128 * 32 + 5000
    sc, cams = small_scene(P, W, H, seed=12, log_scale=-6.0, spread=0.8)
    assert_forward_parity(emu_backend(), sc, cams, H, W, CPU)


def test_camera_and_ray_kernels_match_oracle():
    """`Camera` (gs_core.py:277-316) and `TransformInput` (systems/utils.py:621-757) as single launches vs the oracle's
    torch restatements (the ray restatement is itself pinned against the reference's own function in test_dit_oracle.py)."""
    import torch
    from dgs_amd import cameras
    from oracle import dit_oracle as D
    be = emu_backend()
    B, V, H, W = 2, 3, 24, 40
    c2w = torch.tensor(np.stack([cameras.ring_cameras(V, phase_deg=21.0 * b, radius=2.5 + b) for b in range(B)]))
    k = torch.tensor(cameras.default_fxfycxcy(W, H)).expand(B, V, 4).contiguous() * torch.tensor([1.0, 1.1, 0.97, 1.02])
    ro, rd = be.rays_from_c2w(c2w, k, H, W)
    ro_ref, rd_ref = D.transform_input_rays(c2w, k, H, W)
    np.testing.assert_allclose(ro.numpy(), ro_ref.numpy(), atol=1e-6)
    np.testing.assert_allclose(rd.numpy(), rd_ref.numpy(), atol=2e-6)
    view, proj, campos, tanfov = be.cameras_from_c2w(c2w, k, H, W)
    v_ref, p_ref, c_ref, t_ref = D.camera_matrices(c2w.reshape(-1, 4, 4), k.reshape(-1, 4), H, W)
    np.testing.assert_allclose(view.numpy(), v_ref.numpy(), atol=2e-6)
    np.testing.assert_allclose(proj.numpy(), p_ref.numpy(), atol=2e-5)
    np.testing.assert_allclose(campos.numpy(), c_ref.numpy(), atol=0)
    np.testing.assert_allclose(tanfov.numpy(), t_ref.numpy(), rtol=1e-6)


def test_depth_range_sort_pile_up_beyond_the_lds():
    """The depth range sort (raster_forward.hip range_*_kernel): 9,000 of 9,300 Gaussians share ONE depth key -- one coarse bucket,
    more pairs than a workgroup of the final sort holds in LDS (kRangeSortCap = 8,192): the chunked path (chunks sorted in LDS, merged by
    rank) must still give the reference's order, ties by Gaussian index."""
    H, W = 32, 32
    sc, cams = small_scene(9300, W, H, seed=21, log_scale=-1.6, n_views=1)
    cam = cams[0]
    # the same view-space depth for the first 9,000: move them along the view's z axis onto the plane tz = 3
    vm = np.asarray(cam["viewmatrix"], np.float64)             # row-vector convention: p_view = p @ vm[:3, :3] + vm[3, :3]
    xyz = sc["xyz"].astype(np.float64)
    tz = xyz @ vm[:3, 2] + vm[3, 2]
    xyz[:9000] += np.outer(3.0 - tz[:9000], vm[:3, 2]) / float(vm[:3, 2] @ vm[:3, 2])
    sc["xyz"] = xyz.astype(np.float32)
    assert_forward_parity(emu_backend(), sc, cams, H, W, CPU)


def test_depth_range_sort_nothing_visible_and_one_visible():
    """No key but the culled key (every Gaussian behind the camera), then exactly one visible Gaussian: the range map has nothing /
    a single point to span."""
    H, W = 32, 32
    sc, cams = small_scene(300, W, H, seed=22, n_views=1)
    cam = cams[0]
    vm = np.asarray(cam["viewmatrix"], np.float64)
    xyz = sc["xyz"].astype(np.float64)
    tz = xyz @ vm[:3, 2] + vm[3, 2]
    behind = xyz + np.outer(-5.0 - tz, vm[:3, 2]) / float(vm[:3, 2] @ vm[:3, 2])      # tz = -5 for everyone
    sc["xyz"] = behind.astype(np.float32)
    assert_forward_parity(emu_backend(), sc, cams, H, W, CPU)
    one = behind.copy()
    one[17] = xyz[17] + (2.5 - tz[17]) * vm[:3, 2] / float(vm[:3, 2] @ vm[:3, 2])
    sc["xyz"] = one.astype(np.float32)
    assert_forward_parity(emu_backend(), sc, cams, H, W, CPU)


def test_radix_sort_gives_the_same_lists(monkeypatch):
    """DGS_RASTER_SORT=radix (the four-pass radix sort the range sort replaced as the default; read once per process: child process)."""
    import os, subprocess, sys
    if os.environ.get("DGS_RASTER_BIN") != "2":
        pytest.skip("one run is enough: the child process runs every binning form")
    env = dict(os.environ, DGS_RASTER_SORT="radix")
    env.pop("DGS_RASTER_BIN", None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "small_scenes or many_instances or pile_up"], env=env, capture_output=True, text=True, timeout=1500,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
