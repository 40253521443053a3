"""Rasterizer backward kernels on the CPU emulator vs the oracle's fp64-accumulated gradients (logic of the DPP wave
reduction, contributor bookkeeping, per-set accumulation over views, SH / covariance chains)."""
import numpy as np
import pytest
import torch

from emu_util import emu_backend
from raster_bwd_util import assert_backward_parity
from util_scene import small_scene

DEV = torch.device("cpu")


@pytest.fixture(params=["atomic", "deterministic"], autouse=True)
def backward_form(request):
    """Both forms of the backward (dgs_raster.h `scratch`): one fp32 atomic per value per (tile, Gaussian), or per-instance slots + an
    ordered gather (no floating-point atomic).  Every test of this file runs with each."""
    be = emu_backend()
    old = be.deterministic
    be.deterministic = request.param == "deterministic"
    be.last_backward_deterministic = None
    yield request.param
    assert be.last_backward_deterministic in (None, request.param == "deterministic")
    be.deterministic = old


@pytest.mark.parametrize("deg,seed,H,W,views", [(0, 1, 40, 56, 1), (3, 3, 33, 17, 2), (1, 5, 48, 48, 3)])
def test_backward_matches_oracle(deg, seed, H, W, views):
    sc, cams = small_scene(200, W, H, seed=seed, sh_degree=deg, n_views=views)
    assert_backward_parity(emu_backend(), sc, cams, H, W, DEV, sh_degree=deg, bg=(0.3, 0.6, 0.9), seed=seed)


def test_backward_product_default_arithmetic():
    """`exact_exp` = 0 (what the product runs): compensated hardware exponential + the cut-off guard band, one reciprocal for both
    divisions by (1 - alpha) -- the same gradients at the exact mode's bar, 2e-4 of each tensor's max (on the emulator v_exp_f32 and
    v_rcp_f32 are libm's exp2f and a division: this checks the mode's plumbing and the guard band's logic; the GPU suite measures)."""
    sc, cams = small_scene(200, 48, 48, seed=5, sh_degree=1, n_views=2)
    assert_backward_parity(emu_backend(), sc, cams, 48, 48, DEV, sh_degree=1, exact=False, rtol=2e-4)


def test_backward_precomputed_inputs():
    H, W = 32, 48
    sc, cams = small_scene(120, W, H, seed=8, n_views=2)
    cols = np.random.default_rng(1).uniform(0, 1, size=(120, 3)).astype(np.float32)
    assert_backward_parity(emu_backend(), sc, cams, H, W, DEV, colors_precomp=cols)


def test_backward_long_lists_many_rounds():
    H, W = 32, 32
    sc, cams = small_scene(1500, W, H, seed=4, log_scale=-1.5)      # several 256-entry rounds per tile
    assert_backward_parity(emu_backend(), sc, cams, H, W, DEV)


def test_batched_autograd_matches_dropin_binding_with_torch_activations():
    """Renderer path (raw parameters, activations + their Jacobians fused in the kernels, all views in one call) vs the
    reference call convention: torch exp / normalize / sigmoid + the drop-in `diff_gaussian_rasterization` binding once
    per view, gradients by torch autograd (gs_core.py:874-945,330-334)."""
    import dgs_amd.raster as R
    R._default = emu_backend()                       # the binding's `_C` resolves the backend lazily
    import diff_gaussian_rasterization as dgr
    from dgs_amd import cameras
    from oracle import dit_oracle as D
    H = W = 32
    B, V, P = 2, 2, 150
    g = torch.Generator().manual_seed(0)
    xyz = (torch.rand(B, P, 3, generator=g) - 0.5) * 1.2
    feats = torch.rand(B, P, 1, 3, generator=g) * 3 - 1.5
    scal = torch.randn(B, P, 3, generator=g) * 0.4 - 2.6
    rot = torch.randn(B, P, 4, generator=g)
    opa = torch.randn(B, P, 1, generator=g)
    c2w = torch.tensor(np.stack([cameras.ring_cameras(V, phase_deg=30.0 * b) for b in range(B)]))
    k = torch.tensor(cameras.default_fxfycxcy(W, H)).expand(B, V, 4).contiguous()
    leaves = [t.clone().requires_grad_(True) for t in (xyz, feats, scal, rot, opa)]
    img = R.render_views_autograd(emu_backend(), *leaves, H, W, c2w, k)
    w = torch.randn(img.shape, generator=g) / img.numel()
    (img * w).sum().backward()
    ref_leaves = [t.clone().requires_grad_(True) for t in (xyz, feats, scal, rot, opa)]
    view, proj, campos, tanfov = D.camera_matrices(c2w, k, H, W)
    total = 0.0
    for b in range(B):
        for v in range(V):
            rs = dgr.GaussianRasterizationSettings(H, W, float(tanfov[b, v, 0]), float(tanfov[b, v, 1]), torch.ones(3), 1.0,
                                                   view[b, v], proj[b, v], 0, campos[b, v], False, False)
            x, f, s, r, o = (t[b] for t in ref_leaves)
            color, _ = dgr.GaussianRasterizer(rs)(x, torch.zeros_like(x, requires_grad=True), torch.sigmoid(o), shs=f,
                                                  scales=torch.exp(s), rotations=torch.nn.functional.normalize(r))
            assert float((color.detach() - img[b, v].detach()).abs().max()) < 2e-4
            total = total + (color * w[b, v]).sum()
    total.backward()
    for a, r_, name in zip(leaves, ref_leaves, ("xyz", "features", "scaling", "rotation", "opacity")):
        scale = float(r_.grad.abs().max())
        assert float((a.grad - r_.grad).abs().max()) <= 1e-3 * scale + 1e-9, name


@pytest.mark.parametrize("walk", [1, 2])
def test_both_walks_forced(walk, backward_form):
    """The library picks the walk of the blend backward by grid size (raster_backward.hip `pair_walk`: one pixel per lane below 3,072
    workgroups, two from there on; the deterministic form always two) -- the small scenes above only ever reach one of them per form.
    DGS_RASTER_BWD_WALK forces either; it is read once per process, hence the child."""
    import os, subprocess, sys
    if backward_form != "atomic":
        pytest.skip("the child runs both forms itself")
    env = dict(os.environ, DGS_RASTER_BWD_WALK=str(walk))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "matches_oracle or long_lists or precomputed"], env=env, capture_output=True, text=True, timeout=1200,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
