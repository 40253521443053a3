"""Parity tests proper for the rasterizer backward on MI355X: gradients through the C ABI vs the oracle's
fp64-accumulated sums (the reference sums fp32 atomics in unspecified order, so the bar is relative: 2e-4 of the tensor's
max), drop-in binding autograd, and the batched Renderer path at BASELINE.json's 256^2 size."""
import numpy as np
import pytest
import torch

from dgs_amd import synth
from raster_bwd_util import assert_backward_parity
from util_scene import small_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def poisoned_lds():
    """Every test starts from LDS full of NaN patterns (dgs_debug_poison_lds): the blend kernels read ahead of their per-cell lists
    into record slots no entry was staged into, and the product-default arithmetic multiplies masked-out lanes by a zero weight --
    whatever an earlier kernel left in LDS must not be able to reach a pixel."""
    from dgs_amd.dit import DitOps
    DitOps().poison_lds()
    yield
DEV = torch.device("cuda:0")


def _backend():
    from dgs_amd.raster import default_backend
    return default_backend()


@pytest.mark.parametrize("deg,seed,H,W,views", [(0, 1, 40, 56, 1), (3, 3, 33, 17, 2), (1, 5, 64, 64, 3), (2, 6, 128, 96, 2)])
def test_small_scenes(deg, seed, H, W, views):
    sc, cams = small_scene(300, W, H, seed=seed, sh_degree=deg, n_views=views)
    assert_backward_parity(_backend(), sc, cams, H, W, DEV, sh_degree=deg, bg=(0.3, 0.6, 0.9), seed=seed)


@pytest.mark.parametrize("res,regime,views", [(64, "trained", 4), (64, "init", 2), (128, "trained", 2)])
def test_diffusiongs_shaped(res, regime, views):
    sc = synth.gaussian_scene(res, regime=regime, seed=0)
    cams, _, _ = synth.render_cameras(res, views, phase_deg=10)
    assert_backward_parity(_backend(), sc, cams, res, res, DEV)


@pytest.mark.parametrize("regime,views", [("trained", 2), ("init", 1)])
def test_full_size_256_vs_oracle(regime, views):
    """256^2, P = 262,146 (BASELINE configs[2]/[3]): all gradients against the oracle's fp64-accumulated sums, in the
    trained-like regime and in the regime the bench / a random-init training step renders."""
    sc = synth.gaussian_scene(256, regime=regime, seed=0)
    cams, _, _ = synth.render_cameras(256, 4, phase_deg=10)
    assert_backward_parity(_backend(), sc, cams[:views], 256, 256, DEV)


@pytest.mark.parametrize("regime,views,rtol", [("trained", 2, 2e-4), ("init", 1, 2e-4)])
def test_full_size_256_product_default_exp(regime, views, rtol):
    """The same with the product's blend exponential (`exact_exp` = 0: v_exp_f32 with its argument's rounding error compensated,
    <= 1.3 ulp measured on the device, and det_expf for a pair whose alpha falls within 1e-6 of the 1/255 cut-off: csrc/dgs_device.h
    `blend_exp`) at the SAME bar as the oracle-exponential build: SURVEY 8c's 1e-4 holds with room -- tools/raster_grad_error.py measures
    <= 5.3e-6 of a tensor's max in both scenes (profiles/r06_raster_grad_error.txt).  The reference's `exp(power)` (forward.cu:332-358,
    backward.cu:463-532) is CUDA's <= 2 ulp expf -- its setup.py passes no fast-math flag -- and round 5's bare v_exp_f32(power * log2e)
    (7.6 ulp at the cut-off) sat at 1.8e-3 here: pairs on the other side of the cut-off."""
    sc = synth.gaussian_scene(256, regime=regime, seed=0)
    cams, _, _ = synth.render_cameras(256, 4, phase_deg=10)
    assert_backward_parity(_backend(), sc, cams[:views], 256, 256, DEV, exact=False, rtol=rtol)


@pytest.mark.parametrize("exact,rtol", [(True, 2e-4), (False, 2e-4)], ids=["exact_exp", "product_default"])
def test_full_size_512_vs_oracle(exact, rtol):
    """BASELINE configs[4] (512^2, P = 1,048,578 Gaussians, 1,024 tiles, trained-like regime: N ~ 7 M instances, the list forms of
    the binning incl. the 1,024-thread per-tile sort): every gradient of one view against the oracle's fp64-accumulated sums
    (backward.cu:399-557 blend, :144-274,346-396 preprocess; rasterizer_impl.cu:340-434)."""
    sc = synth.gaussian_scene(512, regime="trained", seed=0)
    cams, _, _ = synth.render_cameras(512, 4, phase_deg=10)
    assert_backward_parity(_backend(), sc, cams[:1], 512, 512, DEV, exact=exact, rtol=rtol)


@pytest.mark.parametrize("regime", ["trained", "init"])
def test_backward_is_bit_reproducible_256(regime):
    """The deterministic form (`backend.deterministic = True`: every (tile, Gaussian) instance stores its sums into a slot of its own, a
    gather adds them in rectangle order, a tile's waves accumulate in LDS copies of their own; no floating-point atomic whose order
    is left to the hardware): two backward passes give identical bits, in the list forms (trained-like)
    and the on-demand scan form (random-init regime), 4 views at 256^2; the atomic form (reference's way) stays within 3e-5 of it."""
    from dgs_amd import cameras
    from dgs_amd.raster import RasterBackend, render_views_autograd
    res, V = 256, 4
    sc = synth.gaussian_scene(res, regime=regime, seed=1, activated=False)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=DEV)
    raw = [t(sc[k])[None] for k in ("xyz", "shs", "scales", "rotations", "opacities")]
    c2w = t(cameras.ring_cameras(V, phase_deg=10))[None]
    k = t(cameras.default_fxfycxcy(res)).expand(1, V, 4).contiguous()
    w = torch.randn(1, V, 3, res, res, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0)) / (3 * res * res)

    def grads(be):
        leaves = [x.clone().requires_grad_(True) for x in raw]
        render_views_autograd(be, *leaves, res, res, c2w, k).backward(w)
        return [x.grad.clone() for x in leaves]

    be = RasterBackend()
    be.deterministic = True
    runs = [grads(be) for _ in range(3)]
    assert be.last_backward_deterministic
    for r in runs[1:]:
        for a, b, name in zip(runs[0], r, ("xyz", "features", "scaling", "rotation", "opacity")):
            assert torch.equal(a, b), name
    at = RasterBackend()
    at.deterministic = False
    for a, b, name in zip(runs[0], grads(at), ("xyz", "features", "scaling", "rotation", "opacity")):
        # the atomic form's summation order is the hardware's: its distance from the fixed-order form varies run to run -- measured
        # 3e-6 ... 1.2e-5 of a tensor's max (profiles/r04_raster_deterministic_ab.txt), 1.08e-5 in a run of round 6 -- so the bar is 3e-5,
        # not the 1e-5 this line carried until then (a flaky pass)
        assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()) + 1e-12, name
    assert at.last_backward_deterministic is False


def test_precomputed_colors_and_long_lists():
    H, W = 32, 48
    sc, cams = small_scene(120, W, H, seed=8, n_views=2)
    cols = np.random.default_rng(1).uniform(0, 1, size=(120, 3)).astype(np.float32)
    assert_backward_parity(_backend(), sc, cams, H, W, DEV, colors_precomp=cols)
    sc, cams = small_scene(5000, 48, 48, seed=4, log_scale=-1.5)
    assert_backward_parity(_backend(), sc, cams, 48, 48, DEV)


def test_dropin_binding_autograd_vs_batched_renderer_256():
    """Full size (256^2, P = 262,146, 4 views): the batched fused-activation path and the reference call convention
    (torch activations + drop-in binding per view + torch autograd) must give the same image and gradients."""
    import diff_gaussian_rasterization as dgr
    from dgs_amd import cameras
    from dgs_amd.raster import render_views_autograd
    res, V = 256, 4
    sc = synth.gaussian_scene(res, regime="trained", seed=2, activated=False)
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=DEV)
    raw = [t(sc["xyz"])[None], t(sc["shs"])[None], t(sc["scales"])[None], t(sc["rotations"])[None], t(sc["opacities"])[None]]
    c2w = t(cameras.ring_cameras(V, phase_deg=10))[None]
    k = t(cameras.default_fxfycxcy(res)).expand(1, V, 4).contiguous()
    leaves = [x.clone().requires_grad_(True) for x in raw]
    img = render_views_autograd(_backend(), *leaves, res, res, c2w, k)
    g = torch.Generator(device=DEV).manual_seed(0)
    w = torch.randn(img.shape, generator=g, device=DEV) / img.numel()
    (img * w).sum().backward()
    assert all(torch.isfinite(x.grad).all() for x in leaves)
    ref = [x.clone().requires_grad_(True) for x in raw]
    view, proj, campos, tanfov = _backend().cameras_from_c2w(c2w, k, res, res)
    total = 0.0
    for v in range(V):
        rs = dgr.GaussianRasterizationSettings(res, res, float(tanfov[v, 0]), float(tanfov[v, 1]), torch.ones(3, device=DEV), 1.0,
                                               view[v], proj[v], 0, campos[v], False, False)
        x, f, s, r, o = (q[0] for q in ref)
        color, radii = dgr.GaussianRasterizer(rs)(x, torch.zeros_like(x, requires_grad=True), torch.sigmoid(o), shs=f,
                                                  scales=torch.exp(s), rotations=torch.nn.functional.normalize(r))
        mse = float(((color.detach() - img[0, v].detach()) ** 2).mean())
        assert mse < 1e-8, mse
        total = total + (color * w[0, v]).sum()
    total.backward()
    for a, b, name in zip(leaves, ref, ("xyz", "features", "scaling", "rotation", "opacity")):
        num = float((a.grad - b.grad).double().norm())
        den = float(b.grad.double().norm())
        assert num <= 2e-3 * den + 1e-12, (name, num, den)


@pytest.mark.parametrize("walk", [1, 2])
def test_both_walks_forced(walk):
    """Both walks of the blend backward (one / two pixels per lane; the library picks by grid size, raster_backward.hip `pair_walk`),
    forced by DGS_RASTER_BWD_WALK in a child process, on the parity cases of this file that finish in seconds."""
    import os, subprocess, sys
    env = dict(os.environ, DGS_RASTER_BWD_WALK=str(walk))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "small_scenes or diffusiongs_shaped or bit_reproducible or precomputed_colors"], env=env, capture_output=True,
                       text=True, timeout=1500, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
