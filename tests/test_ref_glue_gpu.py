"""The reference's OWN render glue -- `Camera`, `GaussianModel.set_data / get_*`, `render_opencv_cam`, `DeferredGaussianRender`,
`deferred_gaussian_render`, `Renderer.forward` (gs_core.py:277-316, 321-575, 874-1064; renderer.py:20-92), byte-for-byte copies under
oracle/_ref/py (oracle/build_ref.py, oracle/ref_glue.py) -- running UNCHANGED on top of the drop-in `diff_gaussian_rasterization`
package, forward and `.backward()`, against the product's batched `Renderer.forward` (one launch sequence, fused activations).
Plus the edge cases of the C ABI on the real gfx950 build (the CPU suite runs them on the emulator build only)."""
import numpy as np
import pytest
import torch

from dgs_amd import cameras, synth

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
NAMES = ("xyz", "features", "scaling", "rotation", "opacity")


def _case(res, V, B, seed):
    t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=DEV)
    raws = []
    for b in range(B):
        sc = synth.gaussian_scene(res, regime="trained", seed=seed + b, activated=False)
        raws.append([t(sc["xyz"]), t(sc["shs"]), t(sc["scales"]), t(sc["rotations"]), t(sc["opacities"]).reshape(-1, 1)])
    raw = [torch.stack([r[i] for r in raws]) for i in range(5)]
    c2w = torch.stack([t(cameras.ring_cameras(V, phase_deg=10.0 + 13.0 * b)) for b in range(B)])
    k = t(cameras.default_fxfycxcy(res)).expand(B, V, 4).contiguous()
    return raw, c2w, k


def _psnr(a, b):
    mse = float(((a.double().clamp(0, 1) - b.double().clamp(0, 1)) ** 2).mean())
    return 200.0 if mse == 0 else -10.0 * np.log10(mse)


@pytest.fixture(scope="module")
def ref():
    from oracle import ref_glue
    if not ref_glue.available():
        pytest.skip("oracle/_ref/py missing: run oracle/build_ref.py where /root/reference exists (it travels with the snapshot)")
    return ref_glue.load()


def test_reference_renderer_unchanged_over_the_dropin_256(ref):
    """256^2, P = 262,146 per sample, B = 2 samples x 4 views: the reference's Renderer.forward (deferred=True -> its
    DeferredGaussianRender autograd Function: b*v per-view render_opencv_cam calls, recomputed in backward) over the drop-in vs
    the product's Renderer.forward: image PSNR >= 80 dB, the five gradients within 2e-3 relative."""
    from dgs_amd import denoiser as dn
    res, V, B = 256, 4, 2
    raw, c2w, k = _case(res, V, B, seed=2)
    cfg = dn.AttrDict(gaussians_sh_degree=0, use_gssplat=False)
    theirs, ours = ref.Renderer(cfg), dn.Renderer(cfg)
    g = torch.Generator(device=DEV).manual_seed(0)
    w = torch.randn(B, V, 3, res, res, generator=g, device=DEV) / (B * V * 3 * res * res)
    la = [x.clone().requires_grad_(True) for x in raw]
    img_a = theirs(*la, res, res, c2w, k)                                  # the reference's code path
    (img_a * w).sum().backward()
    lb = [x.clone().requires_grad_(True) for x in raw]
    img_b = ours(*lb, res, res, c2w, k)
    (img_b * w).sum().backward()
    assert img_a.shape == img_b.shape == (B, V, 3, res, res)
    assert _psnr(img_a.detach(), img_b.detach()) >= 80.0, _psnr(img_a.detach(), img_b.detach())
    for a, b, name in zip(la, lb, NAMES):
        assert a.grad is not None and torch.isfinite(a.grad).all(), name
        num, den = float((a.grad - b.grad).double().norm()), float(a.grad.double().norm())
        assert num <= 2e-3 * den + 1e-12, (name, num, den)
    # the non-deferred branch of the reference's Renderer.forward (plain autograd through render_opencv_cam), forward only
    with torch.no_grad():
        img_c = theirs(*raw, res, res, c2w, k, deferred=False)
    assert _psnr(img_c, img_b.detach()) >= 80.0


def test_reference_camera_and_gaussian_model_against_the_product(ref):
    """The reference's Camera and GaussianModel activations vs the product's camera kernel and fused activations' oracle twins."""
    from diffusionGS.models.gsrenderer import gs_core
    from dgs_amd.raster import default_backend
    res, V = 64, 3
    raw, c2w, k = _case(res, V, 1, seed=5)
    view, proj, campos, tanfov = default_backend().cameras_from_c2w(c2w, k, res, res)
    for v in range(V):
        cam = gs_core.Camera(C2W=c2w[0, v], fxfycxcy=k[0, v], h=res, w=res)
        np.testing.assert_allclose(view[v].cpu().numpy(), cam.world_view_transform.cpu().numpy(), atol=2e-6)
        np.testing.assert_allclose(proj[v].cpu().numpy(), cam.full_proj_transform.cpu().numpy(), atol=2e-5)
        np.testing.assert_allclose(campos[v].cpu().numpy(), cam.camera_center.cpu().numpy(), atol=1e-6)
        assert abs(float(tanfov[v, 0]) - float(cam.tanfovX)) < 1e-6 and abs(float(tanfov[v, 1]) - float(cam.tanfovY)) < 1e-6
    pc = gs_core.GaussianModel(0, None).set_data(*(x[0] for x in raw))
    from dgs_amd import denoiser as dn
    mine = dn.GaussianModel(0, None).set_data(*(x[0] for x in raw))
    for prop in ("get_xyz", "get_scaling", "get_rotation", "get_opacity", "get_features"):
        assert torch.equal(getattr(pc, prop), getattr(mine, prop)), prop


# ---- edge cases of the C ABI on the gfx950 build (GPU twins of tests/test_raster_forward_emu.py / test_oracle_kat.py) ----

def _be():
    from dgs_amd.raster import default_backend
    return default_backend()


@pytest.mark.parametrize("form", ["1", "2", "3", "0"])
def test_empty_scene_all_culled_and_near_plane_on_gpu(form, monkeypatch):
    from parity_util import assert_forward_parity
    from util_scene import small_scene
    monkeypatch.setenv("DGS_RASTER_BIN", form)
    H, W = 32, 32
    sc, cams = small_scene(8, W, H, seed=4)
    sc["xyz"][:] = np.array([10.0, 10.0, 10.0], np.float32)                # everything behind / outside: no instance at all
    for exact in (True, False):
        out = assert_forward_parity(_be(), sc, cams, H, W, DEV, bg=(0.1, 0.2, 0.3), exact=exact)
        assert out[0] == 0
    # P == 0 (rasterize_points.cu:68: outputs stay zero-initialised, not background)
    z = lambda *s: torch.zeros(*s, device=DEV)
    n, color, radii, *_ = _be().rasterize_gaussians(z(3), z(0, 3), z(0), z(0, 1), z(0, 3), z(0, 4), 1.0, z(0), torch.eye(4, device=DEV),
                                                    torch.eye(4, device=DEV), 1.0, 1.0, H, W, z(0, 1, 3), 0, z(3), False, False)
    assert n == 0 and color.shape == (3, H, W) and float(color.abs().sum()) == 0.0
    # a mixed scene: zero 3D covariance (only the 0.3 px dilation remains), Gaussians AT the first camera (near-plane cull,
    # auxiliary.h:154, in that view only) and just in front of / behind its near plane
    sc, cams = small_scene(64, W, H, seed=6, n_views=2)
    sc["scales"][:4] = 0.0
    eye = np.asarray(cams[0]["campos"], np.float32)
    fwd = -eye / np.linalg.norm(eye)
    sc["xyz"][4:8] = eye
    sc["xyz"][8] = eye + 0.19 * fwd
    sc["xyz"][9] = eye + 0.21 * fwd
    sc["xyz"][10] = eye - 0.5 * fwd
    assert_forward_parity(_be(), sc, cams, H, W, DEV, exact=True)
    assert_forward_parity(_be(), sc, cams, H, W, DEV, exact=False)


def test_prefiltered_with_a_culled_gaussian_is_reported_on_gpu():
    """prefiltered=True promises that no Gaussian is culled (forward.cu:182-187 traps); the C ABI reports it as a status."""
    from parity_util import _run_backend_forward
    from util_scene import small_scene
    H, W = 32, 32
    sc, cams = small_scene(16, W, H, seed=7)
    sc["xyz"][3] = np.array([50.0, 50.0, 50.0], np.float32)
    be = _be()
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=DEV)
    vm, pm, cam = t(cams[0]["viewmatrix"])[None], t(cams[0]["projmatrix"])[None], t(cams[0]["campos"])[None]
    with pytest.raises(RuntimeError, match="prefiltered"):
        be.forward_views(t(np.ones(3)), t(sc["xyz"])[None], None, t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), 1.0, None, vm, pm,
                         cam, None, cams[0]["tanfovx"], cams[0]["tanfovy"], H, W, t(sc["shs"]), 0, True, True, views_per_set=1)
    # the same scene without the promise renders
    _run_backend_forward(be, sc, cams[:1], H, W, DEV, (1.0, 1.0, 1.0), 0, None, None, None, True)
