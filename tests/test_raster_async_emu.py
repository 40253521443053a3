"""The product's render path without a host synchronisation (dgs_amd/raster.py `_AsyncPlan`, dgs_raster.h async mode), on the CPU
emulation of the kernels: the reference blocks on `num_rendered` in every forward (rasterizer_impl.cu:281); here the first call of a
shape learns the instance statistics and every later one runs from the plan -- same images and gradients bit for bit, a forced
ordering form from the previous call's statistics -- and a scene that outgrows what the plan provided is still rendered, as the
reference renders it (rasterize_points.cu:27-33 resizes by callback): a list longer than the LDS sort's LDS takes that kernel's chunked
path, a binning buffer below the worst case is verified and the render repeated with a buffer that fits."""
import numpy as np
import pytest
import torch

from dgs_amd import cameras, synth
from dgs_amd.raster import RasterBackend, _AsyncPlan, render_views_autograd
from emu_util import emu_lib

CPU = torch.device("cpu")


def _scene(res, regime, seed, n_views=2):
    sc = synth.gaussian_scene(res, regime=regime, seed=seed, activated=False)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32)
    raw = [t(sc[k])[None] for k in ("xyz", "shs", "scales", "rotations", "opacities")]
    c2w = t(cameras.ring_cameras(n_views, phase_deg=10))[None]
    k = t(cameras.default_fxfycxcy(res)).expand(1, n_views, 4).contiguous()
    return raw, c2w, k


@pytest.mark.parametrize("regime", ["small", "init"])
def test_planned_calls_equal_the_synchronous_call(regime):
    res = 64
    be = RasterBackend(lib=emu_lib())
    raw, c2w, k = _scene(res, regime, 0)
    first = be.render_views(*raw, res, res, c2w, k)                  # learns the statistics (synchronous form)
    plan = be.plan_for(raw[0].shape[1], res, res, 2, 2, CPU)
    assert plan.calls == {"sync": 1, "async": 0, "healed": 0} and plan.capacity >= 2 * plan.seen_max > 0 and plan.form in (1, 2, 3)
    assert plan.form == (2 if regime == "init" else 3)              # dense scenes scan, sparse ones sort their lists in LDS
    second = be.render_views(*raw, res, res, c2w, k)                 # runs from the plan: nothing read back
    third = be.render_views(*raw, res, res, c2w, k)
    assert plan.calls == {"sync": 1, "async": 2, "healed": 0}
    # forward-only renders size the buffer for the worst case of the shape when the budget allows it: such a plan cannot overflow
    assert plan.capacity_for(True) == plan.worst == 2 * 16 * raw[0].shape[1] and not plan.at_risk(plan.capacity_for(True))
    assert torch.equal(first, second) and torch.equal(first, third)
    be.check_async()


def test_planned_autograd_equals_synchronous_gradients():
    res = 64
    raw, c2w, k = _scene(res, "trained", 3)
    w = torch.randn(1, 2, 3, res, res, generator=torch.Generator().manual_seed(0)) / (3 * res * res)
    grads = []
    be = RasterBackend(lib=emu_lib())
    for _ in range(2):                                               # call 1: synchronous form; call 2: from the plan
        leaves = [x.clone().requires_grad_(True) for x in raw]
        img = render_views_autograd(be, *leaves, res, res, c2w, k)
        (img * w).sum().backward()
        grads.append([x.grad.clone() for x in leaves] + [img.detach()])
    assert be.plan_for(raw[0].shape[1], res, res, 2, 2, CPU).calls == {"sync": 1, "async": 1, "healed": 0}
    for a, b in zip(*grads):
        # the backward sums a Gaussian's tiles with fp32 atomics; the emulator runs workgroups in a fixed order
        assert torch.equal(a, b)


def test_stale_form_is_still_correct_and_refreshed():
    """The plan's form comes from the PREVIOUS call's statistics.  Every form is valid for any scene: a sparse scene rendered with the
    dense scene's form gives the same bits, and the call after it has the right form again.  The per-tile LDS sort is launched
    ALONE, sized for 1.5 x the longest list the plan has seen (no radix sort, no fallback kernels in the sequence): lists that outgrow
    its LDS are sorted in chunks and merged by rank inside the same kernel -- same bits, no error."""
    res = 64
    be = RasterBackend(lib=emu_lib())
    dense, c2w, k = _scene(res, "init", 1)
    sparse, _, _ = _scene(res, "small", 1)
    ref_sparse = RasterBackend(lib=emu_lib()).render_views(*sparse, res, res, c2w, k)
    ref_dense = RasterBackend(lib=emu_lib()).render_views(*dense, res, res, c2w, k)
    plan = be.plan_for(dense[0].shape[1], res, res, 2, 2, CPU)
    be.render_views(*dense, res, res, c2w, k)
    assert plan.form == 2
    assert torch.equal(be.render_views(*sparse, res, res, c2w, k), ref_sparse)      # scan form on a sparse scene
    assert torch.equal(be.render_views(*sparse, res, res, c2w, k), ref_sparse)
    assert plan.form == 3
    plan.longest = 100                                                              # what a plan that has only ever seen the sparse scene holds
    assert torch.equal(be.render_views(*sparse, res, res, c2w, k), ref_sparse)      # the LDS sort alone
    img = be.render_views(*dense, res, res, c2w, k)                                 # lists of ~9,000 entries against LDS for 2,048: the chunked path
    assert torch.equal(img, ref_dense)
    be.check_async()
    assert plan.form == 2 and plan.longest > 4096                                   # that call's statistics arrived: dense again
    assert torch.equal(be.render_views(*dense, res, res, c2w, k), ref_dense)
    assert plan.calls["healed"] == 0


def test_long_lists_through_the_lds_sort_alone_keep_lists_and_gradients():
    """The chunked path of the per-tile LDS sort (lists longer than the launch's LDS) leaves the reference's lists: image, point
    list and every gradient equal the synchronous call's."""
    res = 64
    dense, c2w, k = _scene(res, "init", 4)
    w = torch.randn(1, 2, 3, res, res, generator=torch.Generator().manual_seed(1)) / (3 * res * res)
    out = []
    for forced in (False, True):
        be = RasterBackend(lib=emu_lib())
        leaves = [x.clone().requires_grad_(True) for x in dense]
        if forced:
            render_views_autograd(be, *[x.detach() for x in dense], res, res, c2w, k)        # learns the capacity
            plan = be.plan_for(dense[0].shape[1], res, res, 2, 2, CPU)
            plan.form, plan.longest = 3, 64                                          # a stale plan: short lists, the LDS sort alone
            plan.note = lambda *a, **kw: None                                        # ... that stays stale for this call
        img = render_views_autograd(be, *leaves, res, res, c2w, k)
        (img * w).sum().backward()
        out.append([img.detach()] + [x.grad.clone() for x in leaves])
    for a, b in zip(*out):
        assert torch.equal(a, b)


def test_a_scene_that_outgrows_its_buffer_is_rendered_again_not_nan(monkeypatch):
    """A plan whose buffer is below the worst case of its shape verifies its calls: the call that does not fit is repeated with a
    buffer sized for it before it returns -- no NaN, no exception, the synchronous call's bits (what the reference guarantees by
    blocking on num_rendered in every forward, rasterizer_impl.cu:281-284)."""
    res = 64
    be = RasterBackend(lib=emu_lib())
    sparse, c2w, k = _scene(res, "small", 2)
    dense, _, _ = _scene(res, "init", 2)
    monkeypatch.setattr(_AsyncPlan, "MARGIN", 1.0)
    monkeypatch.setattr(_AsyncPlan, "BUDGET_BYTES", 0)                              # no worst-case floor: the history alone sizes the buffer
    be.render_views(*sparse, res, res, c2w, k)
    plan = be.plan_for(sparse[0].shape[1], res, res, 2, 2, CPU)
    cap = plan.capacity_for(True)
    assert plan.at_risk(cap)
    ref = RasterBackend(lib=emu_lib()).render_views(*dense, res, res, c2w, k)
    img = be.render_views(*dense, res, res, c2w, k)                  # > 3 x the instances of anything the plan has seen
    assert torch.equal(img, ref)
    assert plan.calls["healed"] == 1 and plan.capacity > cap and plan.seen_max > 3 * cap
    assert torch.equal(be.render_views(*dense, res, res, c2w, k), ref) and plan.calls["healed"] == 1
    be.check_async()


def test_the_training_render_heals_too(monkeypatch):
    """The autograd entry never takes the worst-case floor (its state feeds a backward whose scratch is sized per slot): its plans
    are verified, and a render that outgrew the buffer hands the backward the state of the repeated call."""
    res = 64
    sparse, c2w, k = _scene(res, "small", 5)
    dense, _, _ = _scene(res, "init", 5)
    monkeypatch.setattr(_AsyncPlan, "MARGIN", 1.0)
    w = torch.randn(1, 2, 3, res, res, generator=torch.Generator().manual_seed(2)) / (3 * res * res)
    out = []
    for warm in (False, True):
        be = RasterBackend(lib=emu_lib())
        if warm:
            render_views_autograd(be, *sparse, res, res, c2w, k)         # the plan knows the sparse scene only
        leaves = [x.clone().requires_grad_(True) for x in dense]
        img = render_views_autograd(be, *leaves, res, res, c2w, k)
        assert torch.isfinite(img).all()
        (img * w).sum().backward()
        out.append([img.detach()] + [x.grad.clone() for x in leaves])
        plan = be.plan_for(dense[0].shape[1], res, res, 2, 2, CPU)
        assert plan.calls["healed"] == (1 if warm else 0)
    for a, b in zip(*out):
        assert torch.equal(a, b)


def test_binning_form_helper_matches_the_kernels_choice():
    lib = emu_lib()
    P, W, H, V = 16386, 64, 64, 2
    T = 16
    tenth = -(-T * P * V // 10)
    assert lib.dgs_raster_binning_form(0, tenth, 100, P, W, H, V) == 2                 # a tenth of all pairs: scan
    assert lib.dgs_raster_binning_form(0, tenth - 1, 100, P, W, H, V) == 3             # below: lists, sorted in LDS ...
    assert lib.dgs_raster_binning_form(0, 1000, 16385, P, W, H, V) == 1                # ... unless one does not fit
    assert lib.dgs_raster_binning_form(3, 1000, 16385, P, W, H, V) == 1
    assert lib.dgs_raster_binning_form(1, 10 ** 9, 5, P, W, H, V) == 1
    assert lib.dgs_raster_binning_form(0, -1, 5, P, W, H, V) < 0
