"""`DGSDenoiser.forward` without a host synchronisation, and as one captured hipGraph (dgs_amd/graph.py), on MI355X: the planned
rasterizer calls and the graph replays give the bits of the eager step; new inputs go through the graph's static tensors; the
30-step sampling loop on replays equals the eager loop; a scene that outgrows what the plan / the captured graph provided is
rendered again with buffers that fit -- eager and graph -- with the bits of the synchronous drop-in: no NaN, no exception (the
reference cannot fail there: rasterizer_impl.cu:281-284 reads num_rendered back in every forward and resizes, which is also why it
cannot be captured at all)."""
import pytest
import torch

from dgs_amd import denoiser as dn, sampler as sm, synth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
CFG = dict(width=1024, in_channels=9, patch_size=8, num_layers=4)


def _model(seed=0):
    m = dn.DGSDenoiser(CFG, device=DEV)
    m.reset_parameters(seed=seed)
    return m.to(DEV).eval()


def test_planned_renders_equal_the_synchronous_render_256():
    """Renderer.forward at 256^2, 4 views: call 1 runs the synchronous form (learns N), calls 2.. run from the plan -- no readback,
    one ordering form launched -- and give the same image bit for bit; so does the differentiable path, image and gradients."""
    from dgs_amd.raster import RasterBackend, render_views_autograd
    m = _model()
    batch, t = synth.make_batch(1, 256, V=4, device=DEV, seed=3, with_t=True)
    with torch.no_grad():
        params, _ = m.image_to_gaussians(batch["image"], batch["ray_o"], batch["ray_d"], t)
    be = RasterBackend()
    args = (params.xyz, params.features, params.scaling, params.rotation, params.opacity, 256, 256, batch["c2w"], batch["fxfycxcy"])
    first = be.render_views(*args)
    plan = be.plan_for(params.xyz.shape[1], 256, 256, 4, 4, DEV)
    assert plan.calls == {"sync": 1, "async": 0, "healed": 0} and plan.form in (1, 2, 3)
    for _ in range(3):
        assert torch.equal(be.render_views(*args), first)
    assert plan.calls == {"sync": 1, "async": 3, "healed": 0}
    # 256^2 x 4 views x P: the worst case (every Gaussian in every tile) fits the forward-only budget, so this plan cannot overflow
    assert not plan.at_risk(plan.capacity_for(True)) and plan.capacity_for(True) == 4 * 256 * params.xyz.shape[1]
    be.check_async()
    w = torch.randn(1, 4, 3, 256, 256, device=DEV, generator=torch.Generator(device=DEV).manual_seed(0)) / (3 * 256 * 256)
    grads = []
    be2 = RasterBackend()
    for _ in range(2):                                   # synchronous form, then planned
        leaves = [x.detach().clone().requires_grad_(True) for x in (params.xyz, params.features, params.scaling, params.rotation, params.opacity)]
        img = render_views_autograd(be2, *leaves, 256, 256, batch["c2w"], batch["fxfycxcy"])
        assert torch.equal(img.detach(), first)
        (img * w).sum().backward()
        grads.append([x.grad.clone() for x in leaves])
    for a, b, name in zip(grads[0], grads[1], ("xyz", "features", "scaling", "rotation", "opacity")):
        # the backward sums a Gaussian's tiles with fp32 atomics: equal to summation order
        assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max()) + 1e-12, name


def test_graph_replays_equal_the_eager_step():
    m = _model(seed=1)
    batch, t = synth.make_batch(1, 256, V=4, device=DEV, seed=4, with_t=True)
    with torch.no_grad():
        ref_render, ref_g = m(batch, t)
        ref_render, ref_xyz, ref_op = ref_render.clone(), ref_g[0]._xyz.clone(), ref_g[0]._opacity.clone()
        g = m.graphed(batch, t)
        assert m.graphed(batch, t) is g                                 # one graph per shape
        for _ in range(3):
            render, gaussians = g(batch, t)
            assert torch.equal(render, ref_render) and torch.equal(gaussians[0]._xyz, ref_xyz) and torch.equal(gaussians[0]._opacity, ref_op)
        # other inputs of the same shapes: copied into the graph's tensors, same result as the eager call on them
        batch2, t2 = synth.make_batch(1, 256, V=4, device=DEV, seed=9, with_t=True)
        want, want_g = m(batch2, t2)
        want, want_xyz = want.clone(), want_g[0]._xyz.clone()
        got, got_g = g(batch2, t2)
        torch.cuda.synchronize()
        plan0, row0 = g._watch[0][0], g._watch[0][1]
        diag = dict(stats=row0.tolist(), capacity=plan0.capacity, seen_max=plan0.seen_max, form=plan0.form, verify=g._verify,
                    xyz_equal=bool(torch.equal(got_g[0]._xyz, want_xyz)), nan_frac=float(torch.isnan(got).float().mean()))
        assert torch.equal(got, want) and torch.equal(got_g[0]._xyz, want_xyz) and not torch.equal(got, ref_render), diag
        # weights change (load_state_dict / optimizer step): the engine's copies are refreshed in place, the SAME graph sees them
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        sd["transformer.1.mlp.fc2.weight"] = sd["transformer.1.mlp.fc2.weight"] * 1.5
        m.load_state_dict(sd)
        want3, _ = m(batch2, t2)
        want3 = want3.clone()
        got3, _ = m.graphed(batch2, t2)(batch2, t2)
        assert m.graphed(batch2, t2) is g and torch.equal(got3, want3) and not torch.equal(got3, want)
    assert g.replays >= 5


def test_sampling_loop_on_graph_replays_equals_the_eager_loop():
    m = _model(seed=2)
    res = 128
    batch, _ = synth.make_batch(1, res, V=4, device=DEV, seed=6, with_t=True)
    d = sm.create_diffusion("30", device=DEV)
    outs = []
    for use_graph in (False, True, True):
        b = dict(batch)
        b["image"] = batch["image"].clone()
        b["image_noisy"] = torch.randn(1, 3, 3, res, res, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        torch.manual_seed(5)
        with torch.no_grad():
            outs.append(d.p_sample_loop(m, b, use_graph=use_graph))
    for o in outs[1:]:
        assert torch.equal(o["sample"], outs[0]["sample"]) and torch.equal(o["pred_xstart"], outs[0]["pred_xstart"])
        assert torch.equal(o["denoiser_output_dict"]["render_images"], outs[0]["denoiser_output_dict"]["render_images"])
        assert torch.equal(o["denoiser_output_dict"]["pred_gaussians"][0]._xyz, outs[0]["denoiser_output_dict"]["pred_gaussians"][0]._xyz)
    # the last loop's outputs were cloned out of the graph's tensors: another replay does not change them
    keep = outs[2]["denoiser_output_dict"]["render_images"].clone()
    with torch.no_grad():
        m.graphed(batch, torch.zeros(1, dtype=torch.int64, device=DEV))(batch, torch.zeros(1, dtype=torch.int64, device=DEV))
    assert torch.equal(outs[2]["denoiser_output_dict"]["render_images"], keep)


def _far(batch, factor):
    """The same batch with the render cameras `factor` times as far from the origin: every Gaussian covers ~factor^2 fewer tiles."""
    far = dict(batch)
    far["c2w"] = batch["c2w"].clone()
    far["c2w"][..., :3, 3] *= factor
    return far


def _fresh_reference(m, batch, t):
    """What the synchronous drop-in renders: the first call of a shape on a backend that has seen nothing reads num_rendered back
    and sizes its buffer exactly (the reference's rasterizer_impl.cu:281-284)."""
    from dgs_amd.raster import RasterBackend
    keep = m.gs_renderer._backend
    m.gs_renderer._backend = RasterBackend()
    try:
        out, _ = m(batch, t)
        return out.clone()
    finally:
        m.gs_renderer._backend = keep


def test_eager_forward_recovers_from_an_instance_jump(monkeypatch):
    """DGSDenoiser.forward, eager: the plan has only seen the far cameras; the near ones need > 4 x the instances.  With the
    forward-only budget taken away the plan is at risk, verifies the call and renders it again: the synchronous call's bits."""
    from dgs_amd.raster import RasterBackend, _AsyncPlan
    monkeypatch.setattr(_AsyncPlan, "MARGIN", 1.0)
    monkeypatch.setattr(_AsyncPlan, "BUDGET_BYTES", 0)
    m = _model(seed=3)
    res = 128
    batch, t = synth.make_batch(1, res, V=4, device=DEV, seed=2, with_t=True)
    far = _far(batch, 5.0)
    with torch.no_grad():
        want_far, want_near = _fresh_reference(m, far, t), _fresh_reference(m, batch, t)
        m.gs_renderer._backend = be = RasterBackend()
        for _ in range(3):
            out, _ = m(far, t)
            assert torch.equal(out, want_far)
        plan = next(iter(be._plans.values()))
        seen_far = plan.seen_max
        assert plan.calls == {"sync": 1, "async": 2, "healed": 0} and plan.at_risk(plan.capacity_for(True))
        out, _ = m(batch, t)
        assert torch.equal(out, want_near) and torch.isfinite(out).all()
        assert plan.calls["healed"] == 1 and plan.seen_max > 4 * seen_far, (plan.seen_max, seen_far)
        out, _ = m(batch, t)
        assert torch.equal(out, want_near) and plan.calls["healed"] == 1
        out, _ = m(far, t)
        assert torch.equal(out, want_far)
        be.check_async(wait=True)


def test_graph_overflow_recovers(monkeypatch):
    """The same jump through ONE captured graph: the replay that outgrew the captured capacity is detected (a graph whose plan is at
    risk verifies every replay), the graph is captured again with buffers that fit and replayed -- the caller gets the synchronous
    call's bits from the same `graphed(...)` object, then keeps replaying the new capture."""
    from dgs_amd.raster import RasterBackend, _AsyncPlan
    monkeypatch.setattr(_AsyncPlan, "MARGIN", 1.0)
    monkeypatch.setattr(_AsyncPlan, "BUDGET_BYTES", 0)
    m = _model(seed=3)
    res = 128
    batch, t = synth.make_batch(1, res, V=4, device=DEV, seed=2, with_t=True)
    far = _far(batch, 5.0)
    with torch.no_grad():
        want_far, want_near = _fresh_reference(m, far, t), _fresh_reference(m, batch, t)
        m.gs_renderer._backend = RasterBackend()             # a backend of its own: fresh plans
        g = m.graphed(far, t)
        assert g._verify and g.healed == 0
        out, _ = g(far, t)
        assert torch.equal(out, want_far)
        seen_far = g._watch[0][0].seen_max
        out, _ = g(batch, t)
        assert torch.equal(out, want_near) and torch.isfinite(out).all()
        assert g.healed == 1 and g.recaptures == 1 and g._watch[0][0].seen_max > 4 * seen_far
        assert m.graphed(batch, t) is g
        for _ in range(2):
            out, _ = g(batch, t)
            assert torch.equal(out, want_near)
        out, _ = g(far, t)
        assert torch.equal(out, want_far) and g.healed == 1


def test_graph_with_the_worst_case_buffer_never_waits():
    """Default budget: at 128^2 (and at the object model's 256^2) the forward-only plan holds the worst case of its shape, the graph
    is not verified (no host wait per replay) and the jump is simply rendered."""
    from dgs_amd.raster import RasterBackend
    m = _model(seed=3)
    res = 128
    batch, t = synth.make_batch(1, res, V=4, device=DEV, seed=2, with_t=True)
    far = _far(batch, 5.0)
    with torch.no_grad():
        want_near = _fresh_reference(m, batch, t)
        m.gs_renderer._backend = RasterBackend()
        g = m.graphed(far, t)
        assert not g._verify
        g(far, t)
        out, _ = g(batch, t)
        g.check(wait=True)
        assert torch.equal(out, want_near) and g.healed == 0 and g.recaptures == 0
