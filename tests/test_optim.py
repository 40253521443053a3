"""The fused AdamW step + weight refresh (include/dgs_optim.h, dgs_amd/optim.py) against torch.optim.AdamW and the engine's own
torch-copy refresh: CPU-emulated kernel here, the same comparison on MI355X under `-m gpu`."""
import pytest
import torch

from dgs_amd import denoiser as dn
from dgs_amd.optim import FusedAdamW

CFG = dict(width=256, in_channels=9, patch_size=8, num_layers=2)


def _case(dev, lib, cfg, steps=3):
    m = dn.DGSDenoiser(cfg, device=dev, lib=lib)
    m.reset_parameters(seed=2)
    m = m.to(dev)
    eng = m.engine()
    eng._train_state()                                   # the transposed weight copies of the training path exist
    names = [n for n, _ in m.named_parameters()]
    dst = eng.weight_destinations()
    assert set(names) <= set(dst), sorted(set(names) - set(dst))       # every parameter has an engine copy
    ref = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    topt = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05, foreach=False, fused=False)
    fopt = FusedAdamW(m, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    g = torch.Generator().manual_seed(0)
    # the reference's scheduler (configs/diffusionGS_rel.yaml:64-68) drives both: FusedAdamW is a torch.optim.Optimizer
    scheds = [torch.optim.lr_scheduler.CosineAnnealingLR(o, T_max=4, eta_min=1e-4) for o in (topt, fopt)]
    for step in range(steps):
        for p, r in zip(m.parameters(), ref):
            gr = (torch.randn(p.shape, generator=g) * (0.1 + step)).to(dev)
            if p.grad is None:
                p.grad = gr.clone()
            else:
                p.grad.copy_(gr)                         # same gradient tensors every step, like the trainer's flat-buffer views
            r.grad = gr.clone()
        topt.step()
        fopt.step()
        for sc in scheds:
            sc.step()
        assert abs(topt.param_groups[0]["lr"] - fopt.param_groups[0]["lr"]) < 1e-12 and fopt.param_groups[0]["lr"] < 3e-3
    sd = fopt.state_dict()                               # round trip of the checkpoint form: torch's layout (state / param_groups)
    tsd = topt.state_dict()
    assert set(sd) == set(tsd) == {"state", "param_groups"} and sd["param_groups"][0]["betas"] == (0.9, 0.99)
    assert sd["param_groups"][0]["params"] == tsd["param_groups"][0]["params"] and set(sd["state"]) == set(tsd["state"])
    assert set(sd["state"][0]) >= {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == steps
    assert sd["state"][3]["exp_avg"].shape == tsd["state"][3]["exp_avg"].shape
    before = (fopt.exp_avg.clone(), fopt.exp_avg_sq.clone())
    fopt.exp_avg.zero_(); fopt.exp_avg_sq.zero_(); fopt.step_count = 0
    fopt.load_state_dict(sd)
    assert fopt.step_count == steps and torch.equal(fopt.exp_avg, before[0]) and torch.equal(fopt.exp_avg_sq, before[1])
    return m, eng, names, ref, topt, fopt


def _check(m, eng, names, ref, topt, fopt):
    worst = 0.0
    for (n, p), r in zip(m.named_parameters(), ref):
        err = float((p.detach() - r.detach()).abs().max() / (r.detach().abs().max() + 1e-12))
        worst = max(worst, err)
        assert err < 2e-6, (n, err)
        st = topt.state[r]
        off, numel = fopt._offsets[names.index(n)], p.numel()
        for mine, theirs in ((fopt.exp_avg, st["exp_avg"]), (fopt.exp_avg_sq, st["exp_avg_sq"])):      # fp32 rounding of another operation order
            assert float((mine[off:off + numel].view(p.shape) - theirs).abs().max()) <= 2e-6 * float(theirs.abs().max()), n
    # the engine's copies are the NEW values: bf16 round-to-nearest-even / fp32 copies, transposed copies of the bf16 ones --
    # exactly what the torch-copy refresh produces from the same parameters
    for n, p in m.named_parameters():
        copy, copy_t = eng.weight_destinations()[n]
        assert torch.equal(copy.reshape(p.shape), p.detach().to(copy.dtype)), n
        if copy_t is not None:
            assert torch.equal(copy_t, copy.t()), n
    assert any(t is not None for _, t in eng.weight_destinations().values())
    # version counters did not move: DGSDenoiser.engine() must not copy everything again
    assert m.engine() is eng and m._engine_version == tuple(p._version for p in m.parameters())
    return worst


def test_fused_adamw_matches_torch_and_refreshes_the_engine_on_the_emulator():
    from emu_util import emu_lib
    _check(*_case(torch.device("cpu"), emu_lib(), CFG))


def test_plan_rejects_bad_tables():
    import ctypes
    from dgs_amd import _native
    from emu_util import emu_lib
    lib = emu_lib()
    x = torch.zeros(64 * 96)
    e = _native.DgsAdamWTensor()
    e.p = e.g = e.m = e.v = x.data_ptr()
    e.rows, e.cols = 64, 96
    e.copy_t = x.data_ptr()                              # a transposed copy needs both sides to be multiples of 64
    assert lib.dgs_adamw_plan((_native.DgsAdamWTensor * 1)(e), 1) < 0
    e.copy_t = None
    assert lib.dgs_adamw_plan((_native.DgsAdamWTensor * 1)(e), 1) == 2          # flat: ceil(6144 / 4096) tiles
    assert x.data_ptr() % 16 == 0
    for field in ("p", "g", "m", "v"):                    # a view at an odd element offset: no 16-byte alignment for the float4 accesses
        setattr(e, field, x.data_ptr() + 4)
        assert lib.dgs_adamw_plan((_native.DgsAdamWTensor * 1)(e), 1) < 0, field
        setattr(e, field, x.data_ptr())
    e.copy, e.copy_kind = x.data_ptr() + 4, _native.OPTIM_COPY_BF16              # bf16 copy: 8-byte stores
    assert lib.dgs_adamw_plan((_native.DgsAdamWTensor * 1)(e), 1) < 0
    e.copy = x.data_ptr() + 8
    assert lib.dgs_adamw_plan((_native.DgsAdamWTensor * 1)(e), 1) == 2
    assert lib.dgs_adamw_step(None, None) != 0


def test_step_takes_a_closure_like_torch_optimizers():
    """Lightning's loop calls optimizer.step(closure=...): the closure runs first (under grad mode), its loss is returned."""
    from emu_util import emu_lib
    dev = torch.device("cpu")
    m = dn.DGSDenoiser(CFG, device=dev, lib=emu_lib())
    m.reset_parameters(seed=2)
    opt = FusedAdamW(m, lr=1e-3)
    calls = []

    def closure():
        assert torch.is_grad_enabled()
        for p in m.parameters():
            p.grad = torch.ones_like(p)
        calls.append(1)
        return torch.tensor(3.5)

    w0 = m.transformer[0].attn.qkv.weight.detach().clone()
    assert float(opt.step(closure=closure)) == 3.5 and calls == [1]
    assert not torch.equal(w0, m.transformer[0].attn.qkv.weight.detach())


def test_sumsq_and_clip_match_clip_grad_norm_on_the_emulator():
    """dgs_sumsq_partials / dgs_sumsq_finish + the clip folded into dgs_adamw_step == torch.nn.utils.clip_grad_norm_(params, 0.5) followed by
    torch.optim.AdamW (Lightning's `gradient_clip_val: 0.5`, configs/diffusionGS_rel.yaml:76-77), for a norm above AND below the bound."""
    from emu_util import emu_lib
    from dgs_amd.parallel import GradNorm
    lib = emu_lib()
    dev = torch.device("cpu")
    for scale, clipped in ((1.0, True), (1e-6, False)):
        m = dn.DGSDenoiser(CFG, device=dev, lib=lib)
        m.reset_parameters(seed=4)
        eng = m.engine()
        fg = eng._train_state()["fg"]
        ref = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
        g = torch.Generator().manual_seed(1)
        fg.flat.copy_(torch.randn(fg.flat.shape, generator=g) * scale)
        views = eng.grad_views()
        for (n, p), r in zip(m.named_parameters(), ref):
            p.grad = views[n].reshape(p.shape)
            r.grad = p.grad.clone()
        # padding between the slices of the flat buffer is not a gradient: it stays zero in the trainer (never written)
        mask = torch.zeros_like(fg.flat, dtype=torch.bool)
        for n in fg.names:
            off, numel, _ = fg.offsets[n]
            mask[off:off + numel] = True
        fg.flat.mul_(mask)
        norm = GradNorm(fg.flat, lib)
        half = (fg.flat.numel() // 2) // GradNorm.CHUNK * GradNorm.CHUNK
        norm.add(half, fg.flat.numel())                   # buckets in any order
        norm.add(0, half)
        total = norm.total()
        want = torch.nn.utils.clip_grad_norm_(ref, 0.5)
        assert abs(float(total.sqrt()) - float(want)) <= 1e-5 * float(want)
        assert (float(want) > 0.5) == clipped
        topt = torch.optim.AdamW(ref, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05, foreach=False, fused=False)
        fopt = FusedAdamW(m, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
        topt.step()
        fopt.step(grad_sumsq=total, max_grad_norm=0.5)
        for (n, p), r in zip(m.named_parameters(), ref):
            err = float((p.detach() - r.detach()).abs().max() / (r.detach().abs().max() + 1e-12))
            assert err < 2e-6, (n, err, clipped)
    with pytest.raises(ValueError):
        fopt.step(max_grad_norm=0.5)


def test_a_non_finite_gradient_norm_skips_the_whole_update():
    """What GradScaler.step does for the reference's 16-mixed training when a step produced inf / NaN: no parameter, no moment, no
    engine copy changes (the norm is the all-reduced one, so every rank skips alike); the next finite step updates as usual."""
    from emu_util import emu_lib
    lib = emu_lib()
    dev = torch.device("cpu")
    m = dn.DGSDenoiser(CFG, device=dev, lib=lib)
    m.reset_parameters(seed=5)
    eng = m.engine()
    fg = eng._train_state()["fg"]
    views = eng.grad_views()
    for n, p in m.named_parameters():
        p.grad = views[n].reshape(p.shape)
    fopt = FusedAdamW(m, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    fg.flat.copy_(torch.randn(fg.flat.shape, generator=torch.Generator().manual_seed(2)))
    fopt.step(grad_sumsq=torch.tensor([4.0]), max_grad_norm=0.5)            # one finite step: the moments exist
    before = [p.detach().clone() for p in m.parameters()]
    moments = (fopt.exp_avg.clone(), fopt.exp_avg_sq.clone())
    dst = eng.weight_destinations()
    copies = {n: tuple(None if c is None else c.clone() for c in pair) for n, pair in dst.items()}
    fg.flat.fill_(float("nan"))
    assert fopt.step_count == 1
    for bad in (float("nan"), float("inf")):
        fopt.step(grad_sumsq=torch.tensor([bad]), max_grad_norm=0.5)
        # ... and the step count (hence the bias corrections of the next real update) does not move: GradScaler never calls step()
        assert fopt.step_count == 1 and fopt.skipped_steps >= 1
        for p, b in zip(m.parameters(), before):
            assert torch.equal(p.detach(), b)
        assert torch.equal(fopt.exp_avg, moments[0]) and torch.equal(fopt.exp_avg_sq, moments[1])
    for n, pair in eng.weight_destinations().items():
        for now, was in zip(pair, copies[n]):
            assert (now is None and was is None) or torch.equal(now, was), n
    fg.flat.copy_(torch.randn(fg.flat.shape, generator=torch.Generator().manual_seed(3)))
    fopt.step(grad_sumsq=torch.tensor([4.0]), max_grad_norm=0.5)
    assert fopt.step_count == 2
    assert all(torch.isfinite(p).all() for p in m.parameters()) and not torch.equal(next(iter(m.parameters())).detach(), before[0])


@pytest.mark.gpu
def test_fused_adamw_matches_torch_on_gpu_at_the_shipped_width():
    worst = _check(*_case(torch.device("cuda:0"), None, dict(width=1024, in_channels=9, patch_size=8, num_layers=2)))
    print("worst relative parameter error vs torch.optim.AdamW:", worst)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["torch_fused", "torch_foreach", "dgs_fused"])
def test_engine_copies_follow_the_optimizer_in_a_trainer_step(kind):
    """After DataParallelTrainer.step the engine's bf16 / transposed copies hold the NEW parameter values, whichever optimizer made
    the update (torch's fused / foreach AdamW are free not to move the version counters DGSDenoiser.engine() watches)."""
    import numpy as np
    from dgs_amd import cameras, synth
    from dgs_amd.train import DataParallelTrainer
    dev = torch.device("cuda:0")
    m = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=2), device=dev)
    m.reset_parameters(seed=2)
    m = m.to(dev)
    m.train()
    if kind == "dgs_fused":
        opt = FusedAdamW(m, lr=1e-3)
    else:
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3, fused=kind == "torch_fused", foreach=kind == "torch_foreach")
    batch, t = synth.make_batch(1, 64, V=4, device=dev, seed=5, with_t=True)
    rc2w = torch.tensor(np.stack([cameras.ring_cameras(2, phase_deg=5.0)])).to(dev)
    rk = torch.tensor(cameras.default_fxfycxcy(64)).expand(1, 2, 4).contiguous().to(dev)
    target = torch.rand(1, 2, 3, 64, 64, device=dev)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    with DataParallelTrainer(m, opt) as tr:
        for _ in range(2):
            tr.step(batch, t, target, rc2w, rk)
        eng = m.engine()
        dst = eng.weight_destinations()
        moved = 0
        for n, p in m.named_parameters():
            copy, copy_t = dst[n]
            assert torch.equal(copy.reshape(p.shape), p.detach().to(copy.dtype)), (kind, n)
            if copy_t is not None:
                assert torch.equal(copy_t, copy.t()), (kind, n)
            moved += int(not torch.equal(p.detach(), before[n]))
        assert moved > 0.9 * len(before), (kind, moved)


@pytest.mark.gpu
def test_two_trainer_runs_from_the_same_state_are_bit_identical():
    """The training step is bit-reproducible by default: DataParallelTrainer selects the rasterizer backward without floating-point
    atomics (dgs_raster.h `scratch`), the DiT backward has none, the norm / clip / AdamW launches sum in a fixed order.  Three steps at
    256^2 (L = 4098, two samples, 4 rendered views each, clip + FusedAdamW), twice from the same seed: identical losses, identical
    gradient norms, every parameter identical bit for bit.  With deterministic=False the atomic form runs and says so."""
    import numpy as np
    from dgs_amd import cameras, synth
    from dgs_amd.train import DataParallelTrainer
    dev = torch.device("cuda:0")
    res, B, RV = 256, 2, 4
    batch, t = synth.make_batch(B, res, V=4, device=dev, seed=5, with_t=True)
    rc2w = torch.tensor(np.stack([cameras.ring_cameras(RV, phase_deg=5.0 + 7 * b) for b in range(B)])).to(dev)
    rk = torch.tensor(cameras.default_fxfycxcy(res)).expand(B, RV, 4).contiguous().to(dev)
    target = torch.rand(B, RV, 3, res, res, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    runs = []
    for _ in range(2):
        m = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=4), device=dev)
        m.reset_parameters(seed=2)
        m = m.to(dev)
        m.train()
        with DataParallelTrainer(m, FusedAdamW(m, lr=1e-4, betas=(0.9, 0.99), weight_decay=0.05), max_grad_norm=0.5) as tr:
            assert tr.deterministic and m.gs_renderer.backend().deterministic
            log = []
            for _ in range(3):
                loss = tr.step(batch, t, target, rc2w, rk)
                log.append((float(loss), float(tr.last_grad_sumsq)))
            assert m.gs_renderer.backend().last_backward_deterministic
        runs.append((log, {n: p.detach().clone() for n, p in m.named_parameters()}))
        # the trainer's choice lived on a backend of the model's own: the process-wide default backend was never switched, and the model
        # renders through it again now
        from dgs_amd.raster import default_backend
        assert m.gs_renderer.backend() is default_backend() and not default_backend().deterministic
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    bad = [n for n in runs[0][1] if not torch.equal(runs[0][1][n], runs[1][1][n])]
    assert not bad, (len(bad), bad[:6])
    m = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=4), device=dev)
    m.reset_parameters(seed=2)
    m = m.to(dev)
    m.train()
    with DataParallelTrainer(m, FusedAdamW(m, lr=1e-4), deterministic=False) as tr:
        tr.step(batch, t, target, rc2w, rk)
        assert m.gs_renderer.backend().last_backward_deterministic is False
