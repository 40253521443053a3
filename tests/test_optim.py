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
    sd = fopt.state_dict()                               # round trip of the checkpoint form
    fopt.load_state_dict(sd)
    assert sd["step"] == steps and sd["param_groups"][0]["betas"] == (0.9, 0.99)
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
    assert lib.dgs_adamw_step(None, None) != 0


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
