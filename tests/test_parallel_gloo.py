"""N > 1 path on CPU: world_size-2 `gloo` processes exercise the flat-gradient bucketed all-reduce and the bench's
max-over-ranks timing reduction (the product uses the same code with backend "nccl" = RCCL)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dgs_amd.parallel import BucketedAllReduce, FlatGrads, init_distributed
    r, w, _ = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    shapes = [("head.w", (7, 5)), ("block1.qkv", (33, 8)), ("block0.qkv", (33, 8)), ("embed", (100,))]
    fg = FlatGrads(shapes, "cpu")
    red = BucketedAllReduce(fg.flat, bucket_bytes=1024)
    g = torch.Generator().manual_seed(100 + rank)
    for step in range(2):                       # two steps: the reducer must reset cleanly
        for name, shape in shapes:              # "backward" fills slices in layout order and releases prefixes
            fg.view(name).copy_(torch.randn(shape, generator=g))
            red.ready_up_to(fg.end_of(name))
        red.finish()
        gs = [torch.Generator().manual_seed(100 + k) for k in range(world)]
        for _ in range(step + 1):
            want = {name: sum(torch.randn(shape, generator=gk) for gk in gs) / world for name, shape in shapes}
        for name, _ in shapes:
            assert torch.allclose(fg.view(name), want[name], atol=1e-6), (rank, step, name)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)      # bench.py: elapsed = max over ranks
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out.put((rank, float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(2))
    assert got == [(0, 2.0), (1, 2.0)]


def _train_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "..", "open-diffusiongs_amd"), os.path.join(here, "..")):
        sys.path.insert(0, os.path.abspath(p))
    from dgs_amd import denoiser as dn
    from dgs_amd.parallel import init_distributed
    from dgs_amd.train import DataParallelTrainer
    from dit_util import synth_inputs
    from emu_util import emu_lib
    from oracle import dit_oracle as D
    init_distributed(backend="gloo")
    cfg = D.Cfg(width=256, num_layers=1)
    m = dn.DGSDenoiser(dict(width=256, in_channels=9, patch_size=8, num_layers=1), device="cpu", lib=emu_lib())
    m.reset_parameters(seed=1)                                   # identical replicas
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, 2, 2, 16, seed=9)      # global batch of 2; rank r takes sample r
    sl = slice(rank, rank + 1) if world > 1 else slice(0, 2)
    batch = dict(image=images[sl], ray_o=ray_o[sl], ray_d=ray_d[sl], c2w=c2w[sl], fxfycxcy=k[sl])
    target = torch.rand(2, 2, 3, 16, 16, generator=torch.Generator().manual_seed(3))[sl]
    tr = DataParallelTrainer(m, torch.optim.SGD(m.parameters(), lr=0.0), bucket_bytes=1 << 20)
    loss = tr.step(batch, t[sl], target)
    g = torch.cat([p.grad.reshape(-1) for p in m.parameters()])
    # the .grad tensors ARE the flat buffer (reduced in place, nothing copied out)
    flat = tr.fg.flat
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    assert all(lo <= p.grad.data_ptr() < hi for p in m.parameters())
    out.put((rank, float(loss), g.numpy(), list(tr.reducer.launch_log)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_data_parallel_step_equals_single_process_step():
    """2 ranks x 1 sample, gradients averaged by the bucketed all-reduce == 1 process x 2 samples (mean loss)."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(2)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_train_worker, args=(0, 1, _free_port(), q1))
    p1.start()
    single = q1.get(timeout=240)
    p1.join(60)
    res.sort(key=lambda r: r[0])
    np.testing.assert_allclose(res[0][2], res[1][2], rtol=0, atol=0)            # ranks hold identical averaged gradients
    # overlap: buckets were enqueued from inside dgs_dit_backward (tag = the block / head group that completed them), i.e.
    # BEFORE the backward call returned; at most the last partial bucket is left for finish()
    log = res[0][3]
    early = [b for b, tag in log if isinstance(tag, int)]
    assert len(early) >= 2 and len(early) >= len(log) - 1, log
    assert [b for b, _ in log] == list(range(len(log)))
    assert any(tag == 0 for _, tag in log) and any(tag == -1 for _, tag in log), log    # completed by block 0, then by the rest
    np.testing.assert_allclose(0.5 * (res[0][1] + res[1][1]), single[1], rtol=1e-5)
    denom = np.abs(single[2]).max()
    assert np.abs(res[0][2] - single[2]).max() <= 2e-3 * denom


def _accum_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "..", "open-diffusiongs_amd"), os.path.join(here, "..")):
        sys.path.insert(0, os.path.abspath(p))
    from dgs_amd import denoiser as dn
    from dgs_amd.train import DataParallelTrainer
    from dit_util import synth_inputs
    from emu_util import emu_lib
    from oracle import dit_oracle as D
    cfg = D.Cfg(width=256, num_layers=1)
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, 2, 2, 16, seed=9)
    batch = dict(image=images, ray_o=ray_o, ray_d=ray_d, c2w=c2w, fxfycxcy=k)
    target = torch.rand(2, 2, 3, 16, 16, generator=torch.Generator().manual_seed(3))
    res = []
    for accumulate in (1, 2):
        m = dn.DGSDenoiser(dict(width=256, in_channels=9, patch_size=8, num_layers=1), device="cpu", lib=emu_lib())
        m.reset_parameters(seed=1)
        tr = DataParallelTrainer(m, torch.optim.SGD(m.parameters(), lr=0.0), bucket_bytes=1 << 20, accumulate_grad_batches=accumulate)
        loss = tr.step(batch, t, target)
        res.append((float(loss), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy()))
    # a per-rank batch above the 4 samples one C call takes (the reference's scene configs: 12 / 24 per rank): the trainer splits it
    # into micro-batches inside ONE optimizer step; then a small batch again (the accumulator of the split step must not leak into it)
    big = synth_inputs(cfg, 6, 2, 16, seed=9)
    bb = dict(image=big[0], ray_o=big[1], ray_d=big[2], c2w=big[4], fxfycxcy=big[5])
    tb = torch.rand(6, 2, 3, 16, 16, generator=torch.Generator().manual_seed(3))
    m = dn.DGSDenoiser(dict(width=256, in_channels=9, patch_size=8, num_layers=1), device="cpu", lib=emu_lib())
    m.reset_parameters(seed=1)
    tr = DataParallelTrainer(m, torch.optim.SGD(m.parameters(), lr=0.0), bucket_bytes=1 << 20)
    loss6 = tr.step(bb, big[3], tb)
    g6 = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy().copy()
    loss2 = tr.step(batch, t, target)
    g2 = torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy().copy()
    tr.close()
    m.zero_grad()
    from dgs_amd import losses
    p6, _ = m.image_to_gaussians(bb["image"], bb["ray_o"], bb["ray_d"], big[3])       # autograd: chunks of 4 + 2, gradients summed by torch
    l6, _, _ = losses.mse_psnr(m.render_gaussians(p6, bb["c2w"], bb["fxfycxcy"], 16, 16), tb, lib=emu_lib())
    l6.backward()
    res.append((float(loss6), g6, float(l6), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).numpy(), float(loss2), g2))
    out.put(res)


def test_accumulate_grad_batches_equals_one_large_batch():
    """accumulate_grad_batches = 2 over micro-batches of 1 == one batch of 2 (the way the 512^2 configuration reaches the
    reference's per-rank batch of 12 with micro-batches of 4): gradients of the earlier micro-batches are folded into the
    flat buffer slice by slice inside the last backward, right before each bucket's all-reduce."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_accum_worker, args=(0, 1, _free_port(), q))
    p.start()
    (l1, g1), (l2, g2), (l6, g6, l6_ref, g6_ref, l2b, g2b) = q.get(timeout=600)
    p.join(60)
    assert p.exitcode == 0
    assert abs(l1 - l2) < 1e-6 * max(1.0, abs(l1))
    assert np.abs(g1 - g2).max() <= 2e-3 * np.abs(g1).max()
    # batch of 6 in one trainer step (split 2 x 3) == autograd over the same batch (chunks 4 + 2)
    assert abs(l6 - l6_ref) < 1e-5 * max(1.0, abs(l6_ref))
    assert np.abs(g6 - g6_ref).max() <= 2e-3 * np.abs(g6_ref).max()
    # and the next, small step is clean
    assert abs(l2b - l1) < 1e-6 * max(1.0, abs(l1)) and np.abs(g2b - g1).max() <= 1e-6 * np.abs(g1).max()


def _fused_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "..", "open-diffusiongs_amd"), os.path.join(here, "..")):
        sys.path.insert(0, os.path.abspath(p))
    from dgs_amd import denoiser as dn
    from dgs_amd.optim import FusedAdamW
    from dgs_amd.parallel import init_distributed
    from dgs_amd.train import DataParallelTrainer
    from dit_util import synth_inputs
    from emu_util import emu_lib
    from oracle import dit_oracle as D
    init_distributed(backend="gloo")
    cfg = D.Cfg(width=256, num_layers=1)
    m = dn.DGSDenoiser(dict(width=256, in_channels=9, patch_size=8, num_layers=1), device="cpu", lib=emu_lib())
    m.reset_parameters(seed=1)
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, 2, 2, 16, seed=9)
    sl = slice(rank, rank + 1) if world > 1 else slice(0, 2)
    batch = dict(image=images[sl], ray_o=ray_o[sl], ray_d=ray_d[sl], c2w=c2w[sl], fxfycxcy=k[sl])
    target = torch.rand(2, 2, 3, 16, 16, generator=torch.Generator().manual_seed(3))[sl]
    tr = DataParallelTrainer(m, FusedAdamW(m, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.05), bucket_bytes=1 << 20)
    for _ in range(2):
        tr.step(batch, t[sl], target)
    p = torch.cat([q.detach().reshape(-1) for q in m.parameters()])
    eng = m.engine()
    n = "transformer.0.mlp.fc1.weight"
    copy, copy_t = eng.weight_destinations()[n]
    ok = bool(torch.equal(copy, dict(m.named_parameters())[n].detach().to(torch.bfloat16)) and torch.equal(copy_t, copy.t()))
    out.put((rank, p.numpy(), ok))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_fused_adamw_keeps_the_replicas_identical():
    """Two ranks x one sample with the one-launch AdamW + weight refresh: after two steps both replicas hold the SAME parameters (the
    averaged gradients are identical bits on both ranks, the update is deterministic) and they match one process x two samples; the
    engine's bf16 / transposed copies follow on every rank."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fused_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    q1 = ctx.Queue()
    p1 = ctx.Process(target=_fused_worker, args=(0, 1, _free_port(), q1))
    p1.start()
    single = q1.get(timeout=300)
    p1.join(60)
    assert res[0][2] and res[1][2] and single[2]
    np.testing.assert_array_equal(res[0][1], res[1][1])
    # Adam normalises the step, so gradient noise of 2e-3 (bf16 operands, another summation order) can move a parameter by up to
    # ~lr where the gradient is tiny: compare against the size of the update, not of the parameter
    assert np.abs(res[0][1] - single[1]).max() <= 2.5e-2
    assert np.abs(res[0][1] - single[1]).mean() <= 2e-3


def _clip_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "..", "open-diffusiongs_amd"), os.path.join(here, "..")):
        sys.path.insert(0, os.path.abspath(p))
    from dgs_amd import denoiser as dn
    from dgs_amd.optim import FusedAdamW
    from dgs_amd.parallel import init_distributed
    from dgs_amd.train import DataParallelTrainer
    from dit_util import synth_inputs
    from emu_util import emu_lib
    from oracle import dit_oracle as D
    init_distributed(backend="gloo")
    cfg = D.Cfg(width=256, num_layers=1)
    m = dn.DGSDenoiser(dict(width=256, in_channels=9, patch_size=8, num_layers=1), device="cpu", lib=emu_lib())
    m.reset_parameters(seed=1 + 10 * rank)           # the ranks START from different weights: the trainer's broadcast makes them rank 0's
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, 2, 2, 16, seed=9)
    sl = slice(rank, rank + 1)
    batch = dict(image=images[sl], ray_o=ray_o[sl], ray_d=ray_d[sl], c2w=c2w[sl], fxfycxcy=k[sl])
    target = torch.rand(2, 2, 3, 16, 16, generator=torch.Generator().manual_seed(3))[sl]
    res = {}
    for kind in ("fused", "sgd"):
        m.reset_parameters(seed=1 + 10 * rank)
        m.refresh_engine_weights()
        opt = FusedAdamW(m, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.0) if kind == "fused" else torch.optim.SGD(m.parameters(), lr=1.0)
        with DataParallelTrainer(m, opt, bucket_bytes=1 << 20, max_grad_norm=1e-3, compress="bf16" if kind == "sgd" else None) as tr:
            start = torch.cat([q.detach().reshape(-1) for q in m.parameters()]).clone()
            assert tr.broadcast_bytes > 0
            tr.step(batch, t[sl], target)
            norm = float(tr.last_grad_sumsq.sqrt())
            grads = torch.cat([q.grad.reshape(-1) for q in m.parameters()]).clone()
        end = torch.cat([q.detach().reshape(-1) for q in m.parameters()])
        res[kind] = (start.numpy(), end.numpy(), norm, grads.numpy())
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_global_norm_clip_two_ranks():
    """DDP's init-time parameter broadcast (ranks constructed with different seeds leave the trainer's constructor with rank 0's values)
    and the global-norm clip (Lightning gradient_clip_val, configs/diffusionGS_rel.yaml:76-77): the norm is the norm of the AVERAGED
    gradient, identical on both ranks; with SGD(lr = 1) the step is exactly -clip_coef * gradient, so its length is max_grad_norm."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_clip_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=400) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    r0, r1 = res[0][1], res[1][1]
    for kind in ("fused", "sgd"):
        np.testing.assert_array_equal(r0[kind][0], r1[kind][0])          # same start: rank 1 took rank 0's parameters
        np.testing.assert_array_equal(r0[kind][1], r1[kind][1])          # same end: identical averaged gradients, identical norm, deterministic update
        assert r0[kind][2] == r1[kind][2] and r0[kind][2] > 1e-3          # the norm before clipping, above the bound
    g = r0["fused"][3].astype(np.float64)                                # FusedAdamW reads g * coef: the gradient tensors keep their values
    assert abs(np.sqrt((g * g).sum()) - r0["fused"][2]) <= 1e-5 * r0["fused"][2]
    start, end, norm, g = r0["sgd"]                                      # any other optimizer: the flat buffer (= the .grad views) is
    g = g.astype(np.float64)                                             # scaled in place, like torch.nn.utils.clip_grad_norm_
    assert abs(np.sqrt((g * g).sum()) - 1e-3 * norm / (norm + 1e-6)) <= 1e-8
    step = (end.astype(np.float64) - start)
    np.testing.assert_allclose(step, -g, atol=1.2e-7)                    # SGD(lr = 1): the step IS the clipped gradient (to the fp32 rounding of p - g, p <= 1)


def _overflow_worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "..", "open-diffusiongs_amd"), os.path.join(here, "..")):
        sys.path.insert(0, os.path.abspath(p))
    from dgs_amd import denoiser as dn
    from dgs_amd.optim import FusedAdamW
    from dgs_amd.parallel import init_distributed
    from dgs_amd.raster import RasterBackend, _AsyncPlan
    from dgs_amd.train import DataParallelTrainer
    from dit_util import synth_inputs
    from emu_util import emu_lib
    from oracle import dit_oracle as D
    init_distributed(backend="gloo")
    _AsyncPlan.MARGIN = 1.3                                     # the buffer holds little more than the plan has seen (an optimizer step moves the scene a bit)
    cfg = D.Cfg(width=256, num_layers=1)
    res = 64                                                    # 16 tiles per view: the near cameras give 2.2 x the far cameras' instances
    images, ray_o, ray_d, t, c2w, k = synth_inputs(cfg, 2, 2, res, seed=9)
    sl = slice(rank, rank + 1)
    batch = dict(image=images[sl], ray_o=ray_o[sl], ray_d=ray_d[sl], c2w=c2w[sl], fxfycxcy=k[sl])
    far = c2w[sl].clone()
    far[..., :3, 3] *= 3.0                                      # cameras three times as far: fewer tiles per Gaussian
    target = torch.rand(2, 2, 3, res, res, generator=torch.Generator().manual_seed(3))[sl]
    runs = {}
    for mode in ("jump", "reference"):
        m = dn.DGSDenoiser(dict(width=256, in_channels=9, patch_size=8, num_layers=1), device="cpu", lib=emu_lib())
        m.reset_parameters(seed=1)
        be = m.gs_renderer._backend = RasterBackend(lib=emu_lib())
        with DataParallelTrainer(m, FusedAdamW(m, lr=1e-2, betas=(0.9, 0.99), weight_decay=0.0), bucket_bytes=1 << 20, max_grad_norm=0.5) as tr:
            if mode == "jump":
                tr.step(batch, t[sl], target, render_c2w=far)        # both ranks: the plan learns the far cameras' scene
            else:                                                    # same first step, then a backend that has seen nothing: every
                tr.step(batch, t[sl], target, render_c2w=far)        # render of step 2 runs the synchronous form (exact buffers)
                be = m.gs_renderer._backend = RasterBackend(lib=emu_lib())
                be.deterministic = tr.deterministic              # the trainer selected the atomic-free backward on the backend it found
            # step 2: rank 0 renders from the near cameras (far more instances than its plan provides), rank 1 stays far
            loss = tr.step(batch, t[sl], target, render_c2w=None if rank == 0 else far)
            plan = next(iter(be._plans.values()))
            runs[mode] = (float(loss), float(tr.last_grad_sumsq), dict(plan.calls),
                          torch.cat([q.detach().reshape(-1) for q in m.parameters()]).numpy())
    out.put((rank, runs))
    dist.barrier()
    dist.destroy_process_group()


def test_one_rank_outgrows_its_raster_plan_and_the_ranks_stay_in_step():
    """Rank 0's render outgrows the binning buffer its plan provides while rank 1's does not: rank 0 repeats the render inside its
    forward (dgs_amd/raster.py `_AsyncPlan`: a plan at risk verifies its calls), nothing non-finite reaches the loss, the gradient
    all-reduce and the clip see the same collectives on both ranks, and both replicas end with the parameters of the run in which
    every render used the synchronous form."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_overflow_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    r0, r1 = res[0][1], res[1][1]
    assert r0["jump"][2]["healed"] == 1 and r1["jump"][2]["healed"] == 0, (r0["jump"][2], r1["jump"][2])
    assert r0["reference"][2]["healed"] == 0 and r0["reference"][2]["async"] == 0
    for r in (r0, r1):
        assert np.isfinite(r["jump"][0]) and np.isfinite(r["jump"][1]) and np.isfinite(r["jump"][3]).all()
        assert r["jump"][0] == r["reference"][0] and r["jump"][1] == r["reference"][1]       # same loss, same (all-reduced) norm
        np.testing.assert_array_equal(r["jump"][3], r["reference"][3])                      # same parameters
    np.testing.assert_array_equal(r0["jump"][3], r1["jump"][3])                             # replicas identical
    assert r0["jump"][1] == r1["jump"][1]
