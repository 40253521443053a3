"""RCCL on the one GPU the builder's box has: a process group of ONE rank with backend "nccl" (= RCCL on ROCm), created exactly
as bench.py / dgs_amd.parallel create it for N > 1 (`device_id`, 127.0.0.1 rendezvous, HSA_ENABLE_IPC_MODE_LEGACY=0), and every
collective of the N > 1 path actually issued (`force_collectives`): the async all-reduce of each gradient bucket launched from
the ctypes callback INSIDE dgs_dit_backward, on RCCL's stream, behind the compute stream's work so far; `finish()` waiting on the
works before the optimizer; the bf16 exchange's cast / sum / copy-back; bench.py's barrier and max-over-ranks reduction.

Stream ordering is checked EXACTLY and inside one run: RCCL short-cuts an in-place all-reduce over one rank (no data moves), so
beside it every bucket is also all-gathered (one rank: a copy performed on RCCL's stream, at the bucket's launch) into a probe
buffer; after the step the probe must equal the flat gradient buffer bit for bit -- a bucket whose collective ran before the
kernels that fill it had finished, or that was written again afterwards, differs.  Against the same step without collectives the
gradients are compared at the step's own run-to-run spread (two plain runs: the rasterizer backward sums a Gaussian's tiles with
fp32 atomics; only the DiT backward is bit-reproducible).  What a single GPU cannot show is xGMI traffic and scaling: no such
number exists.
"""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp



def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(port, out, backend="nccl"):
    import sys
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (here, os.path.join(here, "..", "open-diffusiongs_amd"), os.path.join(here, "..")):
        sys.path.insert(0, os.path.abspath(p))
    import numpy as np
    import torch.distributed as dist
    from dgs_amd import cameras, denoiser as dn, synth
    from dgs_amd.parallel import init_distributed
    from dgs_amd.train import DataParallelTrainer
    gpu = backend == "nccl"
    if gpu:
        dev, lib = torch.device("cuda", 0), None
        torch.cuda.set_device(0)
        res, V, RV, B = 64, 4, 2, 2
        cfg = dict(width=1024, in_channels=9, patch_size=8, num_layers=4)
        bucket_bytes = 16 << 20
    else:                                            # the same worker on gloo + the emulated kernels: checks this test's own logic on CPU
        from emu_util import emu_lib
        dev, lib = torch.device("cpu"), emu_lib()
        res, V, RV, B = 16, 2, 2, 1
        cfg = dict(width=256, in_channels=9, patch_size=8, num_layers=2)
        bucket_bytes = 1 << 20
    rank, world, _ = init_distributed(device=dev if gpu else None, backend=backend, force=True)
    assert (rank, world) == (0, 1) and dist.get_backend() == backend
    batch, t = synth.make_batch(B, res, V=V, device=dev, seed=5, with_t=True)
    rc2w = torch.tensor(np.stack([cameras.ring_cameras(RV, phase_deg=5.0 + 7 * b) for b in range(B)])).to(dev)
    rk = torch.tensor(cameras.default_fxfycxcy(res)).expand(B, RV, 4).contiguous().to(dev)
    target = torch.rand(B, RV, 3, res, res, generator=torch.Generator().manual_seed(2)).to(dev)
    results = {}
    for name, kw in (("none", dict()), ("none2", dict()), ("fp32", dict(force_collectives=True)), ("bf16", dict(force_collectives=True, compress="bf16"))):
        m = dn.DGSDenoiser(cfg, device=dev, lib=lib)
        m.reset_parameters(seed=4)
        m = m.to(dev)
        m.train()
        opt = torch.optim.SGD(m.parameters(), lr=0.0)
        # small buckets: several collectives leave DURING the backward (one per finished block group)
        with DataParallelTrainer(m, opt, bucket_bytes=bucket_bytes, **kw) as tr:
            assert tr.reducer.active == (not name.startswith("none"))
            if name == "fp32":
                tr.reducer.probe = torch.full_like(tr.fg.flat, float("nan"))
            for _ in range(2):                                   # two steps: the reducer resets, works are not leaked
                loss = tr.step(batch, t, target, rc2w, rk)
            if gpu:
                torch.cuda.synchronize()
            results[name] = (float(loss), tr.fg.flat.clone(), list(tr.reducer.launch_log), len(tr.reducer.bounds))
            if name == "fp32":
                probe_equal = bool(torch.equal(tr.reducer.probe, tr.fg.flat))
                probe_bad = int((tr.reducer.probe != tr.fg.flat).sum())
    # bench.py's timing plumbing on RCCL
    dist.barrier()
    tt = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if gpu:
        torch.cuda.synchronize()
    none, none2, fp32, bf16 = results["none"], results["none2"], results["fp32"], results["bf16"]
    rel = lambda x, y: float((x - y).norm() / y.norm())
    early = [b for b, tag in fp32[2] if isinstance(tag, int)]
    out.put(dict(
        loss=(none[0], fp32[0], bf16[0]),
        probe_equal=probe_equal, probe_bad=probe_bad,
        run_to_run_rel=rel(none2[1], none[1]), fp32_rel=rel(fp32[1], none[1]), bf16_rel=rel(bf16[1], none[1]),
        bf16_is_bf16=bool(torch.equal(bf16[1], bf16[1].to(torch.bfloat16).float())),
        finite=bool(torch.isfinite(fp32[1]).all()) and float(none[1].abs().max()) > 0.0,
        buckets=fp32[3], launched_during_backward=len(early), log=[(b, str(tag)) for b, tag in fp32[2]],
        max_reduce=float(tt.item())))
    dist.destroy_process_group()


def _run(backend):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q, backend))
    p.start()
    import queue, time
    r, t0 = None, time.time()
    while r is None and time.time() - t0 < 420:            # a crashed worker must not hold the (GPU) suite for the whole timeout
        try:
            r = q.get(timeout=2)
        except queue.Empty:
            assert p.is_alive() or not q.empty(), f"worker died with exit code {p.exitcode}"
    assert r is not None, "worker timed out"
    p.join(120)
    if p.is_alive():
        p.kill()
    assert p.exitcode == 0
    print(r)
    assert r["finite"]
    assert r["loss"][0] == r["loss"][1] == r["loss"][2]                 # the forward does not depend on the exchange
    assert r["probe_equal"], f"{r['probe_bad']} elements of a bucket were not final when its collective ran on RCCL's stream"
    assert r["fp32_rel"] <= 4 * r["run_to_run_rel"] + 1e-7, r            # a sum over one rank: the step's own run-to-run spread
    assert r["bf16_is_bf16"] and r["bf16_rel"] <= 4e-3 + 4 * r["run_to_run_rel"], r   # ... plus one bf16 rounding
    assert r["buckets"] >= 3 and r["launched_during_backward"] >= r["buckets"] - 1, r["log"]
    assert r["max_reduce"] == 1.25


@pytest.mark.gpu
def test_rccl_world_of_one_training_step_and_timing_reduction():
    _run("nccl")


def test_forced_collectives_world_of_one_on_gloo():
    """The same worker on gloo and the emulated kernels (CPU suite): the forced-collective path and this test's own assertions."""
    _run("gloo")
