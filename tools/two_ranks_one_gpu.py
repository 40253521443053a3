"""Two ranks of `DataParallelTrainer` on ONE MI355X: both processes on cuda:0, the real HIP kernels, `torch.distributed` on gloo (RCCL
refuses two ranks on one device).  Not RCCL and not a performance run -- gloo's collectives stage through the host and `wait()` blocks
it -- but the first time the two-rank step (different data per rank, bucketed all-reduce launched from inside the backward, global-norm
clip across ranks, fused AdamW, deterministic rasterizer backward) runs on the GPU kernels instead of the emulator:
  * after every step the two ranks hold the same parameters bit for bit and report the same gradient norm;
  * their step equals, to summation order, the single-process step on the combined batch of two.
    python tools/two_ranks_one_gpu.py [steps, default 3]
Development tool (profiles/r05_two_ranks_one_gpu.txt)."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import numpy as np
import torch
import torch.multiprocessing as mp

CFG = dict(width=1024, in_channels=9, patch_size=8, num_layers=4, ray_pe_type="relative_plk")
RES, RV = 256, 4


def data(dev):
    from dgs_amd import cameras, synth
    batch, t = synth.make_batch(2, RES, V=4, device=dev, seed=21, with_t=True)      # the global batch of two; rank r takes sample r
    rc2w = torch.tensor(np.stack([cameras.ring_cameras(RV, phase_deg=5.0 + 7 * b) for b in range(2)])).to(dev)
    rk = torch.tensor(cameras.default_fxfycxcy(RES)).expand(2, RV, 4).contiguous().to(dev)
    target = torch.rand(2, RV, 3, RES, RES, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    return batch, t, rc2w, rk, target


def worker(rank, world, port, steps, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dgs_amd import denoiser as dn
    from dgs_amd.optim import FusedAdamW
    from dgs_amd.parallel import init_distributed
    from dgs_amd.train import DataParallelTrainer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if world > 1:
        init_distributed(backend="gloo")
    batch, t, rc2w, rk, target = data(dev)
    sl = slice(rank, rank + 1) if world > 1 else slice(0, 2)
    m = dn.DGSDenoiser(CFG, device=dev)
    m.reset_parameters(seed=3 + (rank if world > 1 else 0))     # DIFFERENT replicas: the trainer's init-time broadcast has to make them rank 0's
    if world == 1:
        m.reset_parameters(seed=3)
    m = m.to(dev)
    m.train()
    log = []
    with DataParallelTrainer(m, FusedAdamW(m, lr=1e-4, betas=(0.9, 0.99), weight_decay=0.05), max_grad_norm=0.5, bucket_bytes=64 << 20) as tr:
        for _ in range(steps):
            loss = tr.step({k: v[sl] for k, v in batch.items()}, t[sl], target[sl], rc2w[sl], rk[sl])
            log.append((float(loss), float(tr.last_grad_sumsq) ** 0.5))
        buckets = len(tr.reducer.bounds)
        early = sum(1 for _b, tag in tr.reducer.launch_log if isinstance(tag, int))
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()]).cpu()
    out.put((rank, world, log, flat.numpy(), buckets, early))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, steps, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda r: r[0])
    for p in procs:
        p.join(120)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    p1 = ctx.Process(target=worker, args=(0, 1, free_port(), steps, q))
    p1.start()
    single = q.get(timeout=600)
    p1.join(120)
    same = bool(np.array_equal(res[0][3], res[1][3]))
    norms = [g for _l, g in res[0][2]] == [g for _l, g in res[1][2]]
    print(f"[two ranks, one GPU] gloo, world 2, both on cuda:0, {steps} steps, 4 blocks x width 1024 at {RES}^2, {RV} rendered views, "
          f"{res[0][4]} buckets per step of which {res[0][5]} launched from inside the backward")
    for r in res:
        print(f"    rank {r[0]}: (loss, global gradient norm) per step {[(round(l, 6), round(g, 6)) for l, g in r[2]]}")
    print(f"    single process, batch of 2: {[(round(l, 6), round(g, 6)) for l, g in single[2]]}")
    print(f"    parameters identical on both ranks, bit for bit: {same}; same global gradient norm on both ranks every step: {norms}")
    d = float(np.abs(res[0][3] - single[3]).max())
    moved = float(np.abs(single[3]).max())
    mean_loss = [0.5 * (a[0] + b[0]) for a, b in zip(res[0][2], res[1][2])]
    print(f"    against the single-process run on the combined batch: mean of the ranks' losses {[round(x, 6) for x in mean_loss]}, "
          f"largest parameter difference after {steps} steps {d:.3e} (largest parameter {moved:.3f})")
    # step 1 starts from identical parameters: the ranks' mean loss is the combined batch's loss up to the rounding of two means
    ok = same and norms and abs(mean_loss[0] - single[2][0][0]) <= 1e-4 * abs(single[2][0][0])
    sys.exit(0 if ok else 1)
