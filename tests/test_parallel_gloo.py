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
