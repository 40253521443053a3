"""Data parallelism for the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  The reference's only parallelism is Lightning DDP (`strategy: ddp_find_unused_parameters_true`,
configs/diffusionGS_rel.yaml:80): per-rank batches, ONE exchange step -- the gradient all-reduce of the 460 M DiT
parameters.  The rasterizer and inference have no collective.

MI355X-first choices (xGMI is a point-to-point mesh, ring collectives are per-link bound, HBM is plentiful):
  * gradients live in ONE flat fp32 buffer (`FlatGrads`); the backward writes every parameter's gradient straight into
    its slice, so there is no per-tensor copy-in / copy-out around the collective;
  * the buffer is reduced in a few LARGE buckets (default 256 MiB, i.e. ~8 collectives for 1.84 GB instead of DDP's ~74
    of 25 MB) -- large messages are what saturates all seven links;
  * a bucket is enqueued the moment the blocks that fill it have finished their backward (reverse layer order): the C
    backward calls back after every block (DgsDitBackwardArgs.block_done -> DataParallelTrainer._on_gradients_final ->
    `ready_up_to`), the collective runs on RCCL's own stream behind the compute stream's work so far, and so overlaps the
    backward of the earlier blocks.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(device=None, backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).  Returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


class FlatGrads:
    """One contiguous fp32 gradient buffer with a named view per parameter, laid out in BACKWARD completion order
    (heads first, block L-1 ... block 0, embedding last) so that finished prefixes are contiguous buckets."""

    def __init__(self, named_shapes, device, dtype=torch.float32):
        self.names = [n for n, _ in named_shapes]
        self.offsets, off = {}, 0
        for n, shape in named_shapes:
            numel = 1
            for s in shape:
                numel *= int(s)
            self.offsets[n] = (off, numel, tuple(shape))
            off += (numel + 63) // 64 * 64          # 256-byte aligned slices
        self.flat = torch.zeros(off, dtype=dtype, device=device)

    def view(self, name):
        off, numel, shape = self.offsets[name]
        return self.flat[off:off + numel].view(shape)

    def end_of(self, name):
        off, numel, _ = self.offsets[name]
        return (off + numel + 63) // 64 * 64

    def zero_(self):
        self.flat.zero_()


class BucketedAllReduce:
    """Average `flat` over the ranks in large buckets, each enqueued as soon as the caller says its bytes are final."""

    def __init__(self, flat, bucket_bytes=256 << 20, group=None):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        per = max(1, bucket_bytes // flat.element_size())
        n = flat.numel()
        self.bounds = [(a, min(n, a + per)) for a in range(0, n, per)]
        self.next_bucket, self.works = 0, []
        self.launch_log = []          # (bucket index, tag) of the last step, in launch order (tests, diagnostics)

    def ready_up_to(self, end_element, tag=None):
        """Everything in flat[:end_element] is final (once the work enqueued on the current stream so far has run): launch
        every not-yet-launched bucket that lies inside it.  `tag` only labels the launch in `launch_log`."""
        if self.next_bucket == 0:
            self.launch_log = []
        while self.next_bucket < len(self.bounds) and self.bounds[self.next_bucket][1] <= end_element:
            a, b = self.bounds[self.next_bucket]
            if self.world > 1:
                self.works.append(dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.launch_log.append((self.next_bucket, tag))
            self.next_bucket += 1

    def finish(self, average=True):
        """Launch what is left, wait for everything; average=True turns sums into means (a caller that already folded
        1 / world into its loss scale passes False and saves the pass over the buffer).  Resets for the next step."""
        self.ready_up_to(self.flat.numel(), tag="finish")
        for w in self.works:
            w.wait()
        if self.world > 1 and average:
            self.flat.mul_(1.0 / self.world)
        self.works, self.next_bucket = [], 0
