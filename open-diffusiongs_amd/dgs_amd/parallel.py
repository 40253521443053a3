"""Data parallelism for the hot path: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  The reference's only parallelism is Lightning DDP (`strategy: ddp_find_unused_parameters_true`,
configs/diffusionGS_rel.yaml:80): per-rank batches, ONE exchange step -- the gradient all-reduce of the 460 M DiT
parameters.  The rasterizer and inference have no collective.

MI355X-first choices (xGMI is a point-to-point mesh, ring collectives are per-link bound, HBM is plentiful):
  * gradients live in ONE flat fp32 buffer (`FlatGrads`); the backward writes every parameter's gradient straight into
    its slice, so there is no per-tensor copy-in / copy-out around the collective;
  * the buffer is reduced in a few LARGE buckets (32 MiB per rank: 256 MiB at 8 GPUs, i.e. ~9 collectives for 1.84 GB instead of
    DDP's ~74 of 25 MB) -- large messages are what saturates all seven links -- except at the END of the buffer, where the
    buckets shrink (128, 64, 32, 32 MiB): the last bucket's collective is the only one nothing overlaps;
  * every block's adaLN Linear gradient (a third of all parameters) is laid out WITH its block and is final with it
    (dgs_dit_backward computes it per block), so what is left for the end of the backward is ~2 M parameters;
  * optional bf16 exchange (`compress="bf16"`): a bucket is cast to bf16, summed, and written back as fp32 -- half the xGMI bytes
    for gradients that were computed from bf16 operands anyway; off by default (the reference's DDP exchanges fp32);
  * a bucket is enqueued the moment the blocks that fill it have finished their backward (reverse layer order): the C
    backward calls back after every block (DgsDitBackwardArgs.block_done -> DataParallelTrainer._on_gradients_final ->
    `ready_up_to`), the collective runs on RCCL's own stream behind the compute stream's work so far, and so overlaps the
    backward of the earlier blocks.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(device=None, backend=None, force=False):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).  Returns (rank, world, local_rank).
    force=True creates the process group even for a world of one (the one-GPU RCCL smoke test: the same init, stream and
    collective calls as N > 1, on the only hardware the builder's box has)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
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


def bucket_bounds(n, per, tail=()):
    """[(begin, end)) element ranges covering [0, n): `tail` sizes (elements, in buffer order) at the END of the buffer, uniform
    buckets of `per` elements in front of them."""
    tail = [t for t in tail if t > 0]
    while tail and sum(tail) > n // 2:        # small buffers: no special tail
        tail = tail[1:]
    head_end = n - sum(tail)
    bounds = [(a, min(head_end, a + per)) for a in range(0, head_end, per)]
    a = head_end
    for t in tail:
        bounds.append((a, a + t))
        a += t
    return bounds


class BucketedAllReduce:
    """Average `flat` over the ranks in large buckets, each enqueued as soon as the caller says its bytes are final."""

    def __init__(self, flat, bucket_bytes=None, group=None, compress=None, force_collectives=False):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a world of one normally skips the collectives; force_collectives issues them anyway (a sum over one rank: the RCCL
        # launch path, its stream ordering against the backward and the bf16 round trip run for real on a single GPU)
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        if bucket_bytes is None:                   # 32 MiB per rank: a ring step moves bucket / world per link
            bucket_bytes = (32 << 20) * max(self.world, 2)
        es = flat.element_size()
        per = max(1, bucket_bytes // es)
        # the buffer is in backward-completion order: the LAST bucket's collective starts when the backward ends and nothing
        # overlaps it -- keep it (and its predecessors, which have little backward left to hide behind) small
        tail = [min(per, (m << 20) // es) for m in (128, 64, 32, 32)] if bucket_bytes > (32 << 20) else []
        self.bounds = bucket_bounds(flat.numel(), per, tail)
        assert self.bounds[0][0] == 0 and self.bounds[-1][1] == flat.numel() and all(a[1] == b[0] for a, b in zip(self.bounds, self.bounds[1:]))
        if compress not in (None, "bf16"):
            raise ValueError("compress: None or 'bf16'")
        self.compress = compress
        # world-of-one diagnostics: a buffer that receives every bucket THROUGH a collective that moves data on RCCL's stream
        # (an all-gather over one rank = a copy there) at the moment the bucket is launched -- equal to `flat` after the step
        # iff every bucket was final when its collective ran (set by the test: tests/test_rccl_world1_gpu.py)
        self.probe = None
        self.next_bucket, self.works = 0, []
        self.launch_log = []          # (bucket index, tag) of the last step, in launch order (tests, diagnostics)

    def ready_up_to(self, end_element, tag=None):
        """Everything in flat[:end_element] is final (once the work enqueued on the current stream so far has run): launch
        every not-yet-launched bucket that lies inside it.  `tag` only labels the launch in `launch_log`."""
        if self.next_bucket == 0:
            self.launch_log = []
        while self.next_bucket < len(self.bounds) and self.bounds[self.next_bucket][1] <= end_element:
            a, b = self.bounds[self.next_bucket]
            if self.active:
                if self.probe is not None:
                    assert self.world == 1, "the launch-time probe is a one-rank diagnostic"
                    self.works.append((dist.all_gather_into_tensor(self.probe[a:b], self.flat[a:b], group=self.group, async_op=True), None, a, b))
                if self.compress == "bf16":
                    half = self.flat[a:b].to(torch.bfloat16)            # on the compute stream, behind the kernels that fill [a, b)
                    self.works.append((dist.all_reduce(half, op=dist.ReduceOp.SUM, group=self.group, async_op=True), half, a, b))
                else:
                    self.works.append((dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True), None, a, b))
            self.launch_log.append((self.next_bucket, tag))
            self.next_bucket += 1

    def finish(self, average=True):
        """Launch what is left, wait for everything; average=True turns sums into means (a caller that already folded
        1 / world into its loss scale passes False and saves the pass over the buffer).  Resets for the next step."""
        self.ready_up_to(self.flat.numel(), tag="finish")
        for w, half, a, b in self.works:
            w.wait()
            if half is not None:
                self.flat[a:b].copy_(half)
        if self.world > 1 and average:
            self.flat.mul_(1.0 / self.world)
        self.works, self.next_bucket = [], 0
