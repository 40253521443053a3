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


def broadcast_parameters(module, src=0, group=None, bucket_bytes=256 << 20):
    """DDP's init-time parameter (and buffer) broadcast (torch DistributedDataParallel._sync_module_states; Lightning wraps the
    model in DDP before the first step): every rank leaves with rank `src`'s values, whatever it was constructed or loaded with.
    Few large messages (xGMI: large messages fill all links): tensors are packed into flat buckets of `bucket_bytes`."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    tensors = [p.data for p in module.parameters()] + [b.data for b in module.buffers()]
    sent, i = 0, 0
    while i < len(tensors):
        dtype, dev = tensors[i].dtype, tensors[i].device
        chunk, nbytes = [], 0
        while i < len(tensors) and tensors[i].dtype == dtype and tensors[i].device == dev and (not chunk or nbytes + tensors[i].numel() * tensors[i].element_size() <= bucket_bytes):
            chunk.append(tensors[i]); nbytes += tensors[i].numel() * tensors[i].element_size(); i += 1
        flat = torch.cat([t.reshape(-1) for t in chunk])
        dist.broadcast(flat, src=src, group=group)
        o = 0
        for t in chunk:
            t.copy_(flat[o:o + t.numel()].view_as(t)); o += t.numel()
        sent += nbytes
    return sent


class GradNorm:
    """Sum of squares of the flat gradient buffer for the global-norm clip (torch.nn.utils.clip_grad_norm_ / Lightning
    `gradient_clip_val`, configs/diffusionGS_rel.yaml:76-77), computed bucket by bucket as each bucket becomes final (behind its
    all-reduce, on the reducer's side stream: hidden behind the backward) -- not as another pass over 1.84 GB at the end.  Two
    deterministic stages (include/dgs_optim.h): one partial per 65,536 elements, written; `total()` adds them in index order."""
    CHUNK = 65536

    def __init__(self, flat, lib):
        import ctypes
        self.flat, self.lib, self._c = flat, lib, ctypes
        self.count = int(lib.dgs_sumsq_count(flat.numel()))
        self.partials = torch.zeros(self.count, dtype=torch.float32, device=flat.device)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=flat.device)

    def _stream(self):
        c = self._c
        return c.c_void_p(torch.cuda.current_stream(self.flat.device).cuda_stream) if self.flat.is_cuda else None

    def add(self, a, b):
        """flat[a:b] is final: (re)write its partials.  a is a multiple of 65,536 (every bucket boundary but the buffer's end is)."""
        assert a % self.CHUNK == 0 and b > a
        c = self._c
        rc = self.lib.dgs_sumsq_partials(c.c_void_p(self.flat.data_ptr() + 4 * a), b - a, c.c_void_p(self.partials.data_ptr() + 4 * (a // self.CHUNK)), self._stream())
        if rc != 0:
            raise RuntimeError(f"dgs_sumsq_partials: status {rc}")

    def total(self):
        """-> device float[1]: sum of squares over the whole buffer (every bucket added since the last call)."""
        c = self._c
        rc = self.lib.dgs_sumsq_finish(c.c_void_p(self.partials.data_ptr()), self.count, c.c_void_p(self.sumsq.data_ptr()), self._stream())
        if rc != 0:
            raise RuntimeError(f"dgs_sumsq_finish: status {rc}")
        return self.sumsq


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

    def __init__(self, flat, bucket_bytes=None, group=None, compress=None, force_collectives=False, norm=None):
        self.flat, self.group = flat, group
        self.norm = norm              # GradNorm: partial sums of squares of each bucket as it becomes final (global-norm clip)
        self.side = torch.cuda.Stream(flat.device) if flat.is_cuda else None      # behind-the-collective work: bf16 copy-back, norm partials
        self._deferred = []           # CPU tensors (gloo tests): the same work, done in finish() behind each wait
        self.timing = None            # set to {} by a caller that wants `exposed_ms` (GPU time finish() waited for collectives)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # a world of one normally skips the collectives; force_collectives issues them anyway (a sum over one rank: the RCCL
        # launch path, its stream ordering against the backward and the bf16 round trip run for real on a single GPU)
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        import os
        self._norm_on_side = os.environ.get("DGS_NORM_SIDE_STREAM", "0") not in ("", "0")
        if bucket_bytes is None:                   # 32 MiB per rank: a ring step moves bucket / world per link
            bucket_bytes = (32 << 20) * max(self.world, 2)
        es = flat.element_size()
        per = max(1, bucket_bytes // es)
        # the buffer is in backward-completion order: the LAST bucket's collective starts when the backward ends and nothing
        # overlaps it -- keep it (and its predecessors, which have little backward left to hide behind) small
        tail = [min(per, (m << 20) // es) for m in (128, 64, 32, 32)] if bucket_bytes > (32 << 20) else []
        if norm is not None:          # bucket starts on partial boundaries
            per = max(GradNorm.CHUNK, per // GradNorm.CHUNK * GradNorm.CHUNK)
            tail = [max(GradNorm.CHUNK, t // GradNorm.CHUNK * GradNorm.CHUNK) for t in tail]
        if norm is not None:          # every bucket STARTS on a partial boundary: lay the buckets out on the length rounded up, clamp the end
            n = flat.numel()
            up = -(-n // GradNorm.CHUNK) * GradNorm.CHUNK
            self.bounds = [(a, min(b, n)) for a, b in bucket_bounds(up, per, tail) if a < n]
        else:
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
            work, half = None, None
            if self.active:
                if self.probe is not None:
                    assert self.world == 1, "the launch-time probe is a one-rank diagnostic"
                    self.works.append(dist.all_gather_into_tensor(self.probe[a:b], self.flat[a:b], group=self.group, async_op=True))
                if self.compress == "bf16":
                    half = self.flat[a:b].to(torch.bfloat16)            # on the compute stream, behind the kernels that fill [a, b)
                    work = dist.all_reduce(half, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                else:
                    work = dist.all_reduce(self.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._after_bucket(work, half, a, b)
            self.launch_log.append((self.next_bucket, tag))
            self.next_bucket += 1

    def _after_bucket(self, work, half, a, b):
        """What follows a bucket's collective -- the bf16 copy-back (the half-size copy is released right there) and the bucket's
        norm partials -- on a SIDE stream that waits for the collective: the compute stream (the backward of the earlier blocks)
        is not held up.  finish() joins the side stream."""
        if work is None and half is None and self.norm is None:
            return
        if work is None and half is None and not self._norm_on_side:
            # a world of one without forced collectives: nothing to wait for, the bucket's norm partials go behind the kernels that
            # filled it on THIS stream -- a second stream beside the backward's one-round kernels costs them more than its 29 small
            # launches hide (profiles/r06_norm_stream_ab.txt; DGS_NORM_SIDE_STREAM=1: the side stream anyway, measurement aid)
            self.norm.add(a, b)
            return
        if self.side is None:
            self._deferred.append((work, half, a, b))
            return
        cur = torch.cuda.current_stream(self.flat.device)
        self.side.wait_stream(cur)                                     # a world of one: the bucket is final behind the kernels enqueued so far
        with torch.cuda.stream(self.side):
            if work is not None:
                work.wait()                                            # stream-side wait: the host does not block
            if half is not None:
                self.flat[a:b].copy_(half)
                half.record_stream(self.side)
            if self.norm is not None:
                self.norm.add(a, b)

    def finish(self, average=True):
        """Launch what is left, wait for everything; average=True turns sums into means (a caller that already folded
        1 / world into its loss scale passes False and saves the pass over the buffer).  Resets for the next step."""
        self.ready_up_to(self.flat.numel(), tag="finish")
        e0 = e1 = None
        if self.timing is not None and self.side is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                                                # behind the last kernel of the backward
        for w in self.works:                                           # the one-rank probe's gathers
            w.wait()
        for work, half, a, b in self._deferred:
            if work is not None:
                work.wait()
            if half is not None:
                self.flat[a:b].copy_(half)
            if self.norm is not None:
                self.norm.add(a, b)
        if self.side is not None:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.side)
        if e1 is not None:
            e1.record()                                                # [e0, e1] on the compute stream = time it idled for collectives
            self.timing.setdefault("events", []).append((e0, e1))
        if self.world > 1 and average:
            self.flat.mul_(1.0 / self.world)
        self.works, self._deferred, self.next_bucket = [], [], 0

    def exposed_ms(self):
        """Mean GPU time per step the compute stream waited in finish() for collectives (and the work behind them) that the backward
        did not hide.  Needs `timing = {}` before the steps; synchronises."""
        ev = (self.timing or {}).get("events", [])
        if not ev:
            return None
        torch.cuda.synchronize(self.flat.device)
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev)
