"""AdamW step + refresh of the engine's weight copies in ONE launch (include/dgs_optim.h, csrc/optim.hip).

The reference trains with `torch.optim.AdamW(lr=1e-5, betas=[0.9, 0.99], eps=1e-8)` (diffusionGS/configs/diffusionGS_rel.yaml:57-62,
diffusionGS/utils/scheduler.py:34-53).  On a bf16-MFMA engine an optimizer step is followed by bringing every device-resident operand
copy of the weights up to date; as torch ops that is a multi-tensor AdamW (one launch per ~50 tensors) plus ~600 cast /
transposed-copy / copy launches per step (3.2 + 1.2 ms of a 91 ms step, `profiles/r03_final_train_step_kernel_stats.txt`).
`FusedAdamW.step()` reads p, g, m, v once and writes p, m, v and every copy: same arithmetic as torch's single-tensor AdamW in fp32.

A `torch.optim.Optimizer` (one parameter group: lr / betas / eps / weight_decay may be changed between steps), so the reference's
`CosineAnnealingLR(optim, ...)` (configs/diffusionGS_rel.yaml:64-68, utils/scheduler.py `parse_scheduler`) and Lightning's checkpointing
take it as they take torch's: `.step()`, `.zero_grad()`, `.param_groups`, `.state_dict()` / `.load_state_dict()`.
"""
import ctypes
import math

import torch

from . import _native
from .dit import _stream


class FusedAdamW(torch.optim.Optimizer):
    refreshes_engine = True          # DataParallelTrainer: the engine's weight copies are written by step() itself

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(list(model.parameters()), dict(lr=float(lr), betas=(float(betas[0]), float(betas[1])), eps=float(eps),
                                                        weight_decay=float(weight_decay)))
        self.model = model
        self.lib = getattr(model, "_lib", None) or _native.lib()
        self.step_count = 0
        self._named = [(n, p) for n, p in model.named_parameters()]
        dev = self._named[0][1].device
        sizes = [(p.numel() + 63) // 64 * 64 for _, p in self._named]          # 256-byte aligned slices
        self.exp_avg = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self._offsets, o = [], 0
        for s in sizes:
            self._offsets.append(o)
            o += s
        self._table = None          # (device table, n_tensors, n_tiles, the pointers it was built from)

    # -- torch.optim.Optimizer surface the trainer / a scheduler touches ---------------------------------
    def zero_grad(self, set_to_none=True):
        """Under DataParallelTrainer the backward OVERWRITES every gradient in the flat buffer (dgs_dit_backward writes, never
        accumulates): nothing to clear, and the `.grad` views must stay.  Anywhere else: torch's semantics."""
        if not getattr(self.model, "_grads_in_place", False):
            super().zero_grad(set_to_none=set_to_none)

    def state_dict(self):
        """torch's layout -- {'state': {param index: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [{..., 'params': [indices]}]} --
        so that generic tooling (Lightning checkpoints, optimizer-state inspection) reads it like torch.optim.AdamW's.  The moments
        themselves live in two flat buffers (the layout follows model.named_parameters()); the entries are copies of their slices."""
        state = {}
        for i, ((_, p), off) in enumerate(zip(self._named, self._offsets)):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[off:off + n].reshape(p.shape).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].reshape(p.shape).clone()}
        groups = [dict({k: v for k, v in g.items() if k != "params"}, params=list(range(len(self._named)))) for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        if "state" not in sd:             # round 3's private layout: flat buffers
            self.step_count = int(sd["step"])
            self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        else:
            steps = set()
            for i, ((_, p), off) in enumerate(zip(self._named, self._offsets)):
                e = sd["state"].get(i, sd["state"].get(str(i)))
                if e is None:
                    continue
                n = p.numel()
                self.exp_avg[off:off + n].copy_(e["exp_avg"].reshape(-1)); self.exp_avg_sq[off:off + n].copy_(e["exp_avg_sq"].reshape(-1))
                steps.add(int(float(e["step"])))
            if len(steps) > 1:
                raise ValueError("FusedAdamW: one step count for all parameters (per-parameter counts differ in this state dict)")
            self.step_count = steps.pop() if steps else 0
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in s.items() if k != "params"})

    # -- the table of tensors (rebuilt when a pointer changed: new .grad tensors, a rebuilt engine) --------
    def _plan(self):
        eng = self.model.engine()
        ptrs = (id(eng), id(eng._train)) + tuple((p.data_ptr(), p.grad.data_ptr() if p.grad is not None else 0) for _, p in self._named)
        if self._table is not None and self._table[3] == ptrs:
            return self._table
        dst = eng.weight_destinations()
        entries = []
        for (name, p), off in zip(self._named, self._offsets):
            if p.grad is None:
                continue                          # a parameter that received no gradient is left alone, like torch.optim
            if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                raise RuntimeError(f"FusedAdamW: {name}: fp32 contiguous master parameters and gradients only")
            e = _native.DgsAdamWTensor()
            e.p, e.g = p.data_ptr(), p.grad.data_ptr()
            e.m, e.v = self.exp_avg.data_ptr() + 4 * off, self.exp_avg_sq.data_ptr() + 4 * off
            rows, cols = (int(p.shape[0]), p.numel() // int(p.shape[0])) if p.dim() >= 2 else (1, p.numel())
            e.rows, e.cols = rows, cols
            copy, copy_t = dst.get(name, (None, None))
            if copy is not None:
                if copy.numel() != p.numel() or not copy.is_contiguous():
                    raise RuntimeError(f"FusedAdamW: engine copy of {name} has another layout")
                e.copy = copy.data_ptr()
                e.copy_kind = _native.OPTIM_COPY_BF16 if copy.dtype == torch.bfloat16 else _native.OPTIM_COPY_F32
            if copy_t is not None:
                if tuple(copy_t.shape) != (cols, rows) or copy_t.dtype != torch.bfloat16 or not copy_t.is_contiguous():
                    raise RuntimeError(f"FusedAdamW: transposed engine copy of {name} has another layout")
                e.copy_t = copy_t.data_ptr()
            entries.append(e)
        host = (_native.DgsAdamWTensor * len(entries))(*entries)
        n_tiles = int(self.lib.dgs_adamw_plan(host, len(entries)))
        if n_tiles <= 0:
            raise RuntimeError("FusedAdamW: dgs_adamw_plan rejected the tensor table")
        raw = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.exp_avg.device)
        self._table = (raw, len(entries), n_tiles, ptrs)
        return self._table

    def step(self, closure=None, grad_sumsq=None, max_grad_norm=None):
        """closure: torch.optim.Optimizer's protocol (Lightning's loop and precision plugins call `optimizer.step(closure=...)`): run
        under grad mode first, its loss returned.  grad_sumsq (device float[1]) + max_grad_norm: the global-norm clip folded into the
        launch (dgs_optim.h; DataParallelTrainer computes the sum of squares bucket by bucket as the gradients become final)."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        with torch.no_grad():
            self._step(grad_sumsq, max_grad_norm)
        return loss

    def _settle_previous(self):
        """The launch skips an update whose gradient norm is not finite (csrc/optim.hip) -- GradScaler does not call optimizer.step() at
        all for such a step, so the step count and with it the bias corrections must not move either.  The previous step's norm was
        copied to pinned memory behind its launch: by now it has long arrived (the wait is for THAT copy, not for the stream)."""
        probe, self._skip_probe = getattr(self, "_skip_probe", None), None
        if probe is None:
            return
        host, ev = probe
        ev.synchronize()
        if not math.isfinite(float(host[0])):
            self.step_count -= 1
            self.skipped_steps = getattr(self, "skipped_steps", 0) + 1

    def _step(self, grad_sumsq, max_grad_norm):
        g = self.param_groups[0]
        raw, n, n_tiles, _ = self._plan()
        self._settle_previous()
        self.step_count += 1
        b1, b2 = g["betas"]
        a = _native.DgsAdamWArgs()
        a.tensors, a.n_tensors, a.n_tiles = raw.data_ptr(), n, n_tiles
        a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = g["lr"], b1, b2, g["eps"], g["weight_decay"]
        a.bias_correction1 = 1.0 - b1 ** self.step_count
        a.bias_correction2_sqrt = math.sqrt(1.0 - b2 ** self.step_count)
        if max_grad_norm is not None and max_grad_norm > 0:
            if grad_sumsq is None:
                raise ValueError("FusedAdamW.step: max_grad_norm needs grad_sumsq (the device word holding the gradients' sum of squares)")
            self._keep_sumsq = grad_sumsq
            a.grad_sumsq, a.max_grad_norm = grad_sumsq.data_ptr(), float(max_grad_norm)
        rc = self.lib.dgs_adamw_step(ctypes.byref(a), _stream(self.exp_avg.device))
        if rc != 0:
            raise RuntimeError(f"dgs_adamw_step: {_native.status_string(self.lib, rc)} (status {rc})")
        if a.grad_sumsq and grad_sumsq.device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            if getattr(self, "_skip_host", None) is None:
                self._skip_host = torch.zeros(1, dtype=torch.float32).pin_memory()
            self._skip_host.copy_(grad_sumsq.reshape(-1)[:1], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._skip_probe = (self._skip_host, ev)
        elif a.grad_sumsq and grad_sumsq.device.type != "cuda" and not math.isfinite(float(grad_sumsq.reshape(-1)[0])):
            self.step_count -= 1                       # CPU emulation build: the word is at hand
            self.skipped_steps = getattr(self, "skipped_steps", 0) + 1
        # the parameters were written through raw pointers: their version counters did not move, and the engine's copies are
        # already up to date -- DGSDenoiser.engine() must not refresh them again
