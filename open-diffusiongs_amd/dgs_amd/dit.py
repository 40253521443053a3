"""Torch-tensor front end of the DiT C ABI (include/dgs_dit.h).

`DitEngine` owns the bf16 device copy of a DGSDenoiser state dict (reference key names) plus one
workspace, and runs `DGSDenoiser.image_to_gaussians` (denoiser.py:306-416) as ONE C call that enqueues every kernel on
the current HIP stream.  `DitOps` exposes the individual kernels for the parity tests.  PyTorch provides device memory
and the stream only -- no torch op computes anything on this path.
"""
import ctypes

import torch

from . import _native
from ._native import (DgsDitAttentionArgs, DgsDitForwardArgs, DgsDitGemmArgs, DgsDitLayerNormArgs, DgsDitLayerWeights,
                      DgsDitModel, DgsDitRowLinearArgs)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream(device):
    if device.type == "cuda":
        return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    return None


class DitOps:
    """Kernel-level entry points (tests, microbenchmarks)."""

    def __init__(self, lib=None):
        self.lib = lib if lib is not None else _native.lib()

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(f"dgs dit: {_native.status_string(self.lib, rc)} (status {rc})")

    def poison_lds(self, device="cuda:0"):
        """Test hook (dgs_debug_poison_lds): NaN patterns into every CU's LDS, so that a fragment read that overtakes its
        LDS-DMA cannot pass on the previous launch's data."""
        self._check(self.lib.dgs_debug_poison_lds(ctypes.c_void_p(torch.cuda.current_stream(torch.device(device)).cuda_stream)))

    def shader_clock_mhz(self, device="cuda:0", microseconds=20):
        """Measurement hook (dgs_debug_clock_probe): the effective shader clock where the stream stands, from the two hardware
        counters (s_memtime: shader clock, s_memrealtime: constant 100 MHz).  Synchronises."""
        out = torch.zeros(3, dtype=torch.int64, device=device)
        self._check(self.lib.dgs_debug_clock_probe(_p(out), int(microseconds), _stream(torch.device(device))))
        c, w, _ = out.tolist()
        return 100.0 * c / w if w > 0 else 0.0

    def gemm(self, A, W, bias=None, epilogue=_native.EPI_BF16, out=None, gate=None, rows_per_batch=0, vt=None, valid_rows=0,
             resid=None, aux=None, shape=None, k_per_batch=0, a_batch_stride=0, w_batch_stride=0, lda=None, ldw=None, algo=0,
             q_scale=0.0, splitk=False):
        """A bf16 [M,K], W bf16 [N,K] -> per epilogue (see dgs_dit.h).  `out` is required for GATE_RESIDUAL (in-place).
        splitk: hand the library its split-K scratch (weight-gradient shapes; a no-op where no split applies)."""
        if shape is not None:          # batched-reduction form (weight gradients): operands are [batch, rows, tokens]
            M, N, K = shape
        else:
            M, K = A.shape
            N = W.shape[0]
        dev = A.device
        if epilogue == _native.EPI_QKV:
            ldo = 2 * N // 3
            if out is None:      # zero-filled: padding rows are never written and must be finite (dgs_dit.h "padding contract")
                out = torch.zeros((M, ldo), dtype=torch.bfloat16, device=dev)
            if vt is None:
                vt = torch.zeros((M // rows_per_batch, N // 3, rows_per_batch), dtype=torch.bfloat16, device=dev)
        elif epilogue in (_native.EPI_BF16, _native.EPI_GELU_BF16, _native.EPI_DGELU_BF16):
            ldo = N
            if out is None:
                out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
        else:
            ldo = N
            if out is None:
                out = torch.zeros((M, N), dtype=torch.float32, device=dev)
        a = DgsDitGemmArgs()
        a.M, a.N, a.K = M, N, K
        a.A, a.lda, a.W, a.ldw = _p(A), (lda if lda is not None else A.stride(0)), _p(W), (ldw if ldw is not None else W.stride(0))
        a.resid, a.aux = _p(resid), _p(aux)
        a.k_per_batch, a.a_batch_stride, a.w_batch_stride = k_per_batch, a_batch_stride, w_batch_stride
        a.bias, a.epilogue, a.out, a.ldo = _p(bias), epilogue, _p(out), ldo
        a.gate, a.gate_stride, a.rows_per_batch, a.vt = _p(gate), (gate.stride(0) if gate is not None else 0), rows_per_batch, _p(vt)
        a.valid_rows, a.algo, a.q_scale = valid_rows, algo, q_scale
        if splitk:
            nbytes = self.lib.dgs_dit_gemm_splitk_bytes(M, N, K, k_per_batch)
            ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=dev)
            a.splitk_ws = _p(ws) if nbytes else None
        self.last_gemm_sliced_tile = int(self.lib.dgs_dit_gemm_sliced_tile(ctypes.byref(a)))
        self._check(self.lib.dgs_dit_gemm(ctypes.byref(a), _stream(dev)))
        return (out, vt) if epilogue == _native.EPI_QKV else out

    def attention(self, qk, vt, L, heads, qkv_layout=False, lse2=None, q_prescaled=False, tail_mode=0, out=None):
        """qk bf16 [B*lpad, 2*heads*64], vt bf16 [B, heads*64, lpad] -> bf16 [B*lpad, heads*64].
        qkv_layout: `qk` is the training tensor [B*lpad, 3W] and `vt` its transposed copy [B, 3W, lpad].
        tail_mode: reserved field of dgs_dit.h (0; the library rejects anything else)."""
        B, _, lpad = vt.shape
        W = heads * 64
        if out is None:
            out = torch.zeros((B * lpad, W), dtype=torch.bfloat16, device=qk.device)
        a = DgsDitAttentionArgs()
        a.B, a.heads, a.L, a.lpad = B, heads, L, lpad
        a.qk, a.vt, a.out, a.scale = _p(qk), _p(vt), _p(out), 0.125
        if qkv_layout:
            a.ld_qk, a.k_offset, a.vt_batch_stride = 3 * W, W, 3 * W * lpad
            a.vt = ctypes.c_void_p(vt.data_ptr() + 2 * W * lpad * 2)
        a.lse2, a.q_prescaled, a.tail_mode = _p(lse2), int(q_prescaled), int(tail_mode)
        nb = int(self.lib.dgs_dit_attention_tail_bytes(B, heads, L))
        if nb:      # caller-owned scratch of the L % 32 tail queries (zero-filled: holds the arrival counters)
            tail = torch.zeros(nb, dtype=torch.uint8, device=qk.device)
            a.tail_ws, a.tail_ws_bytes = _p(tail), nb
        self._check(self.lib.dgs_dit_attention(ctypes.byref(a), _stream(qk.device)))
        return out

    def attention_backward(self, qkv, qkvT, o, dO, dOT, lse2, L, heads, byproducts=False):
        """-> dqkv bf16 [B*lpad, 3W] (dq | dk | dv).  byproducts=True: -> (dqkv, dqkvT bf16 [B, 3W, lpad] -- padding tokens left zero --,
        bias_part f32 [B * slots, 3W]: per-workgroup column sums over valid tokens; dgs_dit.h)."""
        B, _, lpad = qkvT.shape
        dqkv = torch.zeros_like(qkv)
        D = torch.zeros_like(lse2)
        a = _native.DgsDitAttentionBackwardArgs()
        a.B, a.heads, a.L, a.lpad = B, heads, L, lpad
        a.qkv, a.qkvT, a.o, a.dO, a.dOT, a.lse2, a.D, a.dqkv = (_p(t) for t in (qkv, qkvT, o, dO, dOT, lse2, D, dqkv))
        a.scale = 0.125
        dqkvT = part = None
        if byproducts:
            dqkvT = torch.zeros_like(qkvT)
            part = torch.full((B * int(self.lib.dgs_dit_attention_backward_slots(L)), 3 * heads * 64), float("nan"), dtype=torch.float32, device=qkv.device)
            a.dqkvT, a.bias_part = _p(dqkvT), _p(part)
        self._check(self.lib.dgs_dit_attention_backward(ctypes.byref(a), _stream(qkv.device)))
        return (dqkv, dqkvT, part) if byproducts else dqkv

    def layernorm(self, x, weight=None, shift=None, scale=None, rows_per_batch=0, eps=1e-6, out_f32=False):
        rows, width = x.shape
        out = torch.empty((rows, width), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
        a = DgsDitLayerNormArgs()
        a.rows, a.width, a.x, a.weight = rows, width, _p(x), _p(weight)
        a.shift, a.scale = _p(shift), _p(scale)
        a.mod_stride = shift.stride(0) if shift is not None else 0
        a.rows_per_batch, a.eps, a.out, a.out_f32 = rows_per_batch, eps, _p(out), int(out_f32)
        self._check(self.lib.dgs_dit_layernorm(ctypes.byref(a), _stream(x.device)))
        return out

    def layernorm_gemm(self, x, shift, scale, W, bias, epilogue, rows_per_batch, valid_rows, eps=1e-6, algo=0, q_scale=0.0, aux=None):
        """dgs_dit_layernorm_gemm: modulate(LayerNorm(x)) and the QKV / fc1 GEMM that consumes it; returns (ln_out, gemm outputs)."""
        rows, width = x.shape
        N, dev = W.shape[0], x.device
        h = torch.zeros((rows, width), dtype=torch.bfloat16, device=dev)
        l = DgsDitLayerNormArgs()
        l.rows, l.width, l.x, l.shift, l.scale, l.mod_stride = rows, width, _p(x), _p(shift), _p(scale), shift.stride(0)
        l.rows_per_batch, l.eps, l.out = rows_per_batch, eps, _p(h)
        qkv = epilogue == _native.EPI_QKV
        out = torch.zeros((rows, 2 * N // 3 if qkv else N), dtype=torch.bfloat16, device=dev)
        vt = torch.zeros((rows // rows_per_batch, N // 3, rows_per_batch), dtype=torch.bfloat16, device=dev) if qkv else None
        a = DgsDitGemmArgs()
        a.M, a.N, a.K, a.A, a.lda, a.W, a.ldw = rows, N, width, _p(h), width, _p(W), W.stride(0)
        a.bias, a.epilogue, a.out, a.ldo, a.vt, a.aux = _p(bias), epilogue, _p(out), out.shape[1], _p(vt), _p(aux)
        a.rows_per_batch, a.valid_rows, a.algo, a.q_scale = rows_per_batch, valid_rows, algo, q_scale
        self.last_pair_shared_rows = bool(self.lib.dgs_dit_layernorm_gemm_shares_rows(ctypes.byref(l), ctypes.byref(a)))
        self._check(self.lib.dgs_dit_layernorm_gemm(ctypes.byref(l), ctypes.byref(a), _stream(dev)))
        return (h, out, vt) if qkv else (h, out)

    def rowlinear(self, x, W, bias=None, silu_input=False, silu_output=False):
        M, K = x.shape
        N = W.shape[0]
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
        a = DgsDitRowLinearArgs()
        a.M, a.N, a.K, a.x, a.silu_input = M, N, K, _p(x), int(silu_input)
        a.W, a.bias, a.silu_output, a.out = _p(W), _p(bias), int(silu_output), _p(out)
        self._check(self.lib.dgs_dit_rowlinear(ctypes.byref(a), _stream(x.device)))
        return out


class DitEngine:
    """Device-resident DGSDenoiser weights + workspace; `image_to_gaussians` mirrors denoiser.py:306-416."""

    def __init__(self, state_dict, width=1024, patch_size=8, n_gaussians=2, dim_heads=64, num_layers=24, in_channels=9,
                 ray_pe_type="relative_plk", gaussians_sh_degree=0, scene=False, range_near=0.0, range_far=500.0,
                 device="cuda", lib=None):
        if gaussians_sh_degree not in (0, 1, 2, 3):
            raise ValueError("gaussians_sh_degree 0 .. 3 (what the rasterizer evaluates)")
        self.sh_degree = int(gaussians_sh_degree)
        self.lib = lib if lib is not None else _native.lib()
        self.device = torch.device(device)
        self.width, self.patch, self.ng, self.layers = width, patch_size, n_gaussians, num_layers
        self.heads = width // dim_heads
        self.gs_channels = 3 + (self.sh_degree + 1) ** 2 * 3 + 3 + 4 + 1     # to_gs's split, denoiser.py:96,109-111
        sd = state_dict
        bf = lambda k: torch.empty(tuple(sd[k].shape), dtype=torch.bfloat16, device=self.device)
        f32 = lambda k: torch.empty(tuple(sd[k].shape), dtype=torch.float32, device=self.device)
        keep = {}
        keep["t_w0"], keep["t_b0"] = bf("t_embedder.mlp.0.weight"), f32("t_embedder.mlp.0.bias")
        keep["t_w1"], keep["t_b1"] = bf("t_embedder.mlp.2.weight"), f32("t_embedder.mlp.2.bias")
        keep["tok_w"] = bf("image_tokenizer.1.weight")
        keep["pos_emb"] = torch.empty((n_gaussians, width), dtype=torch.float32, device=self.device)
        keep["in_ln_w"] = f32("transformer_input_layernorm.weight")
        self._layers = (DgsDitLayerWeights * num_layers)()
        for i in range(num_layers):
            p = f"transformer.{i}."
            for short, key in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
                keep[f"{i}.{short}_w"] = bf(p + key + ".weight")
                keep[f"{i}.{short}_b"] = f32(p + key + ".bias")
                setattr(self._layers[i], short + "_w", keep[f"{i}.{short}_w"].data_ptr())
                setattr(self._layers[i], short + "_b", keep[f"{i}.{short}_b"].data_ptr())
        nmod = (6 * num_layers + 4) * width
        keep["ada_w"] = torch.empty((nmod, width), dtype=torch.bfloat16, device=self.device)
        keep["ada_b"] = torch.empty((nmod,), dtype=torch.float32, device=self.device)
        keep["up_ln_w"], keep["up_w"] = f32("upsampler.layernorm.weight"), bf("upsampler.linear.weight")
        keep["dec_ln_w"], keep["dec_w"] = f32("image_token_decoder.layernorm.weight"), bf("image_token_decoder.linear.weight")
        self._keep = keep
        m = DgsDitModel()
        m.width, m.heads, m.layers, m.patch, m.in_channels = width, self.heads, num_layers, patch_size, in_channels
        m.n_gaussians, m.gs_channels, m.scene = n_gaussians, self.gs_channels, int(bool(scene))
        m.relative_plk = int(ray_pe_type == "relative_plk")
        m.range_near, m.range_far = float(range_near), float(range_far)
        for k in ("t_w0", "t_b0", "t_w1", "t_b1", "tok_w", "pos_emb", "in_ln_w", "ada_w", "ada_b", "up_ln_w", "up_w", "dec_ln_w", "dec_w"):
            setattr(m, k, keep[k].data_ptr())
        m.layer = ctypes.cast(self._layers, ctypes.POINTER(DgsDitLayerWeights))
        self.model = m
        self._ws, self._ws_shape = None, None
        self._train = None           # lazily built: transposed weights, gradient buffer, arenas
        self.refresh_weights(sd)

    # state-dict key of every engine tensor (adaLN tensors are stacked: see refresh_weights)
    _DIRECT = (("t_w0", "t_embedder.mlp.0.weight"), ("t_b0", "t_embedder.mlp.0.bias"), ("t_w1", "t_embedder.mlp.2.weight"),
               ("t_b1", "t_embedder.mlp.2.bias"), ("tok_w", "image_tokenizer.1.weight"), ("in_ln_w", "transformer_input_layernorm.weight"),
               ("up_ln_w", "upsampler.layernorm.weight"), ("up_w", "upsampler.linear.weight"),
               ("dec_ln_w", "image_token_decoder.layernorm.weight"), ("dec_w", "image_token_decoder.linear.weight"))

    @torch.no_grad()
    def weight_destinations(self):
        """state-dict key -> (engine tensor that holds its copy [same shape, bf16 or fp32], transposed bf16 copy or None): the ONE
        description of where a parameter's value has to go after it changed -- `refresh_weights` walks it with torch copies, the
        fused optimizer step (dgs_amd/optim.py) hands it to the kernel that updates the parameter."""
        k, W, L = self._keep, self.width, self.layers
        tk = self._train["tkeep"] if self._train is not None else None
        dst = {key: (k[name], None) for name, key in self._DIRECT}
        dst["gaussians_pos_embedding"] = (k["pos_emb"], None)
        for i in range(L):
            p = f"transformer.{i}."
            for short, key in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
                dst[p + key + ".weight"] = (k[f"{i}.{short}_w"], tk[f"{i}.{short}"] if tk is not None else None)
                dst[p + key + ".bias"] = (k[f"{i}.{short}_b"], None)
            dst[p + "adaLN_modulation.1.weight"] = (k["ada_w"][6 * W * i:6 * W * (i + 1)], None)
            dst[p + "adaLN_modulation.1.bias"] = (k["ada_b"][6 * W * i:6 * W * (i + 1)], None)
        o = 6 * W * L
        for j, head in enumerate(("upsampler", "image_token_decoder")):
            dst[head + ".adaLN_modulation.1.weight"] = (k["ada_w"][o + 2 * W * j:o + 2 * W * (j + 1)], None)
            dst[head + ".adaLN_modulation.1.bias"] = (k["ada_b"][o + 2 * W * j:o + 2 * W * (j + 1)], None)
        if tk is not None:
            dst["image_token_decoder.linear.weight"] = (k["dec_w"], tk["dec"])
        return dst

    def refresh_weights(self, state_dict):
        """Copy (and convert to bf16 / transpose) the current parameter values INTO the engine's existing device buffers:
        what has to happen after every optimizer step.  Pointers, workspaces, the activation arenas and the flat gradient
        buffer all stay as they are."""
        for key, (copy, copy_t) in self.weight_destinations().items():
            copy.copy_(state_dict[key].reshape(copy.shape))
            if copy_t is not None:
                copy_t.copy_(copy.t())

    def _workspace(self, B, V, H, W):
        need = int(self.lib.dgs_dit_workspace_bytes(ctypes.byref(self.model), B, V, H, W))
        if need == 0:
            raise RuntimeError("dgs dit: invalid shape for workspace")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.uint8, device=self.device)
            self._ws_shape = (B, V, H, W)
        elif self._ws_shape != (B, V, H, W):     # different carving: padding rows must be finite again
            self._ws.zero_()
            self._ws_shape = (B, V, H, W)
        return self._ws

    def num_tokens(self, V, H, W):
        return self.ng + V * (H // self.patch) * (W // self.patch)

    def image_to_gaussians(self, images, ray_o, ray_d, t, return_tokens=False, prof=None):
        """images/ray_o/ray_d [B,V,>=3|3,H,W] f32, t [B] int64 -> (dict(xyz, features, scaling, rotation, opacity), aligned_xyz)."""
        dev = self.device
        B, V, _, H, W = images.shape
        img = images[:, :, :3].to(dev, torch.float32).contiguous()
        ro, rd = ray_o.to(dev, torch.float32).contiguous(), ray_d.to(dev, torch.float32).contiguous()
        tt = t.to(dev, torch.int64).contiguous()
        P = self.ng + V * H * W
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        out = dict(xyz=f(B, P, 3), features=f(B, P, (self.sh_degree + 1) ** 2, 3), scaling=f(B, P, 3), rotation=f(B, P, 4), opacity=f(B, P, 1))
        aligned = f(B, V, 3, H, W)
        tokens = f(B, self.num_tokens(V, H, W), self.width) if return_tokens else None
        ws = self._workspace(B, V, H, W)
        a = DgsDitForwardArgs()
        a.B, a.V, a.H, a.W = B, V, H, W
        a.images, a.ray_o, a.ray_d, a.t = _p(img), _p(ro), _p(rd), _p(tt)
        a.workspace, a.workspace_bytes = _p(ws), ws.numel()
        a.xyz, a.features, a.scaling, a.rotation, a.opacity = (_p(out[k]) for k in ("xyz", "features", "scaling", "rotation", "opacity"))
        a.aligned_xyz, a.tokens = _p(aligned), _p(tokens)
        if prof is not None:   # (kind, [torch.cuda.Event(enable_timing=True), ...]): HIP events around one kernel class
            kind, events = prof
            handles = (ctypes.c_void_p * len(events))(*[e.cuda_event for e in events])
            count = ctypes.c_int32(0)
            a.prof_events = ctypes.cast(handles, ctypes.POINTER(ctypes.c_void_p))
            a.prof_kind, a.prof_capacity, a.prof_count = int(kind), len(events) // 2, ctypes.pointer(count)
        rc = self.lib.dgs_dit_forward(ctypes.byref(self.model), ctypes.byref(a), _stream(dev))
        if rc != 0:
            raise RuntimeError(f"dgs dit forward: {_native.status_string(self.lib, rc)} (status {rc})")
        if return_tokens:
            out["tokens"] = tokens
        if prof is not None:
            out["prof_count"] = int(count.value)
        return out, aligned


    def run_blocks(self, tokens, cvec, first, last, views=None):
        """DiT blocks [first, last) on tokens [B, L, W] (reference order: gaussian tokens first) under cvec [B, W] = t_embedder(t):
        DGSDenoiser.run_layers (denoiser.py:441-447).  Inference-mode utility; any token count (the workspace depends on (B, L)
        only; `views`, when given, is just checked against L)."""
        dev = self.device
        B, L, W = tokens.shape
        need = int(self.lib.dgs_dit_workspace_bytes_for_tokens(ctypes.byref(self.model), B, L))
        if need == 0:
            raise RuntimeError("dgs dit: invalid shape for workspace")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.zeros(need, dtype=torch.uint8, device=dev)
        elif self._ws_shape != ("tokens", B, L):       # different carving: padding rows must be finite again
            self._ws.zero_()
        self._ws_shape = ("tokens", B, L)
        ws = self._ws
        tin = tokens.to(dev, torch.float32).contiguous()
        c = cvec.to(dev, torch.float32).contiguous()
        out = torch.empty_like(tin)
        a = _native.DgsDitRunBlocksArgs()
        a.B, a.L, a.V, a.first, a.last = B, L, int(views or 0), int(first), int(last)
        a.tokens_in, a.cvec, a.tokens_out, a.workspace, a.workspace_bytes = _p(tin), _p(c), _p(out), _p(ws), ws.numel()
        rc = self.lib.dgs_dit_run_blocks(ctypes.byref(self.model), ctypes.byref(a), _stream(dev))
        if rc != 0:
            raise RuntimeError(f"dgs dit run_blocks: {_native.status_string(self.lib, rc)} (status {rc})")
        return out

    # ------------------------------------------------------------------------------------------------------------
    # training: forward that saves activations + backward (include/dgs_dit.h "training")
    # ------------------------------------------------------------------------------------------------------------
    def _train_state(self):
        """Transposed bf16 weight copies (K-contiguous operands of the input-gradient GEMMs) and the flat fp32 gradient
        buffer, laid out in backward-completion order (heads, block L-1 .. 0, embeddings) for bucketed all-reduce."""
        if self._train is not None:
            return self._train
        from .parallel import FlatGrads
        from ._native import DgsDitGrads, DgsDitLayerGrads, DgsDitLayerWeightsT, DgsDitModelT
        k, W, L = self._keep, self.width, self.layers
        tkeep = {}
        layersT = (DgsDitLayerWeightsT * L)()
        for i in range(L):
            for short in ("qkv", "proj", "fc1", "fc2"):
                tkeep[f"{i}.{short}"] = k[f"{i}.{short}_w"].t().contiguous()
                setattr(layersT[i], short + "_wT", tkeep[f"{i}.{short}"].data_ptr())
        tkeep["dec"] = k["dec_w"].t().contiguous()
        mt = DgsDitModelT()
        mt.layer = ctypes.cast(layersT, ctypes.POINTER(DgsDitLayerWeightsT))
        mt.dec_wT = tkeep["dec"].data_ptr()
        # flat gradient buffer: one slot per engine tensor (ada_w / ada_b are the stacked adaLN tensors)
        # (the adaLN Linear of a block -- 6W x W, a third of all parameters -- sits with its block: its gradient is final when the
        # block's backward is, not at the end)
        shapes = [("dec_w", k["dec_w"].shape), ("dec_ln_w", (W,)), ("up_w", k["up_w"].shape), ("up_ln_w", (W,)),
                  ("head_ada_w", (4 * W, W)), ("head_ada_b", (4 * W,))]
        for i in reversed(range(L)):
            for short in ("fc2", "fc1", "proj", "qkv"):
                shapes += [(f"{i}.{short}_w", k[f"{i}.{short}_w"].shape), (f"{i}.{short}_b", k[f"{i}.{short}_b"].shape)]
            shapes += [(f"{i}.ada_w", (6 * W, W)), (f"{i}.ada_b", (6 * W,))]
        shapes += [("in_ln_w", (W,)), ("pos_emb", k["pos_emb"].shape), ("tok_w", k["tok_w"].shape),
                   ("t_w1", k["t_w1"].shape), ("t_b1", (W,)), ("t_w0", k["t_w0"].shape), ("t_b0", (W,))]
        fg = FlatGrads(shapes, self.device)
        lgr = (DgsDitLayerGrads * L)()
        for i in range(L):
            for short in ("qkv", "proj", "fc1", "fc2", "ada"):
                setattr(lgr[i], short + "_w", fg.view(f"{i}.{short}_w").data_ptr())
                setattr(lgr[i], short + "_b", fg.view(f"{i}.{short}_b").data_ptr())
        gr = DgsDitGrads()
        for name in ("t_w0", "t_b0", "t_w1", "t_b1", "tok_w", "pos_emb", "in_ln_w", "head_ada_w", "head_ada_b", "up_ln_w", "up_w", "dec_ln_w", "dec_w"):
            setattr(gr, name, fg.view(name).data_ptr())
        gr.layer = ctypes.cast(lgr, ctypes.POINTER(DgsDitLayerGrads))
        # activation arenas: one per training forward whose backward has not run yet (`forward_train` takes a free one of the
        # right shape or allocates one); `saved` / `shape` / `recompute` / `ray_d` mirror the most recently used arena
        self._train = dict(tkeep=tkeep, layersT=layersT, mt=mt, fg=fg, lgr=lgr, gr=gr, saved=None, bws=None, shape=None, arenas=[])
        return self._train

    def grad_views(self):
        """state-dict key -> fp32 gradient view inside the flat buffer (per-block adaLN keys are slices of the stack)."""
        tr, W, L = self._train_state(), self.width, self.layers
        fg, out = tr["fg"], {}
        out["t_embedder.mlp.0.weight"], out["t_embedder.mlp.0.bias"] = fg.view("t_w0"), fg.view("t_b0")
        out["t_embedder.mlp.2.weight"], out["t_embedder.mlp.2.bias"] = fg.view("t_w1"), fg.view("t_b1")
        out["image_tokenizer.1.weight"] = fg.view("tok_w")
        out["gaussians_pos_embedding"] = fg.view("pos_emb")
        out["transformer_input_layernorm.weight"] = fg.view("in_ln_w")
        for i in range(L):
            p = f"transformer.{i}."
            for short, key in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2"), ("ada", "adaLN_modulation.1")):
                out[p + key + ".weight"], out[p + key + ".bias"] = fg.view(f"{i}.{short}_w"), fg.view(f"{i}.{short}_b")
        aw, ab = fg.view("head_ada_w"), fg.view("head_ada_b")
        for j, head in enumerate(("upsampler", "image_token_decoder")):
            out[head + ".adaLN_modulation.1.weight"] = aw[2 * W * j:2 * W * (j + 1)]
            out[head + ".adaLN_modulation.1.bias"] = ab[2 * W * j:2 * W * (j + 1)]
        out["upsampler.layernorm.weight"], out["upsampler.linear.weight"] = fg.view("up_ln_w"), fg.view("up_w")
        out["image_token_decoder.layernorm.weight"], out["image_token_decoder.linear.weight"] = fg.view("dec_ln_w"), fg.view("dec_w")
        return out

    def saved_bytes(self, B, V, H, W, recompute=False):
        """Bytes of the activation arena of one training forward (save-all, or the reference's per-block recompute mode)."""
        return int(self.lib.dgs_dit_saved_bytes(ctypes.byref(self.model), B, V, H, W, int(bool(recompute))))

    def _arena(self, B, V, H, W, recompute, keep_busy):
        """A free activation arena for this shape (allocated on first need).  keep_busy=False (the engine-level API: forward_train
        then backward, one pass at a time) reuses THE arena of the shape; True (autograd: several forwards may be pending) marks it
        busy until `release_arena`."""
        tr = self._train_state()
        for ar in tr["arenas"]:
            if ar["shape"] == (B, V, H, W) and ar["recompute"] == recompute and not ar["busy"]:
                break
        else:
            if not keep_busy:                # shape changed: drop the idle arenas of other shapes before allocating
                tr["arenas"][:] = [x for x in tr["arenas"] if x["busy"]]
                tr["saved"] = tr["bws"] = None
            m = ctypes.byref(self.model)
            ar = dict(shape=(B, V, H, W), recompute=recompute, busy=False, ray_d=None,
                      saved=torch.zeros(self.saved_bytes(B, V, H, W, recompute), dtype=torch.uint8, device=self.device),
                      bws=torch.zeros(int(self.lib.dgs_dit_backward_workspace_bytes(m, B, V, H, W)), dtype=torch.uint8, device=self.device))
            tr["arenas"].append(ar)
        ar["busy"] = bool(keep_busy)
        tr["saved"], tr["bws"], tr["shape"], tr["recompute"], tr["current"] = ar["saved"], ar["bws"], ar["shape"], ar["recompute"], ar
        return ar

    def release_arena(self, ar, drop_extra=True):
        """The backward of the forward that took `ar` has run (or its graph was dropped).  Idle duplicates of a shape -- arenas that
        only existed because several forwards were pending at once -- are freed."""
        ar["busy"] = False
        if drop_extra:
            tr, seen = self._train, set()
            keep = []
            for x in tr["arenas"]:
                key = (x["shape"], x["recompute"])
                if x["busy"] or key not in seen:
                    keep.append(x)
                if not x["busy"]:
                    seen.add(key)
            tr["arenas"][:] = keep

    @property
    def pending_backward(self):
        return any(ar["busy"] for ar in (self._train or {}).get("arenas", []))

    def forward_train(self, images, ray_o, ray_d, t, recompute=False, keep_busy=False):
        """image_to_gaussians that keeps what `backward` needs.  recompute=False: every activation is saved, nothing is
        recomputed; True: only block inputs are kept and `backward` re-runs each block (torch.utils.checkpoint's role,
        denoiser.py:348-354).  Returns (dict, aligned_xyz); the arena the pass used is `self._train["current"]`."""
        if self.sh_degree != 0:
            raise NotImplementedError("training with gaussians_sh_degree > 0: the training calls (dgs_dit_forward_train / dgs_dit_backward) take degree 0 "
                                      "-- every shipped config; the inference forward takes 0 .. 3")
        dev = self.device
        B, V, _, H, W = images.shape
        recompute = bool(recompute)
        ar = self._arena(B, V, H, W, recompute, keep_busy)
        img = images[:, :, :3].to(dev, torch.float32).contiguous()
        ro, rd = ray_o.to(dev, torch.float32).contiguous(), ray_d.to(dev, torch.float32).contiguous()
        tt = t.to(dev, torch.int64).contiguous()
        P = self.ng + V * H * W
        f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
        out = dict(xyz=f(B, P, 3), features=f(B, P, 1, 3), scaling=f(B, P, 3), rotation=f(B, P, 4), opacity=f(B, P, 1))
        aligned = f(B, V, 3, H, W)
        a = DgsDitForwardArgs()
        a.B, a.V, a.H, a.W = B, V, H, W
        a.images, a.ray_o, a.ray_d, a.t = _p(img), _p(ro), _p(rd), _p(tt)
        a.xyz, a.features, a.scaling, a.rotation, a.opacity = (_p(out[k]) for k in ("xyz", "features", "scaling", "rotation", "opacity"))
        a.aligned_xyz = _p(aligned)
        a.train_recompute = int(recompute)
        rc = self.lib.dgs_dit_forward_train(ctypes.byref(self.model), ctypes.byref(a), _p(ar["saved"]), ar["saved"].numel(), _stream(dev))
        if rc != 0:
            ar["busy"] = False
            raise RuntimeError(f"dgs dit forward_train: {_native.status_string(self.lib, rc)} (status {rc})")
        ar["ray_d"] = rd
        return out, aligned

    def backward(self, dxyz, dfeatures, dscaling, drotation, dopacity, block_hook=None, arena=None):
        """Gradients of every parameter into the flat buffer (overwritten).  Must follow `forward_train` (arena: the one that
        forward used; default the most recent).
        block_hook(stage) is called on the host as soon as a group of gradients has been ENQUEUED on the current stream
        (stage = layers: heads, layers-1..0: that block incl. its adaLN Linear, -1: the rest) -- the place to start a bucket's all-reduce."""
        tr = self._train_state()
        ar = arena if arena is not None else tr["current"]
        B, V, H, W = ar["shape"]
        dev = self.device
        g = [x.to(dev, torch.float32).contiguous() for x in (dxyz, dfeatures, dscaling, drotation, dopacity)]
        a = _native.DgsDitBackwardArgs()
        a.B, a.V, a.H, a.W = B, V, H, W
        a.ray_d = _p(ar["ray_d"])
        a.saved, a.saved_bytes, a.workspace, a.workspace_bytes = _p(ar["saved"]), ar["saved"].numel(), _p(ar["bws"]), ar["bws"].numel()
        a.dxyz, a.dfeatures, a.dscaling, a.drotation, a.dopacity = (_p(x) for x in g)
        a.recompute = int(ar["recompute"])
        errors = []
        if block_hook is not None:
            def _cb(_user, stage):
                try:
                    block_hook(int(stage))
                except BaseException as e:       # never unwind through the C frame
                    errors.append(e)
            cb = _native.BLOCK_DONE_FN(_cb)
            a.block_done = ctypes.cast(cb, ctypes.c_void_p)
        rc = self.lib.dgs_dit_backward(ctypes.byref(self.model), ctypes.byref(tr["mt"]), ctypes.byref(tr["gr"]), ctypes.byref(a),
                                       _stream(dev))
        if rc != 0:
            raise RuntimeError(f"dgs dit backward: {_native.status_string(self.lib, rc)} (status {rc})")
        if errors:
            raise errors[0]
        return tr["fg"]
