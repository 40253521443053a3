"""Logic tests of the DiT HIP kernels on the CPU emulator (tests/hipemu): fragment maps, LDS swizzles, epilogues,
online softmax, masking.  The same sources run on gfx950 in tests/test_dit_gpu.py; this file needs no GPU."""
import math

import pytest
import torch
import torch.nn.functional as F

from dgs_amd import _native
from dgs_amd.dit import DitOps
from emu_util import emu_lib


@pytest.fixture(scope="module")
def ops():
    return DitOps(lib=emu_lib())


def _bf(t):
    return t.to(torch.bfloat16)


def test_gemm_epilogues(ops):
    g = torch.Generator().manual_seed(1)
    M, N, K = 256, 384, 128
    A = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.1)
    bias = torch.randn(N, generator=g)
    ref = A.float() @ W.float().t() + bias
    out = ops.gemm(A, W, bias, _native.EPI_F32)
    assert torch.allclose(out, ref, atol=1e-3, rtol=1e-4)
    out = ops.gemm(A, W, None, _native.EPI_F32)
    assert torch.allclose(out, ref - bias, atol=1e-3, rtol=1e-4)
    out = ops.gemm(A, W, bias, _native.EPI_BF16)
    assert torch.allclose(out.float(), ref, atol=2e-2, rtol=1e-2)
    out = ops.gemm(A, W, bias, _native.EPI_GELU_BF16)
    assert torch.allclose(out.float(), F.gelu(ref, approximate="tanh"), atol=2e-2, rtol=1e-2)
    # gated residual, two samples of 128 rows
    x0 = torch.randn(M, N, generator=g)
    gate = torch.randn(2, 3 * N, generator=g)[:, N:2 * N]          # strided view like the adaLN buffer
    x = x0.clone()
    ops.gemm(A, W, bias, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, rows_per_batch=128)
    want = x0 + gate.repeat_interleave(128, 0) * ref
    assert torch.allclose(x, want, atol=2e-3, rtol=1e-4)


def test_gemm_padding_blocks_are_skipped(ops):
    g = torch.Generator().manual_seed(7)
    rows, B, N, K = 256, 2, 128, 128
    A = _bf(torch.randn(B * rows, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.1)
    ref = A.float() @ W.float().t()
    for valid in (130, 33, 200, 256):
        out = torch.full((B * rows, N), 7.0)
        ops.gemm(A, W, None, _native.EPI_F32, out=out, rows_per_batch=rows, valid_rows=valid)
        for b in range(B):
            blk = out[b * rows:(b + 1) * rows]
            rblk = ref[b * rows:(b + 1) * rows]
            # every valid row is computed; a padding row is either computed (finite, correct) or left untouched
            assert torch.allclose(blk[:valid], rblk[:valid], atol=1e-3, rtol=1e-4)
            pad_ok = torch.isclose(blk[valid:], rblk[valid:], atol=1e-3, rtol=1e-4).all(1) | (blk[valid:] == 7.0).all(1)
            assert bool(pad_ok.all())
        if valid == 33:       # second 128-row tile of each sample holds one live block only: its other 96 rows are skipped
            assert bool((out[128 + 32:256] == 7.0).all())
    # narrow-tile variant (chosen when 128 x 128 tiles would under-fill the chip) gives the same numbers
    out = ops.gemm(A, W, None, _native.EPI_F32)
    assert torch.allclose(out, ref, atol=1e-3, rtol=1e-4)


def test_gemm_qkv_epilogue(ops):
    g = torch.Generator().manual_seed(2)
    lpad, B, Wd, K = 128, 2, 128, 128
    A = _bf(torch.randn(B * lpad, K, generator=g))
    W = _bf(torch.randn(3 * Wd, K, generator=g) * 0.1)
    bias = torch.randn(3 * Wd, generator=g)
    ref = A.float() @ W.float().t() + bias
    qk, vt = ops.gemm(A, W, bias, _native.EPI_QKV, rows_per_batch=lpad)
    assert torch.allclose(qk.float(), ref[:, :2 * Wd], atol=2e-2, rtol=1e-2)
    qk2, vt2 = ops.gemm(A, W, bias, _native.EPI_QKV, rows_per_batch=lpad, q_scale=0.18)     # pre-scaled queries: q features only
    assert torch.allclose(qk2.float()[:, :Wd], 0.18 * ref[:, :Wd], atol=4e-3, rtol=1e-2)
    assert torch.equal(qk2[:, Wd:], qk[:, Wd:]) and torch.equal(vt2, vt)
    v = ref[:, 2 * Wd:].reshape(B, lpad, Wd).transpose(1, 2)
    assert torch.allclose(vt.float(), v, atol=2e-2, rtol=1e-2)
    # 2 live rows in each sample: 128-wide tiles with a single live block take the unstaged path (one column block per wave)
    qk3, vt3 = ops.gemm(A, W, bias, _native.EPI_QKV, rows_per_batch=lpad, valid_rows=2)
    for b in range(B):
        assert torch.equal(qk3[b * lpad:b * lpad + 2], qk[b * lpad:b * lpad + 2]) and torch.equal(vt3[b, :, :2], vt[b, :, :2])


# 520 / 600: enough key tiles for the branch-free steady-state rounds; at 600 the last workgroup also has waves without queries
# 97 / 130 / 67 / 132: 1, 2, 3, 4 tail queries = 8, 4, 2, 2 record slices in the merge of the tail records; 520 / 600 / 20: one thread per item
@pytest.mark.parametrize("L,prescaled", [(128, False), (130, False), (130, True), (67, True), (258, False), (520, True), (600, True), (20, False), (97, False), (132, True)])
def test_attention(ops, L, prescaled):
    g = torch.Generator().manual_seed(3)
    B, heads = 2, 2
    lpad = (L + 127) // 128 * 128
    q = torch.randn(B, heads, lpad, 64, generator=g)
    k = torch.randn(B, heads, lpad, 64, generator=g)
    v = torch.randn(B, heads, lpad, 64, generator=g)
    q[0, 0, 3] *= 6.0   # a spiky row: exercises the running-max rescale across tiles
    k[0, 0, 70 % L] *= 6.0
    c = 0.125 * 1.4426950408889634
    qb, kb, vb = (_bf(q * c) if prescaled else _bf(q)), _bf(k), _bf(v)
    qk = torch.cat([qb.permute(0, 2, 1, 3).reshape(B * lpad, heads * 64), kb.permute(0, 2, 1, 3).reshape(B * lpad, heads * 64)], 1).contiguous()
    vt = vb.permute(0, 1, 3, 2).reshape(B, heads * 64, lpad).contiguous()
    lse2 = torch.zeros(B, heads, lpad)
    out = ops.attention(qk, vt, L, heads, lse2=lse2, q_prescaled=prescaled).float().reshape(B, lpad, heads, 64).permute(0, 2, 1, 3)
    s = (qb.float()[:, :, :L] @ kb.float()[:, :, :L].transpose(-1, -2)) * (0.6931471805599453 if prescaled else 0.125)
    ref = s.softmax(-1) @ vb.float()[:, :, :L]
    assert torch.allclose(out[:, :, :L], ref, atol=2e-2, rtol=2e-2)
    # log2-domain log-sum-exp of every valid query (main path and the L % 32 tail queries)
    assert torch.allclose(lse2[:, :, :L], torch.logsumexp(s, -1) * 1.4426950408889634, atol=5e-2, rtol=5e-3)


def test_layernorm_modulate(ops):
    g = torch.Generator().manual_seed(4)
    rows, Wd = 24, 1024
    x = torch.randn(rows, Wd, generator=g) * 3 + 1
    w = torch.randn(Wd, generator=g)
    mod = torch.randn(2, 6 * Wd, generator=g)
    shift, scale = mod[:, :Wd], mod[:, Wd:2 * Wd]
    out = ops.layernorm(x, None, shift, scale, rows_per_batch=12, eps=1e-6)
    ref = F.layer_norm(x, (Wd,), eps=1e-6) * (1 + scale.repeat_interleave(12, 0)) + shift.repeat_interleave(12, 0)
    assert torch.allclose(out.float(), ref, atol=3e-2, rtol=1e-2)
    out = ops.layernorm(x, w, None, None, eps=1e-5, out_f32=True)
    assert torch.allclose(out, F.layer_norm(x, (Wd,), w, None, 1e-5), atol=1e-5, rtol=1e-5)
    x256 = x[:, :256].contiguous()
    out = ops.layernorm(x256, None, None, None, eps=1e-6, out_f32=True)
    assert torch.allclose(out, F.layer_norm(x256, (256,), eps=1e-6), atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("M,K", [(1, 256), (3, 1024), (16, 512)])
def test_rowlinear(ops, M, K):
    g = torch.Generator().manual_seed(5)
    N = 37
    x = torch.randn(M, K, generator=g)
    W = _bf(torch.randn(N, K, generator=g) * 0.1)
    b = torch.randn(N, generator=g)
    out = ops.rowlinear(x, W, b, silu_input=True)
    assert torch.allclose(out, F.linear(F.silu(x), W.float(), b), atol=1e-4, rtol=1e-4)
    out = ops.rowlinear(x, W, None, silu_output=True)
    assert torch.allclose(out, F.silu(F.linear(x, W.float())), atol=1e-4, rtol=1e-4)


def test_gemm_training_epilogues(ops):
    g = torch.Generator().manual_seed(11)
    rows, B, N, K = 128, 2, 128, 64
    M = B * rows
    A = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.2)
    bias = torch.randn(N, generator=g)
    ref = A.float() @ W.float().t() + bias
    tr = lambda t: t.reshape(B, rows, N).transpose(1, 2)
    # bf16 + transposed copy
    vt = torch.zeros(B, N, rows, dtype=torch.bfloat16)
    out = ops.gemm(A, W, bias, _native.EPI_BF16, rows_per_batch=rows, vt=vt)
    assert torch.allclose(out.float(), ref, atol=3e-2, rtol=1e-2) and torch.equal(vt, tr(out))
    # GELU: pre-activation saved, transposed copy of the activation
    aux = torch.zeros(M, N, dtype=torch.bfloat16)
    out = ops.gemm(A, W, bias, _native.EPI_GELU_BF16, rows_per_batch=rows, vt=vt, aux=aux)
    assert torch.allclose(aux.float(), ref, atol=3e-2, rtol=1e-2) and torch.equal(vt, tr(out))
    # dGELU: out = acc * gelu'(u)
    u = _bf(torch.randn(M, N, generator=g) * 2)
    out = ops.gemm(A, W, None, _native.EPI_DGELU_BF16, rows_per_batch=rows, vt=vt, aux=u)
    uu = u.float().requires_grad_(True)
    F.gelu(uu, approximate="tanh").sum().backward()
    assert torch.allclose(out.float(), (ref - bias) * uu.grad, atol=3e-2, rtol=2e-2) and torch.equal(vt, tr(out))
    # out-of-place gated residual + saved pre-gate value
    x0 = torch.randn(M, N, generator=g)
    gate = torch.randn(B, N, generator=g)
    x1 = torch.zeros(M, N)
    y = torch.zeros(M, N, dtype=torch.bfloat16)
    ops.gemm(A, W, bias, _native.EPI_GATE_RESIDUAL, out=x1, gate=gate, rows_per_batch=rows, resid=x0, aux=y)
    assert torch.allclose(x1, x0 + gate.repeat_interleave(rows, 0) * ref, atol=2e-3, rtol=1e-4)
    assert torch.allclose(y.float(), ref, atol=3e-2, rtol=1e-2)


def test_gemm_weight_gradient_batched_reduction(ops):
    """dW[N,K] = sum over samples and tokens of dY^T X with both operands stored transposed [batch, features, tokens]."""
    g = torch.Generator().manual_seed(12)
    B, T, N, K = 3, 128, 128, 256
    dyT = _bf(torch.randn(B, N, T, generator=g))
    xT = _bf(torch.randn(B, K, T, generator=g))
    ref = torch.einsum("bnt,bkt->nk", dyT.float(), xT.float())
    out = ops.gemm(dyT, xT, None, _native.EPI_F32, shape=(N, K, B * T), k_per_batch=T, a_batch_stride=N * T, w_batch_stride=K * T,
                   lda=T, ldw=T)
    assert torch.allclose(out, ref, atol=2e-3, rtol=1e-4)


def _attention_case(L, B, heads, seed):
    """q, k, v (bf16-rounded) in the training layout + torch autograd reference."""
    g = torch.Generator().manual_seed(seed)
    lpad = (L + 127) // 128 * 128
    W = heads * 64
    qkv = _bf(torch.randn(B, lpad, 3 * W, generator=g))          # padding rows hold finite garbage, like in the model
    dO = torch.zeros(B, lpad, W)
    dO[:, :L] = torch.randn(B, L, W, generator=g)
    dO = _bf(dO)
    x = qkv.float()[:, :L].reshape(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4).contiguous().requires_grad_(True)
    q, k, v = x[0], x[1], x[2]
    o = ((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v                       # [B, heads, L, 64]
    o.backward(dO.float()[:, :L].reshape(B, L, heads, 64).permute(0, 2, 1, 3))
    dref = x.grad.permute(1, 3, 0, 2, 4).reshape(B, L, 3 * W)                    # [B, L, (q|k|v) W]
    return lpad, W, qkv, dO, o.detach().permute(0, 2, 1, 3).reshape(B, L, W), dref


@pytest.mark.parametrize("L,B,heads", [(128, 1, 2), (130, 2, 2), (300, 1, 2), (258, 2, 2), (261, 1, 2), (258, 1, 8)])
def test_attention_backward(ops, L, B, heads):
    """258, 261: tail-token workgroups; 8 heads: the XCD-aware numbering of the workgroups (B x heads a multiple of 8)"""
    lpad, W, qkv, dO, oref, dref = _attention_case(L, B, heads, seed=L)
    qkv2 = qkv.reshape(B * lpad, 3 * W).contiguous()
    qkvT = qkv.transpose(1, 2).contiguous()
    lse2 = torch.zeros(B, heads, lpad)
    o = ops.attention(qkv2, qkvT, L, heads, qkv_layout=True, lse2=lse2)
    assert torch.allclose(o.float().reshape(B, lpad, W)[:, :L], oref, atol=2e-2, rtol=2e-2)
    dOT = dO.transpose(1, 2).contiguous()
    dqkv = ops.attention_backward(qkv2, qkvT, o, dO.reshape(B * lpad, W).contiguous(), dOT, lse2, L, heads)
    got = dqkv.float().reshape(B, lpad, 3 * W)
    for name, sl in (("dq", slice(0, W)), ("dk", slice(W, 2 * W)), ("dv", slice(2 * W, 3 * W))):
        a, r = got[:, :L, sl], dref[:, :, sl]
        err = float((a - r).norm() / r.norm())
        assert err < 2e-2, (name, err)
        if L > 256:                     # the rows of the tokens behind the last full 256-block on their own
            err = float((a[:, 256:] - r[:, 256:]).norm() / r[:, 256:].norm())
            assert err < 2e-2, (name, "tail rows", err)
    if L < lpad:
        assert float(got[:, L:].abs().max()) == 0.0   # padding rows receive exactly zero gradient
    _check_attention_backward_byproducts(ops, qkv2, qkvT, o, dO.reshape(B * lpad, W).contiguous(), dOT, lse2, L, heads, dqkv)


def _check_attention_backward_byproducts(ops, qkv2, qkvT, o, dO2, dOT, lse2, L, heads, dqkv):
    """DgsDitAttentionBackwardArgs.dqkvT / bias_part: the same launch again with the by-products on -- dqkv bit-identical, the
    token-contiguous copy equal to its transpose bit for bit on the valid tokens (zero on the padding), the per-workgroup partial rows
    all written and adding up to the column sums of the gradient (fp32 sums of the values BEFORE their bf16 rounding: bf16 tolerance)."""
    B, _, lpad = qkvT.shape
    W3 = qkv2.shape[1]
    again, dT, part = ops.attention_backward(qkv2, qkvT, o, dO2, dOT, lse2, L, heads, byproducts=True)
    assert torch.equal(again, dqkv)
    want_T = dqkv.reshape(B, lpad, W3).transpose(1, 2)
    assert torch.equal(dT[:, :, :L], want_T[:, :, :L]) and float(dT[:, :, L:].float().abs().max() if L < lpad else 0.0) == 0.0
    assert torch.isfinite(part).all() and part.shape[0] == B * int(ops.lib.dgs_dit_attention_backward_slots(L))
    colsum = dqkv.float().reshape(B * lpad, W3).sum(0)
    scale = float(dqkv.float().abs().max()) * (B * L) ** 0.5
    assert float((part.sum(0) - colsum).abs().max()) <= 4e-3 * scale + 1e-6


def _call(ops, fn, struct, **kw):
    import ctypes
    a = struct()
    keep = []
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            keep.append(v)
            setattr(a, k, v.data_ptr())
        elif v is not None:
            setattr(a, k, v)
    rc = getattr(ops.lib, fn)(ctypes.byref(a), None)
    assert rc == 0, rc


@pytest.mark.parametrize("cus,rows,rpb", [(0, 64, 32), (3, 66, 66), (8, 96, 24)])
def test_layernorm_backward(ops, monkeypatch, cus, rows, rpb):
    """rows per workgroup: 32 when the CU count is unknown, else the smallest divisor of the sample's rows (>= 16) that puts all rows
    into one round of workgroups (DGS_EMU_CUS plays the device query on the emulator): 2 x 66 rows on 3 CUs -> 66, 2 x 96 on 8 -> 24."""
    monkeypatch.setenv("DGS_EMU_CUS", str(cus))
    g = torch.Generator().manual_seed(21)
    B, Wd = 2, 1024
    x = (torch.randn(B * rows, Wd, generator=g) * 2 + 0.5).requires_grad_(True)
    w = (1 + 0.2 * torch.randn(Wd, generator=g)).requires_grad_(True)
    mod = torch.randn(B, 3 * Wd, generator=g)
    shift = mod[:, :Wd].clone().requires_grad_(True)
    scale = mod[:, Wd:2 * Wd].clone().requires_grad_(True)
    dh = _bf(torch.randn(B * rows, Wd, generator=g))
    dx_in = torch.randn(B * rows, Wd, generator=g)
    h = F.layer_norm(x, (Wd,), w, None, 1e-6) * (1 + scale.repeat_interleave(rows, 0)) + shift.repeat_interleave(rows, 0)
    h.backward(dh.float())
    dmod = torch.zeros(B, 3 * Wd)
    dmod[:, :2 * Wd] = 7.0                         # the column sums are WRITTEN (per-workgroup partial rows + one ordered reduce)
    dw = torch.full((Wd,), 7.0)
    dx = torch.zeros(B * rows, Wd)
    nb = ops.lib.dgs_dit_layernorm_backward_scratch_bytes(B * rows, Wd, rows)
    assert nb == B * (rows // rpb) * 3 * Wd * 4
    scratch = torch.full((nb // 4,), float("nan"))
    _call(ops, "dgs_dit_layernorm_backward", _native.DgsDitLayerNormBackwardArgs, rows=B * rows, width=Wd, x=x.detach(), dh=dh,
          dh_f32=0, weight=w.detach(), scale=mod[:, Wd:], mod_stride=3 * Wd, rows_per_batch=rows, eps=1e-6, dx_in=dx_in, dx_out=dx,
          dshift=dmod, dscale=dmod[:, Wd:], dweight=dw, scratch=scratch, scratch_bytes=nb)
    assert torch.allclose(dx, x.grad + dx_in, atol=2e-4, rtol=1e-3)
    assert torch.allclose(dmod[:, :Wd], shift.grad, atol=1e-3, rtol=1e-3)
    assert torch.allclose(dmod[:, Wd:2 * Wd], scale.grad, atol=2e-3, rtol=1e-3)
    assert torch.allclose(dw, w.grad, atol=2e-3, rtol=1e-3)
    assert float(dmod[:, 2 * Wd:].abs().max()) == 0.0


def test_rowlinear_backward_and_gate_mul(ops):
    g = torch.Generator().manual_seed(22)
    M, N, K = 3, 70, 256
    x = torch.randn(M, K, generator=g).requires_grad_(True)
    Wb = _bf(torch.randn(N, K, generator=g) * 0.1)
    Wf = Wb.float().requires_grad_(True)
    b = torch.zeros(N, requires_grad=True)
    dy = torch.randn(M, N, generator=g)
    F.linear(F.silu(x), Wf, b).backward(dy)
    dW, db, dx = torch.zeros(N, K), torch.zeros(N), torch.full((M, K), 7.0)
    nb = ops.lib.dgs_dit_rowlinear_backward_scratch_bytes(M, N, K)
    scratch = torch.full((nb // 4,), float("nan"))
    _call(ops, "dgs_dit_rowlinear_backward", _native.DgsDitRowLinearBackwardArgs, M=M, N=N, K=K, x=x.detach(), silu_input=1, W=Wb,
          dy=dy, dW=dW, db=db, dx=dx, scratch=scratch, scratch_bytes=nb)
    assert torch.allclose(dW, Wf.grad, atol=1e-4, rtol=1e-4) and torch.allclose(db, b.grad, atol=1e-5)
    assert torch.allclose(dx, x.grad, atol=1e-4, rtol=1e-4)
    # gate_mul
    B, rows, Wd = 2, 64, 128
    dxr = torch.randn(B * rows, Wd, generator=g)
    y = _bf(torch.randn(B * rows, Wd, generator=g))
    gate = torch.randn(B, 2 * Wd, generator=g)
    dyo = torch.zeros(B * rows, Wd, dtype=torch.bfloat16)
    dyT = torch.zeros(B, Wd, rows, dtype=torch.bfloat16)
    dgate = torch.zeros(B, 2 * Wd)
    dgate[:, Wd:] = 7.0
    dbias = torch.full((Wd,), 7.0)
    nb = ops.lib.dgs_dit_gate_mul_scratch_bytes(B, rows, Wd)
    scratch = torch.full((nb // 4,), float("nan"))
    _call(ops, "dgs_dit_gate_mul", _native.DgsDitGateMulArgs, B=B, rows=rows, width=Wd, dx=dxr, y=y, gate=gate[:, Wd:], gate_stride=2 * Wd,
          dy=dyo, dyT=dyT, dgate=dgate[:, Wd:], dbias=dbias, scratch=scratch, scratch_bytes=nb)
    assert torch.allclose(dbias, dyo.float().sum(0), atol=1e-3, rtol=1e-4)
    assert float(dgate[:, :Wd].abs().max()) == 0.0
    want = gate[:, Wd:].repeat_interleave(rows, 0) * dxr
    assert torch.allclose(dyo.float(), want, atol=3e-2, rtol=1e-2)
    assert torch.equal(dyT, dyo.reshape(B, rows, Wd).transpose(1, 2))
    assert torch.allclose(dgate[:, Wd:], (dxr * y.float()).reshape(B, rows, Wd).sum(1), atol=1e-3, rtol=1e-4)


def test_gemm_split_k_weight_gradient(ops, monkeypatch):
    """Split-K form of the weight-gradient GEMM: per-sample operands, uneven K ranges (9 units of 128 over 2 splits), partial planes
    summed into a strided output; the scratch size query says when the path applies."""
    monkeypatch.setenv("DGS_SPLITK_MIN_ITEMS", "1")
    g = torch.Generator().manual_seed(97)
    B, N, K, T = 2, 256, 512, 1152
    dyT = _bf(torch.randn(B, N, T, generator=g) * 0.3)
    xT = _bf(torch.randn(B, K, T, generator=g) * 0.3)
    ref = torch.einsum("bnt,bkt->nk", dyT.float(), xT.float())
    assert ops.lib.dgs_dit_gemm_splitk_bytes(N, K, B * T, T) == 4 * N * K * 4          # 2 samples x 2 K ranges
    assert ops.lib.dgs_dit_gemm_splitk_bytes(N, K + 64, B * T, T) == 0
    assert ops.lib.dgs_dit_gemm_splitk_bytes(N, K, B * T, B * T) == 4 * N * K * 4      # one sample of 18 units: 4 ranges
    out = torch.full((N, K), 7.0)
    ops.gemm(dyT, xT, None, _native.EPI_F32, out=out, shape=(N, K, B * T), k_per_batch=T, a_batch_stride=N * T, w_batch_stride=K * T,
             lda=T, ldw=T, splitk=True)
    assert torch.allclose(out, ref, atol=2e-3, rtol=1e-4)
    plain = ops.gemm(dyT, xT, None, _native.EPI_F32, shape=(N, K, B * T), k_per_batch=T, a_batch_stride=N * T, w_batch_stride=K * T,
                     lda=T, ldw=T)
    assert torch.allclose(out, plain, atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("N", [256, 128, -256])
def test_gemm_sliced(ops, N):
    """Sliced-schedule kernel (256 x 256|128 x 32 tiles, 4-stage LDS-DMA ring): several trips round the ring, the padding-row
    path, and the QKV / gate-residual epilogues."""
    sl = _native.GEMM_SLICED
    if N < 0:                                       # the 4-wave variant (128 x 128 per wave) of the 256 x 256 tile
        sl, N = _native.GEMM_QUAD, -N
    g = torch.Generator().manual_seed(41 + N)
    M, K = 512, 1024                                # 32 K slabs = 8 trips round the ring; K % 1024 == 0: single-block tiles go direct
    if N == 256:
        N = 768                                     # 2 x 3 tiles of 256 x 256: needs >= 160 tiles for AUTO, forced via algo here
    A = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.2)
    bias = torch.randn(N, generator=g)
    ref = A.float() @ W.float().t() + bias
    out = ops.gemm(A, W, bias, _native.EPI_F32, algo=sl)
    assert torch.allclose(out, ref, atol=3e-3, rtol=1e-4)
    out = torch.full((M, N), 7.0)
    ops.gemm(A, W, bias, _native.EPI_F32, out=out, rows_per_batch=256, valid_rows=130, algo=sl)
    for b in range(2):                              # per sample: rows [0, 130) live; the tile is computed and stored completely
        assert torch.allclose(out[b * 256:b * 256 + 130], ref[b * 256:b * 256 + 130], atol=3e-3, rtol=1e-4)
    out = torch.full((M, N), 7.0)
    ops.gemm(A, W, bias, _native.EPI_F32, out=out, rows_per_batch=256, valid_rows=2, algo=sl)     # one live block: direct path
    for b in range(2):
        assert torch.allclose(out[b * 256:b * 256 + 2], ref[b * 256:b * 256 + 2], atol=3e-3, rtol=1e-4)
        assert bool((out[b * 256 + 32:(b + 1) * 256] == 7.0).all())
    x0 = torch.randn(M, N, generator=g)
    gate = torch.randn(2, N, generator=g)
    x = x0.clone()
    ops.gemm(A, W, bias, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, rows_per_batch=256, algo=sl)
    assert torch.allclose(x, x0 + gate.repeat_interleave(256, 0) * ref, atol=5e-3, rtol=1e-4)
    if N % 384 == 0:
        qk, vt = ops.gemm(A, W, bias, _native.EPI_QKV, rows_per_batch=256, algo=sl, q_scale=0.5)
        Wd = N // 3
        assert torch.allclose(qk.float()[:, :Wd], 0.5 * ref[:, :Wd], atol=3e-2, rtol=1e-2)
        assert torch.allclose(qk.float()[:, Wd:], ref[:, Wd:2 * Wd], atol=3e-2, rtol=1e-2)
        assert torch.allclose(vt.float(), ref[:, 2 * Wd:].reshape(2, 256, Wd).transpose(1, 2), atol=3e-2, rtol=1e-2)



@pytest.mark.parametrize("N,K", [(256, 1024), (1536, 1024), (512, 2048)])
def test_gemm_sliced_learned_token_rows(ops, N, K):
    """A full tile row plus two live rows behind it (L = 256 + 2): the two rows are GEMV side jobs -- on workgroups of their own
    when the launch leaves CUs idle (1 or 2 tiles on the emulator's 6 CUs), inside the first tile workgroups otherwise (6 tiles)."""
    g = torch.Generator().manual_seed(N + K)
    M, L = 1024, 258
    A = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.1)
    bias = torch.randn(N, generator=g)
    ref = A.float() @ W.float().t() + bias
    for algo in (_native.GEMM_SLICED, _native.GEMM_QUAD):
        out = torch.full((M, N), 7.0)
        ops.gemm(A, W, bias, _native.EPI_F32, out=out, rows_per_batch=512, valid_rows=L, algo=algo)
        for b in range(2):
            assert torch.allclose(out[b * 512:b * 512 + L], ref[b * 512:b * 512 + L], atol=4e-3, rtol=1e-4), (algo, b)
    x0 = torch.randn(M, N, generator=g)
    gate = torch.randn(2, N, generator=g)
    x = x0.clone()
    ops.gemm(A, W, bias, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, rows_per_batch=512, valid_rows=L, algo=_native.GEMM_SLICED)
    want = x0 + gate.repeat_interleave(512, 0) * ref
    for b in range(2):
        assert torch.allclose(x[b * 512:b * 512 + L], want[b * 512:b * 512 + L], atol=6e-3, rtol=1e-4)


def _check_layernorm_gemm_pair(ops, dev, width, N, qkv, valids=(258, 257), B=2, rpb=512, algo=_native.GEMM_SLICED):
    """dgs_dit_layernorm_gemm against dgs_dit_layernorm + dgs_dit_gemm: the learned tokens' output rows come from the first workgroups
    of the LayerNorm launch (layernorm_rows_gemv_kernel) instead of the GEMM's side jobs -- every output bit for bit the same."""
    g = torch.Generator().manual_seed(width + N)
    epi = _native.EPI_QKV if qkv else _native.EPI_GELU_BF16
    for valid in valids:
        x = torch.randn(B * rpb, width, generator=g).to(dev)
        mod = (torch.randn(B, 2 * width, generator=g) * 0.3).to(dev)
        shift, scale = mod[:, :width], mod[:, width:]
        W = _bf(torch.randn(N, width, generator=g) * 0.05).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        h_ref = ops.layernorm(x, shift=shift, scale=scale, rows_per_batch=rpb)
        ref = ops.gemm(h_ref, W, bias, epi, rows_per_batch=rpb, valid_rows=valid, algo=algo, q_scale=0.7)
        got = ops.layernorm_gemm(x, shift, scale, W, bias, epi, rpb, valid, algo=algo, q_scale=0.7)
        assert ops.last_pair_shared_rows, (width, N, valid)
        live = ((torch.arange(B * rpb) % rpb) < valid).to(dev)
        assert torch.equal(got[0].view(torch.int16), h_ref.view(torch.int16))
        out_ref = ref[0] if qkv else ref
        assert torch.equal(got[1][live].view(torch.int16), out_ref[live].view(torch.int16)), (width, N, valid)
        assert bool((got[1][~live] == 0).all())                      # padding rows: never written
        if qkv:
            assert torch.equal(got[2].view(torch.int16), ref[1].view(torch.int16))
        assert float(got[1][live].float().abs().max()) > 0.1


@pytest.mark.parametrize("width,N,qkv", [(1024, 1536, True), (512, 1536, True), (1024, 512, False), (2048, 256, False)])
def test_layernorm_gemm_pair_moves_the_learned_token_rows(ops, width, N, qkv):
    _check_layernorm_gemm_pair(ops, "cpu", width, N, qkv)


def test_gemm_sliced_2d_xcd_map(ops, monkeypatch):
    """One tile per CU and a tile grid that splits 4 x 2 over the XCDs: the workgroup id -> tile map is blocks of the grid, not runs of
    rows (same tiles, every one exactly once: the output is prefilled, a tile done twice or never shows).  16 'CUs' on the emulator."""
    monkeypatch.setenv("DGS_EMU_CUS", "16")
    g = torch.Generator().manual_seed(42)
    for M, N, K in ((1024, 512, 256), (2048, 256, 128), (1024, 256, 128)):
        A = _bf(torch.randn(M, K, generator=g))
        W = _bf(torch.randn(N, K, generator=g) * 0.1)
        bias = torch.randn(N, generator=g)
        out = torch.full((M, N), 7.0)
        ops.gemm(A, W, bias, _native.EPI_F32, out=out, rows_per_batch=M, algo=_native.GEMM_SLICED)
        assert ops.last_gemm_sliced_tile in (128, 256)               # (few tiles: the 128-wide form of the ring kernel; 16 or 8 tiles here)
        assert torch.allclose(out, A.float() @ W.float().t() + bias, atol=4e-3, rtol=1e-4), (M, N, K)


def test_gemm_sliced_192_wide_qkv_tiles(ops, monkeypatch):
    """QKV on 256 x 192 tiles (chosen when 256-wide tiles leave CUs idle and 192-wide ones fit: N = 768 on the emulator's 6 CUs is 3 -> 4
    tiles; the model's N = 3072 on 256 CUs is 192 -> 256): three column blocks per wave, the doubled W piece of waves 4..7, the strip
    that straddles the K | V boundary (columns 480..543 around 512) stored as two single blocks -- against the 256 / 128-wide tiles
    (DGS_GEMM_NO_BN192 is read once per process, so the comparison is the other kernel family) and fp32 math; then the pair with the
    learned tokens' rows in the LayerNorm launch on top of it."""
    g = torch.Generator().manual_seed(192)
    N, K, Wd = 768, 1024, 256
    for M, rpb, valid in ((512, 512, 258), (256, 256, 256), (1024, 512, 257)):
        if (M // rpb) * 4 > 6:
            monkeypatch.setenv("DGS_EMU_CUS", "8")
        A = _bf(torch.randn(M, K, generator=g))
        W = _bf(torch.randn(N, K, generator=g) * 0.1)
        bias = torch.randn(N, generator=g)
        ref = A.float() @ W.float().t() + bias
        qk, vt = ops.gemm(A, W, bias, _native.EPI_QKV, rows_per_batch=rpb, valid_rows=valid, algo=_native.GEMM_SLICED, q_scale=0.5)
        assert ops.last_gemm_sliced_tile == 192
        qk2, vt2 = ops.gemm(A, W, bias, _native.EPI_QKV, rows_per_batch=rpb, valid_rows=valid, algo=_native.GEMM_SIMPLE128, q_scale=0.5)
        live = (torch.arange(M) % rpb) < valid
        assert torch.allclose(qk.float()[live][:, :Wd], 0.5 * ref[live][:, :Wd], atol=3e-2, rtol=1e-2)
        assert torch.allclose(qk.float()[live][:, Wd:], ref[live][:, Wd:2 * Wd], atol=3e-2, rtol=1e-2)
        vref = ref[:, 2 * Wd:].reshape(M // rpb, rpb, Wd).transpose(1, 2)
        assert torch.allclose(vt.float()[:, :, :valid], vref[:, :, :valid], atol=3e-2, rtol=1e-2)
        # same k order per output (32-wide slabs, fp32 accumulate in the MFMA): the two kernel families agree to the bf16 rounding of the output
        assert torch.allclose(qk.float()[live], qk2.float()[live], atol=2e-2, rtol=1e-2) and torch.allclose(vt.float()[:, :, :valid], vt2.float()[:, :, :valid], atol=2e-2, rtol=1e-2)
    _check_layernorm_gemm_pair(ops, "cpu", 1024, 768, True, B=1)


def test_layernorm_gemm_pair_falls_back_to_two_launches(ops):
    """shapes the fused form does not take (a LayerNorm weight, full tiles only, a 128-wide GEMM): the pair is the two plain launches"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(512, 1024, generator=g)
    mod = torch.randn(1, 2048, generator=g) * 0.3
    W = _bf(torch.randn(256, 1024, generator=g) * 0.05)
    bias = torch.randn(256, generator=g)
    for valid, algo in ((512, _native.GEMM_SLICED), (258, _native.GEMM_SIMPLE128), (300, _native.GEMM_SLICED)):
        h_ref = ops.layernorm(x, shift=mod[:, :1024], scale=mod[:, 1024:], rows_per_batch=512)
        ref = ops.gemm(h_ref, W, bias, _native.EPI_GELU_BF16, rows_per_batch=512, valid_rows=valid, algo=algo)
        h, out = ops.layernorm_gemm(x, mod[:, :1024], mod[:, 1024:], W, bias, _native.EPI_GELU_BF16, 512, valid, algo=algo)
        assert not ops.last_pair_shared_rows
        assert torch.equal(h.view(torch.int16), h_ref.view(torch.int16)) and torch.equal(out[:valid].view(torch.int16), ref[:valid].view(torch.int16))


def test_gemm_128_wide_gemv_tail_rows(ops):
    """The 128-wide kernel with one or two live rows behind a sample's last full tile (the DiT's learned tokens): those rows are
    GEMV items of the first workgroups, not a tile row.  Every epilogue the N = 1024 GEMMs use, two samples, K = 512 and a K
    beyond one 2048-element pass; rows past `valid` in the touched 32-row block are padding (any finite value), the rest untouched."""
    g = torch.Generator().manual_seed(91)
    rows, B, N = 256, 2, 128
    for K, valid in ((512, 130), (1024, 130), (3072, 129), (4096, 130)):
        A = _bf(torch.randn(B * rows, K, generator=g) * 0.5)
        W = _bf(torch.randn(N, K, generator=g) * 0.05)
        bias = torch.randn(N, generator=g)
        ref = A.float() @ W.float().t() + bias
        live = (torch.arange(B * rows) % rows) < valid
        untouched = (torch.arange(B * rows) % rows) >= valid           # the GEMV items write live rows only
        out = torch.full((B * rows, N), 7.0)
        ops.gemm(A, W, bias, _native.EPI_F32, out=out, rows_per_batch=rows, valid_rows=valid, algo=_native.GEMM_SIMPLE128)
        assert torch.allclose(out[live], ref[live], atol=2e-3, rtol=1e-4)
        assert bool((out[untouched] == 7.0).all())
        x0 = torch.randn(B * rows, N, generator=g)
        gate = torch.randn(B, N, generator=g)
        x, aux = x0.clone(), torch.zeros(B * rows, N, dtype=torch.bfloat16)
        ops.gemm(A, W, bias, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, rows_per_batch=rows, valid_rows=valid, aux=aux, resid=x0,
                 algo=_native.GEMM_SIMPLE128)
        want = x0 + gate.repeat_interleave(rows, 0) * ref
        assert torch.allclose(x[live], want[live], atol=5e-3, rtol=1e-4)
        assert torch.allclose(aux.float()[live], ref[live], atol=3e-2, rtol=1e-2)
        vt = torch.zeros(B, N, rows, dtype=torch.bfloat16)
        o = ops.gemm(A, W, bias, _native.EPI_BF16, rows_per_batch=rows, valid_rows=valid, vt=vt, algo=_native.GEMM_SIMPLE128)
        assert torch.allclose(o.float()[live], ref[live], atol=3e-2, rtol=1e-2)
        assert torch.equal(vt[:, :, :valid], o.reshape(B, rows, N).transpose(1, 2)[:, :, :valid])


def test_gemm_sliced_128_tiles(ops):
    """The sliced ring on 128 x 128 tiles, 4 waves of 64 x 64 (the N = 1024 GEMMs at one sample: one tile per CU): several trips
    round the ring, padding rows, the two-live-row GEMV side job, a live-block MFMA side job, and the epilogues those GEMMs use."""
    sl = _native.GEMM_SLICED128
    g = torch.Generator().manual_seed(47)
    M, N, K = 512, 256, 1024
    A = _bf(torch.randn(M, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.2)
    bias = torch.randn(N, generator=g)
    ref = A.float() @ W.float().t() + bias
    out = ops.gemm(A, W, bias, _native.EPI_F32, algo=sl)
    assert torch.allclose(out, ref, atol=3e-3, rtol=1e-4)
    for valid in (130, 150, 2):                      # 2 rows past a full tile (GEMV items) / 22 rows (MFMA items) / no full tile at all
        out = torch.full((M, N), 7.0)
        ops.gemm(A, W, bias, _native.EPI_F32, out=out, rows_per_batch=256, valid_rows=valid, algo=sl)
        for b in range(2):
            assert torch.allclose(out[b * 256:b * 256 + valid], ref[b * 256:b * 256 + valid], atol=3e-3, rtol=1e-4), valid
            assert bool((out[b * 256 + 160:(b + 1) * 256] == 7.0).all()), valid
    x0 = torch.randn(M, N, generator=g)
    gate = torch.randn(2, N, generator=g)
    x, aux = x0.clone(), torch.zeros(M, N, dtype=torch.bfloat16)
    ops.gemm(A, W, bias, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, rows_per_batch=256, valid_rows=130, aux=aux, resid=x0, algo=sl)
    live = (torch.arange(M) % 256) < 130
    want = x0 + gate.repeat_interleave(256, 0) * ref
    assert torch.allclose(x[live], want[live], atol=5e-3, rtol=1e-4)
    assert torch.allclose(aux.float()[live], ref[live], atol=3e-2, rtol=1e-2)
    vt = torch.zeros(2, N, 256, dtype=torch.bfloat16)
    o = ops.gemm(A, W, bias, _native.EPI_BF16, rows_per_batch=256, valid_rows=130, vt=vt, algo=sl)
    assert torch.allclose(o.float()[live], ref[live], atol=3e-2, rtol=1e-2)
    assert torch.equal(vt[:, :, :130], o.reshape(2, 256, N).transpose(1, 2)[:, :, :130])


_GENERIC_ATTENTION = r"""
import sys, torch
sys.path[:0] = [{root!r}, {pkg!r}, {tests!r}]
from dgs_amd.dit import DitOps
from emu_util import emu_lib
L, B, heads, out = int(sys.argv[1]), 2, 2, sys.argv[2]
d = torch.load(out + ".in")
lse2 = torch.zeros(B, heads, d["vt"].shape[2])
o = DitOps(lib=emu_lib()).attention(d["qk"], d["vt"], L, heads, lse2=lse2, q_prescaled=True)
torch.save(dict(o=o, lse2=lse2), out)
"""


@pytest.mark.parametrize("L", [520, 600, 1026])
def test_attention_steady_state_copies_equal_the_generic_loop(ops, L, tmp_path):
    """The branch-free steady-state rounds (dit_attention.hip: fast_tag / live_tag) against the generic loop the same library takes
    when a debug bit is set (DGS_ATTN_DBG is read once per process, hence the second process): the two execute the same
    arithmetic in the same order, so the outputs must be bit-identical -- queries, tail queries and log-sum-exp."""
    import os
    import subprocess
    import sys
    g = torch.Generator().manual_seed(L)
    B, heads = 2, 2
    lpad = (L + 127) // 128 * 128
    qk = _bf(torch.randn(B * lpad, 2 * heads * 64, generator=g))
    vt = _bf(torch.randn(B, heads * 64, lpad, generator=g))
    qk[3, :64] *= 6.0                                   # spiky query / key rows: the rescale branch
    qk[70, heads * 64:heads * 64 + 64] *= 6.0
    lse2 = torch.zeros(B, heads, lpad)
    fast = ops.attention(qk, vt, L, heads, lse2=lse2, q_prescaled=True)
    out = str(tmp_path / "generic.pt")
    torch.save(dict(qk=qk, vt=vt), out + ".in")
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    code = _GENERIC_ATTENTION.format(root=root, pkg=os.path.join(root, "open-diffusiongs_amd"), tests=here)
    env = dict(os.environ, DGS_ATTN_DBG="16")          # an unused debug bit: p.dbg != 0 keeps every round on the generic form
    subprocess.run([sys.executable, "-c", code, str(L), out], check=True, env=env, timeout=900)
    gen = torch.load(out)
    assert torch.equal(fast.view(torch.int16), gen["o"].view(torch.int16))
    assert torch.equal(lse2, gen["lse2"])
