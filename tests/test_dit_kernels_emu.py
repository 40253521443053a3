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
    rows, B, N, K = 256, 2, 128, 64
    A = _bf(torch.randn(B * rows, K, generator=g))
    W = _bf(torch.randn(N, K, generator=g) * 0.1)
    ref = A.float() @ W.float().t()
    for valid in (130, 33, 256):
        out = torch.full((B * rows, N), 7.0)
        ops.gemm(A, W, None, _native.EPI_F32, out=out, rows_per_batch=rows, valid_rows=valid)
        live = (valid + 31) // 32 * 32      # blocks with at least one valid row are computed completely
        for b in range(B):
            assert torch.allclose(out[b * rows: b * rows + live], ref[b * rows: b * rows + live], atol=1e-3, rtol=1e-4)
            assert bool((out[b * rows + live: (b + 1) * rows] == 7.0).all())
    # narrow-tile variant (chosen when 128 x 128 tiles would under-fill the chip) gives the same numbers
    out = ops.gemm(A, W, None, _native.EPI_F32)
    assert torch.allclose(out, ref, atol=1e-3, rtol=1e-4)


def test_gemm_qkv_epilogue(ops):
    g = torch.Generator().manual_seed(2)
    lpad, B, Wd, K = 128, 2, 128, 64
    A = _bf(torch.randn(B * lpad, K, generator=g))
    W = _bf(torch.randn(3 * Wd, K, generator=g) * 0.1)
    bias = torch.randn(3 * Wd, generator=g)
    ref = A.float() @ W.float().t() + bias
    qk, vt = ops.gemm(A, W, bias, _native.EPI_QKV, rows_per_batch=lpad)
    assert torch.allclose(qk.float(), ref[:, :2 * Wd], atol=2e-2, rtol=1e-2)
    v = ref[:, 2 * Wd:].reshape(B, lpad, Wd).transpose(1, 2)
    assert torch.allclose(vt.float(), v, atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("L", [128, 130, 67, 258, 520])
def test_attention(ops, L):
    g = torch.Generator().manual_seed(3)
    B, heads = 2, 2
    lpad = (L + 127) // 128 * 128
    q = torch.randn(B, heads, lpad, 64, generator=g)
    k = torch.randn(B, heads, lpad, 64, generator=g)
    v = torch.randn(B, heads, lpad, 64, generator=g)
    q[0, 0, 3] *= 6.0   # a spiky row: exercises the running-max rescale across tiles
    k[0, 0, 70 % L] *= 6.0
    qb, kb, vb = _bf(q), _bf(k), _bf(v)
    qk = torch.cat([qb.permute(0, 2, 1, 3).reshape(B * lpad, heads * 64), kb.permute(0, 2, 1, 3).reshape(B * lpad, heads * 64)], 1).contiguous()
    vt = vb.permute(0, 1, 3, 2).reshape(B, heads * 64, lpad).contiguous()
    out = ops.attention(qk, vt, L, heads).float().reshape(B, lpad, heads, 64).permute(0, 2, 1, 3)
    s = (qb.float() @ kb.float()[:, :, :L].transpose(-1, -2)) * 0.125
    ref = s.softmax(-1) @ vb.float()[:, :, :L]
    assert torch.allclose(out[:, :, :L], ref[:, :, :L], atol=2e-2, rtol=2e-2)


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
