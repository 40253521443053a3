// dit_gemm_epilogue.h -- row-major epilogue of the DiT GEMM kernels, staged through LDS.
//
// The MFMA D fragment gives a lane ONE output column and 16 rows: writing it straight to a row-major tensor costs 16
// two- or four-byte stores per 32 x 32 block per lane, each wave-instruction touching two 64/128-byte row segments.  In-kernel
// cycle stamps on MI355X: that store tail took 11k (QKV) / 21k (fc1 + GELU) / 40k (fc2 gate + residual) cycles per
// workgroup -- 25-45 % of the K loop it follows (the stores are issue-bound, not bandwidth-bound).
// Here a wave parks a 32 x (32 NB) strip of fp32 results in its private LDS patch and re-reads it row-major, so that
// every lane stores 16 contiguous bytes (8 bf16 or 4 fp32) and the residual / gate / aux operands are 16-byte loads too:
// 2-4 store instructions per block instead of 16.
#pragma once
#include "dit_common.h"
#include "dgs_dit.h"

namespace dgs {

__device__ __forceinline__ float epi_gelu_tanh(float x) {
    // nn.GELU(approximate="tanh"): 0.5 x (1 + tanh(u)) == x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3)
    //   = x / (1 + 2^(-2 log2(e) u)) : two multiplies, one fma, v_exp_f32, one add, v_rcp_f32 (1 ulp), one multiply --
    // an IEEE division and __expf's range handling cost ~25 VALU per element, which made this epilogue a third of fc1.
    const float t = x * __builtin_fmaf(x * x, -2.0f * 1.4426950408889634f * 0.7978845608028654f * 0.044715f, -2.0f * 1.4426950408889634f * 0.7978845608028654f);
    return x * hw_rcp(1.0f + hw_exp2(t));
}

// d/dx of the above: s + x s (1 - s) 2 sqrt(2/pi) (1 + 3 * 0.044715 x^2),  s = sigmoid(2u)
__device__ __forceinline__ float epi_dgelu_tanh(float x) {
    const float t = x * __builtin_fmaf(x * x, -2.0f * 1.4426950408889634f * 0.7978845608028654f * 0.044715f, -2.0f * 1.4426950408889634f * 0.7978845608028654f);
    const float sg = hw_rcp(1.0f + hw_exp2(t));
    return sg + x * sg * (1.0f - sg) * (2.0f * 0.7978845608028654f) * __builtin_fmaf(x * x, 3.0f * 0.044715f, 1.0f);
}

// One 32 x 32 output block straight from global memory (no LDS): acc += A[32 rows, 16 U] . W[32 rows, 16 U]^T with both operands
// read in MFMA fragment layout (arow / wrow already point at this lane's row and 8-element half).  Used for the tiles that hold a
// single live 32-row block.  All 2 U loads are issued before the first MFMA: such a block is a chain of L2 round trips, and
// each one costs 3-4 us when the rest of the chip is streaming GEMM tiles (measured: 8 of them made 16 of these tiles cost
// 36 us behind 256 full tiles that take 40 us).  Left to `#pragma unroll`, hipcc even emitted load, load, s_waitcnt vmcnt(0),
// MFMA per k-step -- one round trip per 16 columns of K.
template <int U>
__device__ __forceinline__ void direct_block_mfma(const bf16_t* arow, const bf16_t* wrow, f32x16& acc) {
    bf16x8 a[U], w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        a[u] = *reinterpret_cast<const bf16x8*>(arow + 16 * u);
        w[u] = *reinterpret_cast<const bf16x8*>(wrow + 16 * u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], w[u], acc, 0, 0, 0);
}

// dot product of 8 bf16 pairs held in two 16-byte registers, fp32 accumulate (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot8_bf16(const uint4& a, const uint4& b, float acc) {
#ifdef HIPEMU
    const unsigned x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
    for (int i = 0; i < 4; ++i)
        acc += __uint_as_float(x[i] << 16) * __uint_as_float(y[i] << 16) + __uint_as_float(x[i] & 0xffff0000u) * __uint_as_float(y[i] & 0xffff0000u);
    return acc;
#else
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a.x), __builtin_bit_cast(bf2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a.y), __builtin_bit_cast(bf2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a.z), __builtin_bit_cast(bf2_t, b.z), acc, false);
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a.w), __builtin_bit_cast(bf2_t, b.w), acc, false);
#endif
}

// Transposed copies ([feature][token], a lane owns ONE feature): the D-fragment layout gives lane (c, half) the tokens
// 8 g + 4 half + {0..3} of its feature, i.e. 8-byte pieces; 32 features per wave-instruction = 32 cache lines for 512 bytes, and the
// V third of the QKV tiles (and every tile of the training forward) left through four such instructions per block.  A half-wave
// exchange (v_permlane32_swap) of the pieces of token groups g and g + 1 gives the lower lane tokens 8 g .. 8 g + 7 and the upper
// lane 8 (g + 1) .. 8 (g + 1) + 7: one 16-byte store per lane, 32 contiguous bytes per feature and instruction, half the requests.
// `tdst` = the feature's row at the block's first token (no half offset); u0 / u1 = this lane's pieces of groups g and g + 1.
__device__ __forceinline__ void store_token_octet(bf16_t* tdst, int g, int half, uint2 u0, uint2 u1) {
    half_swap(u0.x, u1.x);
    half_swap(u0.y, u1.y);
    *reinterpret_cast<uint4*>(tdst + 8 * (g + half)) = make_uint4(u0.x, u0.y, u1.x, u1.y);
}

constexpr int epi_strip_bytes(int nb) { return 32 * (32 * nb + 4) * 4; }     // LDS per wave


// Which (epilogue, arguments) take the staged path: all of them.  (The training epilogues -- DGELU and the transposed `vt`
// copies of BF16 / GELU / DGELU -- used to keep the per-register path of their kernel: 16 two-byte loads and stores per block
// per lane made the fc2 input-gradient GEMM 489 us at 4 samples, 175 us more than the forward fc1 of the same shape.)
template <int EPI, class P>
__device__ __forceinline__ bool epi_staged(const P&) { return true; }

// acc[0 .. NB): NB side-by-side 32 x 32 accumulator blocks: rows m0 .. m0+31 (m0 = first row of the block), columns
// n0 .. n0 + 32 NB - 1.  `patch` = this wave's private LDS patch (epi_strip_bytes(NB) bytes); nobody else touches it, so no
// barrier is needed -- LDS operations of one wave execute in order.
// `pre`: the strip's residual values (GATE_RESIDUAL), loaded by the caller ahead of time in the layout of the row-major pass
// below (pre[it] = row it * RPI + rr, columns cc .. cc + 3; 4 NB of them): residual_prefetch.
template <int NB, class P>
__device__ __forceinline__ void residual_prefetch(const P& p, int m0, int n0, int lane, float4 (&pre)[4 * NB]) {
    constexpr int C = 32 * NB, LPR = C / 4, RPI = 64 / LPR;
    const int rr = lane / LPR, cc = (lane % LPR) * 4;
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) pre[it] = *reinterpret_cast<const float4*>(p.resid + (size_t)(m0 + it * RPI + rr) * p.ldo + n0 + cc);
}

template <int EPI, int NB, bool HAVE_PRE, class P>
__device__ __forceinline__ void store_strip_impl(const P& p, const f32x16* acc, int m0, int n0, int lane, char* patch, const float4 (&pre)[4 * NB]) {
    constexpr int C = 32 * NB, S = C + 4;
    const int c = lane & 31, half = lane >> 5;
    const int b = m0 / p.rows_per_batch;                       // a 32-row block never straddles samples
    if (EPI == DGS_EPI_QKV && n0 >= (p.N / 3) * 2) {
        // V features: only the transposed copy V^T[b][feature][token] exists: 4 consecutive tokens per 8-byte store
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            const int n = n0 + 32 * blk + c;
            const float bias = p.bias ? p.bias[n] : 0.0f;
            bf16_t* tdst = p.vt + ((size_t)b * (p.N / 3) + (n - (p.N / 3) * 2)) * p.rows_per_batch + (m0 - b * p.rows_per_batch);
#pragma unroll
            for (int g = 0; g < 4; g += 2)
                store_token_octet(tdst, g, half,
                                  make_uint2(pack_bf2(acc[blk][4 * g] + bias, acc[blk][4 * g + 1] + bias), pack_bf2(acc[blk][4 * g + 2] + bias, acc[blk][4 * g + 3] + bias)),
                                  make_uint2(pack_bf2(acc[blk][4 * g + 4] + bias, acc[blk][4 * g + 5] + bias), pack_bf2(acc[blk][4 * g + 6] + bias, acc[blk][4 * g + 7] + bias)));
        }
        return;
    }
    float* st = reinterpret_cast<float*>(patch);
    // ---- park: lane (column c, half) owns rows (r & 3) + 8 (r >> 2) + 4 half; a 32-lane group writes 32 consecutive floats ----
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
        const int n = n0 + 32 * blk + c;
        const float bias = p.bias ? p.bias[n] : 0.0f;
        const float qs = (EPI == DGS_EPI_QKV && n < p.N / 3) ? p.q_scale : 1.0f;      // pre-scaled queries
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            st[row * S + 32 * blk + c] = EPI == DGS_EPI_QKV ? (acc[blk][r] + bias) * qs : acc[blk][r] + bias;
        }
    }
    __builtin_amdgcn_wave_barrier();       // hardware: lanes run in lockstep, LDS operations of a wave execute in order
    // ---- row-major: 16 bytes per lane ----
    if (EPI == DGS_EPI_F32 || EPI == DGS_EPI_GATE_RESIDUAL) {
        constexpr int LPR = C / 4, RPI = 64 / LPR;             // lanes per row, rows per instruction
        const int rr = lane / LPR, cc = (lane % LPR) * 4;
        float4 gate = make_float4(0.f, 0.f, 0.f, 0.f);
        if (EPI == DGS_EPI_GATE_RESIDUAL) gate = *reinterpret_cast<const float4*>(p.gate + (size_t)b * p.gate_stride + n0 + cc);
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row = it * RPI + rr;
            const float4 v = *reinterpret_cast<const float4*>(st + row * S + cc);
            const size_t o = (size_t)(m0 + row) * p.ldo + n0 + cc;
            if (EPI == DGS_EPI_GATE_RESIDUAL) {
                float4 x;
                if constexpr (HAVE_PRE) x = pre[it]; else x = *reinterpret_cast<const float4*>(p.resid + o);
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) =
                    make_float4(x.x + gate.x * v.x, x.y + gate.y * v.y, x.z + gate.z * v.z, x.w + gate.w * v.w);
                if (p.aux) *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.aux) + o) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + o) = v;
            }
        }
    } else {
        constexpr int LPR = C / 8, RPI = 64 / LPR;
        const int rr = lane / LPR, cc = (lane % LPR) * 8;
        // BF16 / GELU / DGELU with `vt`: the final values go back into the patch (each lane over what it has just read) and
        // leave a second time in the D-fragment layout: 4 consecutive tokens of one feature per 8-byte store
        const bool transposed = EPI != DGS_EPI_QKV && p.vt != nullptr;
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int row = it * RPI + rr;
            float4 v0 = *reinterpret_cast<const float4*>(st + row * S + cc), v1 = *reinterpret_cast<const float4*>(st + row * S + cc + 4);
            const size_t o = (size_t)(m0 + row) * p.ldo + n0 + cc;
            if (EPI == DGS_EPI_GELU_BF16) {
                if (p.aux) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.aux) + o) =
                    make_uint4(pack_bf2(v0.x, v0.y), pack_bf2(v0.z, v0.w), pack_bf2(v1.x, v1.y), pack_bf2(v1.z, v1.w));
                v0 = make_float4(epi_gelu_tanh(v0.x), epi_gelu_tanh(v0.y), epi_gelu_tanh(v0.z), epi_gelu_tanh(v0.w));
                v1 = make_float4(epi_gelu_tanh(v1.x), epi_gelu_tanh(v1.y), epi_gelu_tanh(v1.z), epi_gelu_tanh(v1.w));
            } else if (EPI == DGS_EPI_DGELU_BF16) {
                const uint4 ax = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(p.aux) + o);   // 8 pre-activations
                v0 = make_float4(v0.x * epi_dgelu_tanh(__uint_as_float(ax.x << 16)), v0.y * epi_dgelu_tanh(__uint_as_float(ax.x & 0xffff0000u)),
                                 v0.z * epi_dgelu_tanh(__uint_as_float(ax.y << 16)), v0.w * epi_dgelu_tanh(__uint_as_float(ax.y & 0xffff0000u)));
                v1 = make_float4(v1.x * epi_dgelu_tanh(__uint_as_float(ax.z << 16)), v1.y * epi_dgelu_tanh(__uint_as_float(ax.z & 0xffff0000u)),
                                 v1.z * epi_dgelu_tanh(__uint_as_float(ax.w << 16)), v1.w * epi_dgelu_tanh(__uint_as_float(ax.w & 0xffff0000u)));
            }
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(p.out) + o) =
                make_uint4(pack_bf2(v0.x, v0.y), pack_bf2(v0.z, v0.w), pack_bf2(v1.x, v1.y), pack_bf2(v1.z, v1.w));
            if (transposed && EPI != DGS_EPI_BF16) {
                *reinterpret_cast<float4*>(st + row * S + cc) = v0;
                *reinterpret_cast<float4*>(st + row * S + cc + 4) = v1;
            }
        }
        if (transposed) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                bf16_t* tdst = p.vt + ((size_t)b * p.N + n0 + 32 * blk + c) * p.rows_per_batch + (m0 - b * p.rows_per_batch);
#pragma unroll
                for (int g = 0; g < 4; g += 2) {
                    const float* src = st + (4 * half + 8 * g) * S + 32 * blk + c;
                    store_token_octet(tdst, g, half, make_uint2(pack_bf2(src[0], src[S]), pack_bf2(src[2 * S], src[3 * S])),
                                      make_uint2(pack_bf2(src[8 * S], src[9 * S]), pack_bf2(src[10 * S], src[11 * S])));
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();       // the patch is rewritten by the next strip
}

template <int EPI, int NB, class P>
__device__ __forceinline__ void store_strip(const P& p, const f32x16* acc, int m0, int n0, int lane, char* patch) {
    const float4 none[4 * NB] = {};
    store_strip_impl<EPI, NB, false>(p, acc, m0, n0, lane, patch, none);
}
template <int EPI, int NB, class P>
__device__ __forceinline__ void store_strip(const P& p, const f32x16* acc, int m0, int n0, int lane, char* patch, const float4 (&pre)[4 * NB]) {
    store_strip_impl<EPI, NB, true>(p, acc, m0, n0, lane, patch, pre);
}

// One output element (row m, column n) through epilogue EPI: the scalar twin of store_strip, same arithmetic and roundings.
// Used by the GEMV items that compute the one or two live rows behind a sample's last full tile.
struct TailOperands { float bias, resid, gate; };           // what a GEMV item can load BEFORE its dot products: no second round trip
template <int EPI, class P>
__device__ __forceinline__ TailOperands tail_prefetch(const P& p, int m, int n) {
    TailOperands t{0.f, 0.f, 0.f};
    if (p.bias) t.bias = p.bias[n];
    if (EPI == DGS_EPI_GATE_RESIDUAL) {
        t.resid = p.resid[(size_t)m * p.ldo + n];
        t.gate = p.gate[(size_t)(m / p.rows_per_batch) * p.gate_stride + n];
    }
    return t;
}
template <int EPI, class P>
__device__ __forceinline__ void tail_store(const P& p, int m, int n, float v, const TailOperands& t) {
    v += t.bias;
    const int b = m / p.rows_per_batch;
    const size_t o = (size_t)m * p.ldo + n;
    auto bf1 = [](float x) { return (bf16_t)(pack_bf2(x, 0.0f) & 0xffffu); };
    if (EPI == DGS_EPI_F32) { reinterpret_cast<float*>(p.out)[o] = v; return; }
    if (EPI == DGS_EPI_QKV) {                                   // q (pre-scaled) | k row-major, V only as V^T[b][feature][token]
        const int third = p.N / 3;
        if (n >= 2 * third) p.vt[((size_t)b * third + (n - 2 * third)) * p.rows_per_batch + (m - b * p.rows_per_batch)] = bf1(v);
        else reinterpret_cast<bf16_t*>(p.out)[o] = bf1(n < third ? v * p.q_scale : v);
        return;
    }
    if (EPI == DGS_EPI_GATE_RESIDUAL) {
        reinterpret_cast<float*>(p.out)[o] = t.resid + t.gate * v;
        if (p.aux) reinterpret_cast<bf16_t*>(p.aux)[o] = bf1(v);
        return;
    }
    if (EPI == DGS_EPI_GELU_BF16) {
        if (p.aux) reinterpret_cast<bf16_t*>(p.aux)[o] = bf1(v);
        v = epi_gelu_tanh(v);
    } else if (EPI == DGS_EPI_DGELU_BF16) {
        v *= epi_dgelu_tanh(__uint_as_float((unsigned)reinterpret_cast<const bf16_t*>(p.aux)[o] << 16));
    }
    reinterpret_cast<bf16_t*>(p.out)[o] = bf1(v);
    if (p.vt) p.vt[((size_t)b * p.N + n) * p.rows_per_batch + (m - b * p.rows_per_batch)] = bf1(v);
}


}  // namespace dgs
