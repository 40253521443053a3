// dit_attention_backward.hip -- flash-attention backward for the DiT blocks (gfx950, wave64, bf16 MFMA).
//
// The reference obtains these gradients from torch autograd through F.scaled_dot_product_attention (timm 0.9.16
// Attention.forward, used by DiTBlock, utils_transformer.py:254-256,286).  With P = softmax(S), S = scale * Q K^T:
//     D_q  = sum_d dO[q,d] O[q,d]
//     dV   = P^T dO            dP = dO V^T            dS = P o (dP - D) * scale
//     dQ   = dS K              dK = dS^T Q
// VALU diet (both loops were VALU-bound at 13 VALU per MFMA): the queries enter the S MFMAs as bf16(-scale log2(e) q)
// (the same rounding as the forward's pre-scaled queries, sign flipped) and the S accumulators START at +lse, so the
// matrix pipe hands back lse - S' and P = exp2(-(.)) is a bare v_exp_f32 with a source negation; the dP accumulators START
// at -D (the D scratch holds -D), so dP - D needs no instruction either; `scale` is applied once to the dQ / dK
// accumulators at the end, and the ragged last tile has its own loop copy: 2.5-3 VALU per score instead of 10.
// Two kernels, both recomputing S from the saved log-sum-exp (no S x S matrix is ever stored), both free of atomics:
//   attention_bwd_dq_kernel   one workgroup per 256-query block (like the forward): walks the key tiles, accumulates dQ; also
//                             produces D.
//   attention_bwd_dkv_kernel  one workgroup per 256-key block: walks the query tiles, accumulates dK and dV.
// As in the forward, every MFMA is arranged so that the reduction index of the NEXT MFMA is the accumulator-row index of
// the previous one (P / dS accumulator registers become B fragments after bf16 packing), and the operands that have to be
// read "transposed" come from token-contiguous copies ([B, features, lpad]) written by the producing GEMM epilogues.
#include "dit_common.h"
#include <stdio.h>
#include <stdlib.h>
#include "dgs_dit.h"

namespace dgs {

constexpr int BQ = 256, BNW = 8, BT = 64;      // rows per workgroup, waves, rows per tile of the walked dimension
constexpr int TILE_B = BT * 64 * 2;            // 8 KiB: one [64][64] bf16 tile
constexpr int DQ_STAGE = 3 * TILE_B, DKV_STAGE = 4 * TILE_B + 512, BWD_RING = 3;   // LDS bytes per staged tile set, stages

struct AttnBwdParams {
    int B, heads, L, lpad, ld;                 // ld: row stride of the row-major q / k / v / o / do / dqkv tensors
    long long t_batch_stride_qkv, t_batch_stride_do;   // element stride between samples of the transposed tensors
    const bf16_t *q, *k, *v;                   // row-major [B*lpad, ld] (already offset to the q / k / v feature blocks)
    const bf16_t *qT, *kT;                     // transposed [B, *, lpad] (already offset to the q / k feature blocks)
    const bf16_t *o, *dO, *dOT;                // o, dO row-major [B*lpad, ld_o]; dOT transposed [B, W, lpad]
    int ld_o;
    const float* lse2;                         // [B, heads, lpad]  log2-domain log-sum-exp of the forward
    float* D;                                  // [B, heads, lpad]
    bf16_t *dq, *dk, *dv;                      // row-major [B*lpad, ld_d] (offset to the dq / dk / dv feature blocks)
    int ld_d;
    float scale, scale_log2e;
    int nmain, ntail;                          // 256-row blocks walked by the MFMA workgroups; tokens behind them (tail roles) or 0
    // optional by-products the training backward used to take from two more kernels (transpose_kernel, colsum_wide_kernel): the
    // gradients once more token-contiguous -- the accumulators ARE feature-major with the token in the lane, so the copy is a plain
    // coalesced store -- and their per-workgroup column sums (the qkv bias gradient's partial rows, summed in slot order by col_reduce)
    bf16_t *dqT, *dkT, *dvT;                   // [B, *, lpad] (offset to the dq / dk / dv feature blocks of a [B, 3W, lpad] tensor) or nullptr
    float* bias_part;                          // [B * (nmain + ntail)][bias_stride] or nullptr: slot b * (nmain + ntail) + block, columns as in dqkv
    int bias_stride;
    int dbg;                                   // instrumented library only (DGS_ATTN_DBG & 16: phase stamps of the dK/dV loop)
};
__device__ long long dgs_attn_bwd_dbg[2 * 8 * 8];       // [wave 0 | wave 5][tiles 20..27][tile top, scores done, products done, published, barrier]

// row-major [64][64] tile, 16-byte chunk swizzle c ^ ((row >> 1) & 7)
__device__ __forceinline__ void put_rows(char* tile, int r, int c, uint4 v) {
    *reinterpret_cast<uint4*>(tile + r * 128 + ((c ^ ((r >> 1) & 7)) << 4)) = v;
}
__device__ __forceinline__ bf16x8 get_rows(const char* tile, int row, int ks, int half) {
    return *reinterpret_cast<const bf16x8*>(tile + row * 128 + (((2 * ks + half) ^ ((row >> 1) & 7)) << 4));
}
// transposed [64 features][64 tokens] tile: every 16-token group stored as [t0-3, t8-11 | t4-7, t12-15] so that the 8
// tokens a lane contributes to one MFMA k-step (accumulator rows 4h..4h+3, 8+4h..8+4h+3) are ONE 16-byte slot (2g + h)
__device__ __forceinline__ void put_perm(char* tile, int r, int c, uint4 v) {
    const int sw = (r >> 1) & 7, slot = (c >> 1) * 2;
    *reinterpret_cast<uint2*>(tile + r * 128 + ((slot ^ sw) << 4) + 8 * (c & 1)) = make_uint2(v.x, v.y);
    *reinterpret_cast<uint2*>(tile + r * 128 + (((slot + 1) ^ sw) << 4) + 8 * (c & 1)) = make_uint2(v.z, v.w);
}
__device__ __forceinline__ bf16x8 pack_rows(const f32x16& s, int r0) {
    union { bf16x8 v; uint32_t u[4]; } f;
#pragma unroll
    for (int j = 0; j < 4; ++j) f.u[j] = pack_bf2(s[r0 + 2 * j], s[r0 + 2 * j + 1]);
    return f.v;
}
// 1-D grids, XCD-aware.  Consecutive workgroup ids go round-robin over the 8 XCDs, each with its own L2, and the workgroups of one
// (sample, head) all walk the SAME tiles of that head.  With a (block, head, sample) grid a head's 16 workgroups sat on all 8 XCDs
// and every L2 fetched every head's tiles.  Here group g = sample * heads + head runs entirely on XCD g % 8: measured at L = 4098,
// FETCH_SIZE of the dK / dV kernel 5.4 x lower, the pair of kernels -11 % at 4 samples and -32 % at 1 (one round of workgroups: the
// slowest XCD no longer sets the time).
__device__ __forceinline__ void block_coords(const AttnBwdParams& p, int& blk, int& head, int& b) {
    const int id = blockIdx.x, nblk = p.nmain + p.ntail, groups = p.heads * p.B;
    int g;
    if (groups % 8 == 0) { const int slot = id >> 3; g = (id & 7) + 8 * (slot / nblk); blk = slot % nblk; }
    else { g = id / nblk; blk = id % nblk; }
    head = g % p.heads; b = g / p.heads;
}
template <bool V> struct BoolTag { static constexpr bool value = V; };
// Knock-out builds (tools/knockout_build.sh: -DDGS_INSTRUMENT -DDGS_KNOCK=n, timing only, wrong results): what one ingredient of the
// dK / dV loop costs.  1 no exponentials, 2 no dS (multiply + pack), 3 no statistics reads, 4 no query pre-scaling in the publish,
// 5 no product MFMAs (and their fragment reads), 6 no score fragment reads, 7 score fragment reads into registers no MFMA uses.  The product library compiles none of it.
#if defined(DGS_INSTRUMENT) && defined(DGS_KNOCK)
constexpr int kKnock = DGS_KNOCK;
#else
constexpr int kKnock = 0;
#endif
#define DGS_SCHED_FENCE() sched_fence()
__device__ __forceinline__ bf16x8 words4(const uint32_t* w) {
    union { bf16x8 v; uint32_t u[4]; } f;
    f.u[0] = w[0]; f.u[1] = w[1]; f.u[2] = w[2]; f.u[3] = w[3];
    return f.v;
}
// The VALU work of accumulator registers (2U, 2U + 1) of one 32-row block: P = exp2(-(lse - S')), dS / scale = P (dP - D), packed
// to bf16 pairs (word U of the block: the B fragment of k-step U / 4 of the block's second-stage MFMAs is words 4 (U / 4) .. + 3).
// One such unit rides in the shadow of ONE MFMA of the other block (see the tile bodies).
template <bool RAGGED, bool WITH_P, int U>
__device__ __forceinline__ void softmax_grad_unit(const f32x16& s, const f32x16& e, uint32_t* pw, uint32_t* dw, int row0, int L) {
    constexpr int r = 2 * U;
    float p0 = kKnock == 1 ? -s[r] : fast_exp2(-s[r]), p1 = kKnock == 1 ? -s[r + 1] : fast_exp2(-s[r + 1]);
    float d0 = e[r], d1 = e[r + 1];
    if (RAGGED) {                                // rows >= L of the walked dimension: no valid statistics, contribute nothing
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        if (row >= L) { p0 = 0.f; d0 = 0.f; }
        if (row + 1 >= L) { p1 = 0.f; d1 = 0.f; }
    }
    if (WITH_P) pw[U] = pack_bf2(p0, p1);
    if constexpr (kKnock == 2 && WITH_P) { dw[U] = pw[U]; return; }
    dw[U] = pack_bf2(p0 * d0, p1 * d1);
}

__device__ __forceinline__ f32x16 zero_acc() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return z;
}


// ------------------------------------------------------------------------------------------------------------------
// The tokens behind the last full 256-block (the model's sequences are V x patches + 2: L % 256 = 2).  As one more block of the
// two grids they cost a whole workgroup walk each for two live lanes: with B x heads = 64 that was a fifth round of workgroups
// on the 256 CUs behind four full ones (and a second round behind one at B = 1).  Instead one extra workgroup per tail token
// (block >= nmain) does that token's row of dQ (and D), or of dK / dV, on the VALU: eight lanes per walked row (one 16-byte
// chunk of its 64 features each), dot products by v_dot2c_f32_bf16 and a sum over the eight lanes, with the operand roundings of
// the MFMA path (bf16(-scale log2(e) q), bf16 k / v / dO; P and dS stay fp32 here); the 64 row slots of the workgroup are summed
// in a fixed order (lanes, then waves through LDS): as deterministic as the MFMA path.  A few % of one MFMA workgroup's time.
// ------------------------------------------------------------------------------------------------------------------
constexpr int BWD_TAIL_MAX = 8;

__device__ __forceinline__ uint4 prescaled(uint4 q, float c) {
    q.x = scale_bf2(q.x, c); q.y = scale_bf2(q.y, c); q.z = scale_bf2(q.z, c); q.w = scale_bf2(q.w, c);
    return q;
}
__device__ __forceinline__ void axpy8(float (&acc)[8], float a, uint4 x) {
    acc[0] += a * bf2f(x.x & 0xffffu); acc[1] += a * bf2f(x.x >> 16); acc[2] += a * bf2f(x.y & 0xffffu); acc[3] += a * bf2f(x.y >> 16);
    acc[4] += a * bf2f(x.z & 0xffffu); acc[5] += a * bf2f(x.z >> 16); acc[6] += a * bf2f(x.w & 0xffffu); acc[7] += a * bf2f(x.w >> 16);
}
// sum of acc over the workgroup's 64 row slots (lanes of equal chunk, then the waves in order) -> out[64] (bf16, times c), tid < 64
__device__ __forceinline__ void tail_reduce_store(float (&acc)[8], float* red, float c, bf16_t* out, bf16_t* outT = nullptr, int ldT = 0,
                                                  float* bias = nullptr) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v = acc[i];
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v = xor32_sum(v);
        if (lane < 8) red[wave * 64 + lane * 8 + i] = v;
    }
    __syncthreads();
    if (tid < 64) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < BNW; ++w) v += red[w * 64 + tid];
        out[tid] = (bf16_t)f2bf(c * v);
        if (outT) outT[(size_t)tid * ldT] = (bf16_t)f2bf(c * v);         // the token's column of the transposed copy
        if (bias) bias[tid] = c * v;                                       // a one-token workgroup: its partial row is the row itself
    }
    __syncthreads();
}
// rows [L, end of the 32-row unit L falls into) of one head's 64 columns: the exact zeros the MFMA path's last block used to store
__device__ __forceinline__ void tail_zero_rows(const AttnBwdParams& p, bf16_t* base /* row 0 of the sample, head's column 0 */) {
    const int first = p.L, last = min(p.lpad, (p.L + 31) / 32 * 32);
    const int tid = threadIdx.x;
    if (tid < (last - first) * 8) *reinterpret_cast<uint4*>(base + (size_t)(first + (tid >> 3)) * p.ld_d + (tid & 7) * 8) = make_uint4(0u, 0u, 0u, 0u);
}

// The walk of a tail workgroup: 64 row slots x TAIL_UNR rows per batch, the next batch's loads in flight while this one is used
// (one row per slot and trip was 65 dependent L2 / HBM round trips: the tail workgroup took as long as an MFMA one).
constexpr int TAIL_UNR = 8, TAIL_BATCH = 64 * TAIL_UNR;
struct TailRows { uint4 a[TAIL_UNR], b[TAIL_UNR]; float s0[TAIL_UNR], s1[TAIL_UNR]; };
template <class Fetch, class Use>
__device__ __forceinline__ void tail_walk(int L, Fetch&& fetch, Use&& use) {
    TailRows A, B;
    fetch(A, 0);
    for (int r0 = 0; r0 < L; r0 += 2 * TAIL_BATCH) {         // (uniform trip count: the lane sums run with the whole wave)
        fetch(B, r0 + TAIL_BATCH);
        use(A, r0);
        fetch(A, r0 + 2 * TAIL_BATCH);
        use(B, r0 + TAIL_BATCH);
    }
}

__device__ __forceinline__ void tail_query_role(const AttnBwdParams& p, int blk, int head, int b, float* red) {
    const int tid = threadIdx.x, c = tid & 7, slot = tid >> 3;
    const int tq = p.nmain * BQ + (blk - p.nmain);
    const size_t row0 = (size_t)b * p.lpad;
    const bf16_t* Kg = p.k + row0 * p.ld + head * 64 + c * 8;
    const bf16_t* Vg = p.v + row0 * p.ld + head * 64 + c * 8;
    const uint4 qs = prescaled(*reinterpret_cast<const uint4*>(p.q + (row0 + tq) * p.ld + head * 64 + c * 8), -p.scale_log2e);
    const uint4 dov = *reinterpret_cast<const uint4*>(p.dO + (row0 + tq) * p.ld_o + head * 64 + c * 8);
    const uint4 ov = *reinterpret_cast<const uint4*>(p.o + (row0 + tq) * p.ld_o + head * 64 + c * 8);
    const float Dq = oct_sum(dot8_bf16(ov, dov));
    const size_t stat = ((size_t)b * p.heads + head) * p.lpad + tq;
    const float lse = p.lse2[stat];
    if (tid == 0) p.D[stat] = -Dq;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    tail_walk(p.L,
        [&](TailRows& r, int k0) {
#pragma unroll
            for (int u = 0; u < TAIL_UNR; ++u) {
                const int k = min(k0 + slot + 64 * u, p.L - 1);            // rows past L: loaded from the last one, not used
                r.a[u] = *reinterpret_cast<const uint4*>(Kg + (size_t)k * p.ld);
                r.b[u] = *reinterpret_cast<const uint4*>(Vg + (size_t)k * p.ld);
            }
        },
        [&](const TailRows& r, int k0) {
#pragma unroll
            for (int u = 0; u < TAIL_UNR; ++u) {
                const float s = oct_sum(dot8_bf16(qs, r.a[u])), dp = oct_sum(dot8_bf16(dov, r.b[u]));      // s = -scale log2(e) q . k
                axpy8(acc, k0 + slot + 64 * u < p.L ? fast_exp2(-(lse + s)) * (dp - Dq) : 0.f, r.a[u]);
            }
        });
    bf16_t* dq_head = p.dq + row0 * p.ld_d + head * 64;
    tail_reduce_store(acc, red, p.scale, dq_head + (size_t)tq * p.ld_d,
                      p.dqT ? p.dqT + (size_t)b * p.t_batch_stride_qkv + (size_t)head * 64 * p.lpad + tq : nullptr, p.lpad,
                      p.bias_part ? p.bias_part + ((size_t)b * (p.nmain + p.ntail) + blk) * p.bias_stride + head * 64 : nullptr);
    if (blk == p.nmain) tail_zero_rows(p, dq_head);
}

__device__ __forceinline__ void tail_key_role(const AttnBwdParams& p, int blk, int head, int b, float* red) {
    const int tid = threadIdx.x, c = tid & 7, slot = tid >> 3;
    const int tk = p.nmain * BQ + (blk - p.nmain);
    const size_t row0 = (size_t)b * p.lpad;
    const bf16_t* Qg = p.q + row0 * p.ld + head * 64 + c * 8;
    const bf16_t* dOg = p.dO + row0 * p.ld_o + head * 64 + c * 8;
    const float* lseg = p.lse2 + ((size_t)b * p.heads + head) * p.lpad;
    const float* Dg = p.D + ((size_t)b * p.heads + head) * p.lpad;
    const uint4 kk = *reinterpret_cast<const uint4*>(p.k + (row0 + tk) * p.ld + head * 64 + c * 8);
    const uint4 vk = *reinterpret_cast<const uint4*>(p.v + (row0 + tk) * p.ld + head * 64 + c * 8);
    float ak[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, av[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    tail_walk(p.L,
        [&](TailRows& r, int q0) {
#pragma unroll
            for (int u = 0; u < TAIL_UNR; ++u) {
                const int q = min(q0 + slot + 64 * u, p.L - 1);
                r.a[u] = *reinterpret_cast<const uint4*>(Qg + (size_t)q * p.ld);
                r.b[u] = *reinterpret_cast<const uint4*>(dOg + (size_t)q * p.ld_o);
                r.s0[u] = lseg[q];
                r.s1[u] = Dg[q];
            }
        },
        [&](const TailRows& r, int q0) {
#pragma unroll
            for (int u = 0; u < TAIL_UNR; ++u) {
                const float s = oct_sum(dot8_bf16(prescaled(r.a[u], -p.scale_log2e), kk)), dp = oct_sum(dot8_bf16(r.b[u], vk));
                const float pr = q0 + slot + 64 * u < p.L ? fast_exp2(-(r.s0[u] + s)) : 0.f;
                axpy8(av, pr, r.b[u]);
                axpy8(ak, pr * (dp + r.s1[u]), r.a[u]);        // the scratch holds -D
            }
        });
    bf16_t* dk_head = p.dk + row0 * p.ld_d + head * 64;
    bf16_t* dv_head = p.dv + row0 * p.ld_d + head * 64;
    const size_t tcol = (size_t)b * p.t_batch_stride_qkv + (size_t)head * 64 * p.lpad + tk;
    float* const brow = p.bias_part ? p.bias_part + ((size_t)b * (p.nmain + p.ntail) + blk) * p.bias_stride + head * 64 : nullptr;
    const int W = p.heads * 64;
    tail_reduce_store(ak, red, p.scale, dk_head + (size_t)tk * p.ld_d, p.dkT ? p.dkT + tcol : nullptr, p.lpad, brow ? brow + W : nullptr);
    tail_reduce_store(av, red, 1.f, dv_head + (size_t)tk * p.ld_d, p.dvT ? p.dvT + tcol : nullptr, p.lpad, brow ? brow + 2 * W : nullptr);
    if (blk == p.nmain) { tail_zero_rows(p, dk_head); tail_zero_rows(p, dv_head); }
}

// By-products of an MFMA workgroup's 256 tokens x 64 features (two accumulators of 32 features: register r of `lo` / `hi` is feature
// 8 (r >> 2) + 4 half + (r & 3) (+ 32), the lane's l31 is the token): the token-contiguous copy and the column sums over the workgroup's
// tokens.  Both through ONE fp32 image of the block in LDS, [feature][token] (65 KiB of the ring nobody reads any more): the accumulator
// registers are written as they lie (a register = 32 consecutive tokens of one feature: conflict-free), then 32 lanes take a feature's 256
// tokens eight apiece -- 16-byte stores of 8 bf16, 512 contiguous bytes per feature -- and their sum is the feature's partial.  (First
// form of this round: one 2-byte store per register and an xor-shuffle tree per register -- 640 ds_bpermute per wave in the dK / dV
// kernel; it cost the two kernels +75 us per block and the launches it replaced had cost 67: profiles/r06_attn_bwd_byproducts_ab.txt.)
// Every thread of the workgroup calls it; invalid tokens (>= L: padding rows of the last block, waves without a live token) enter as zeros.
constexpr int BYP_LD = 256 + 4;                // floats per feature row of the image (16-byte aligned rows, reads of 8 lanes span all banks)
__device__ __forceinline__ void block_byproducts(const f32x16& lo, const f32x16& hi, float c, bool token_ok, bf16_t* rowT /* feature 0, the block's token 0 */,
                                                 int lpad, int tokens_here /* lpad - the block's first token: the last block may reach past lpad */,
                                                 float* bias /* 64 floats or nullptr */, float* img, int wave, int l31, int half) {
    const int tok = wave * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = 8 * (r >> 2) + 4 * half + (r & 3);
        img[f * BYP_LD + tok] = token_ok ? c * lo[r] : 0.f;
        img[(32 + f) * BYP_LD + tok] = token_ok ? c * hi[r] : 0.f;
    }
    __syncthreads();
    const int ch = threadIdx.x & 31;                               // tokens 8 ch .. 8 ch + 7
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int f = 16 * pass + (threadIdx.x >> 5);
        const float4 v0 = *reinterpret_cast<const float4*>(img + f * BYP_LD + 8 * ch), v1 = *reinterpret_cast<const float4*>(img + f * BYP_LD + 8 * ch + 4);
        if (rowT && 8 * ch < tokens_here) *reinterpret_cast<uint4*>(rowT + (size_t)f * lpad + 8 * ch) = make_uint4(pack_bf2(v0.x, v0.y), pack_bf2(v0.z, v0.w), pack_bf2(v1.x, v1.y), pack_bf2(v1.z, v1.w));
        if (bias) {
            float sum = ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
#pragma unroll
            for (int d = 1; d < 32; d <<= 1) sum += __shfl_xor(sum, d);
            if (ch == 0) bias[f] = sum;
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------------
// dQ (and D).  Lane = query (column of every accumulator).  Per 64-key tile:
//   S^T  = K . Q^T         (A: K rows from LDS,  B: Q fragment registers)
//   dP^T = V . dO^T        (A: V rows from LDS,  B: dO fragment registers)
//   dQ^T += K^T . dS^T     (A: K^T permuted tile, B: packed dS^T accumulator registers)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void attention_bwd_dq_kernel(AttnBwdParams p) {
    DGS_DYNAMIC_LDS(lds);                                                // [3 stages][K | V | K^T] : 72 KiB
    constexpr int STAGE = DQ_STAGE;
    int qblk, head, b;
    block_coords(p, qblk, head, b);
    if (qblk >= p.nmain) { tail_query_role(p, qblk, head, b, reinterpret_cast<float*>(lds)); return; }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const size_t row0 = (size_t)b * p.lpad;
    const int q_raw = qblk * BQ + wave * 32 + l31;
    const int q = q_raw < p.lpad ? q_raw : p.lpad - 1;          // the last block may reach past lpad: clamp loads, never store
    const bool wave_live = qblk * BQ + wave * 32 < p.L;
    const bf16_t* Kg = p.k + row0 * p.ld + head * 64;
    const bf16_t* Vg = p.v + row0 * p.ld + head * 64;
    const bf16_t* KTg = p.kT + (size_t)b * p.t_batch_stride_qkv + (size_t)head * 64 * p.lpad;

    bf16x8 qf[4], dof[4];
    float Dq = 0.f;
    {
        const bf16_t* qp = p.q + (row0 + q) * p.ld + head * 64;
        const bf16_t* op = p.o + (row0 + q) * p.ld_o + head * 64;
        const bf16_t* dp = p.dO + (row0 + q) * p.ld_o + head * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 raw = *reinterpret_cast<const uint4*>(qp + (2 * ks + half) * 8);      // -> bf16(-scale log2(e) q)
            raw.x = scale_bf2(raw.x, -p.scale_log2e); raw.y = scale_bf2(raw.y, -p.scale_log2e);
            raw.z = scale_bf2(raw.z, -p.scale_log2e); raw.w = scale_bf2(raw.w, -p.scale_log2e);
            qf[ks] = __builtin_bit_cast(bf16x8, raw);
            dof[ks] = *reinterpret_cast<const bf16x8*>(dp + (2 * ks + half) * 8);
            const bf16x8 of = *reinterpret_cast<const bf16x8*>(op + (2 * ks + half) * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) Dq += bf2f((uint16_t)of[j]) * bf2f((uint16_t)dof[ks][j]);
        }
        Dq = xor32_sum(Dq);
    }
    const size_t stat = ((size_t)b * p.heads + head) * p.lpad + q;
    const float lse = p.lse2[stat];
    if (half == 0 && q_raw < p.lpad) p.D[stat] = -Dq;          // the scratch holds -D: it is the dP accumulators' initial value
    f32x16 lse16, negd16;                                       // per-lane (= per-query) constants as MFMA C operands
#pragma unroll
    for (int r = 0; r < 16; ++r) { lse16[r] = lse; negd16[r] = -Dq; }

    f32x16 dq0 = zero_acc(), dq1 = zero_acc();
    const int ntiles = (p.L + BT - 1) / BT;
    const int sr = tid >> 3, sc = tid & 7;            // staging chunk: row sr (key / feature), 16-byte column sc
    uint4 kreg, vreg, ktreg;
    auto issue = [&](int t) {
        kreg = *reinterpret_cast<const uint4*>(Kg + (size_t)(t * BT + sr) * p.ld + sc * 8);
        vreg = *reinterpret_cast<const uint4*>(Vg + (size_t)(t * BT + sr) * p.ld + sc * 8);
        ktreg = *reinterpret_cast<const uint4*>(KTg + (size_t)sr * p.lpad + t * BT + sc * 8);
    };
    auto publish_piece = [&](int stage, int piece) {
        char* base = lds + stage * STAGE;
        if (piece == 0) put_rows(base, sr, sc, kreg);
        if (piece == 1) put_rows(base + TILE_B, sr, sc, vreg);
        if (piece == 2) put_perm(base + 2 * TILE_B, sr, sc, ktreg);
    };
    auto publish = [&](int stage) { publish_piece(stage, 0); publish_piece(stage, 1); publish_piece(stage, 2); };
    // Three stages, ONE barrier per tile: tile t + 2 is fetched (registers) at the top of tile t and published behind its last
    // MFMAs, into the stage tile t - 1 was read from (every wave left that tile through the barrier before this one); the barrier
    // at the bottom of tile t makes it visible one whole tile before its first read.  So a wave may read tile t + 1's first
    // fragments BEFORE the barrier that ends tile t (published during tile t - 1): the LDS latency after each barrier -- 8 waves
    // issuing their first reads at once, both waves of every SIMD waiting on them -- is off the tile's critical path.  The
    // publish (wait for the fetch, swizzled LDS writes: 440 cycles as a phase of its own between the last MFMA and the barrier,
    // stamped) rides in the gaps of the last k-steps, which carry no exponentials.  (Past the last fetched tile it re-writes stale
    // registers into a stage nobody reads again.)
    issue(0);
    publish(0);
    if (ntiles > 1) { issue(1); publish(1); }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): retire the fragment loads before the loop (see forward kernel)
    __syncthreads();
    bf16x8 fa[4][2];                                  // K rows | V rows of score step i in slot i & 3: a ring ACROSS tiles
    auto read_a = [&](const char* base, int i) {                    // step i = (block, k-step)
        fa[i & 3][0] = get_rows(base, 32 * (i >> 2) + l31, i & 3, half);
        fa[i & 3][1] = get_rows(base + TILE_B, 32 * (i >> 2) + l31, i & 3, half);
    };
    int cur = 0, nxt = 1, nn = 2;                     // stages of tiles t, t + 1, t + 2
    if (wave_live) { read_a(lds, 0); read_a(lds, 1); read_a(lds, 2); }
    auto tile = [&](int t, auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const bool more2 = t + 2 < ntiles;
        if (more2) issue(t + 2);
        if (wave_live) {
            const char* base = lds + cur * STAGE;
            const char* nbase = lds + nxt * STAGE;
            f32x16 s[2], e[2];                           // per 32-key block: s = lse - S', e = dP - D
            // Every LDS fragment is read THREE steps (six MFMAs) before the MFMAs that consume it, into a register ring (hipcc waits
            // for a fragment at its first use); sched_barrier fences keep the order.  Key block 0's scores first, block 1's next
            // with block 0's exponentials in their shadow, then the dQ MFMAs of block 0's k-steps with block 1's exponentials in
            // theirs.  (Measured, profiles/r03_attn_bwd_*.txt: this in-wave interleave and the ring depth are each worth < 1 % --
            // the two waves of a SIMD overlap each other's phases anyway; what the loop's time went to was found elsewhere: the
            // grid's XCD placement, the tail tokens' workgroups, the publish phase.)
            const char* kt = base + 2 * TILE_B;
            bf16x8 fk[4][2];
            uint32_t dw[2][8];
            const int key0 = t * BT + 4 * half;
            auto read2 = [&](int ks) { fk[ks][0] = get_rows(kt, l31, ks, half); fk[ks][1] = get_rows(kt, 32 + l31, ks, half); };
            static_for<0, 8>([&](auto ic) {
                constexpr int I = decltype(ic)::value, blk = I >> 2, ks = I & 3, h = I & 3;
                s[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][0], qf[ks], ks == 0 ? lse16 : s[blk], 0, 0, 0);
                if constexpr (I < 5) read_a(base, I + 3); else read2(I - 5);
                if constexpr (blk == 1) softmax_grad_unit<RAGGED, false, 2 * ks>(s[0], e[0], nullptr, dw[0], key0, p.L);
                DGS_SCHED_FENCE();
                e[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][1], dof[ks], ks == 0 ? negd16 : e[blk], 0, 0, 0);
                if constexpr (blk == 1) softmax_grad_unit<RAGGED, false, 2 * ks + 1>(s[0], e[0], nullptr, dw[0], key0, p.L);
                DGS_SCHED_FENCE();
            });
            static_for<0, 4>([&](auto kc) {
                constexpr int ks = decltype(kc)::value, h = ks;
                const bf16x8 dsf = words4(dw[ks >> 1] + 4 * (ks & 1));
                dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[h][0], dsf, dq0, 0, 0, 0);
                if constexpr (ks == 0) read2(3); else read_a(nbase, ks - 1);                // (past the last tile: unused reads)
                if constexpr (ks < 2) {
                    softmax_grad_unit<RAGGED, false, 4 * ks>(s[1], e[1], nullptr, dw[1], key0 + 32, p.L);
                    softmax_grad_unit<RAGGED, false, 4 * ks + 1>(s[1], e[1], nullptr, dw[1], key0 + 32, p.L);
                } else if constexpr (ks == 2) publish_piece(nn, 0);
                else publish_piece(nn, 2);
                DGS_SCHED_FENCE();
                dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[h][1], dsf, dq1, 0, 0, 0);
                if constexpr (ks < 2) {
                    softmax_grad_unit<RAGGED, false, 4 * ks + 2>(s[1], e[1], nullptr, dw[1], key0 + 32, p.L);
                    softmax_grad_unit<RAGGED, false, 4 * ks + 3>(s[1], e[1], nullptr, dw[1], key0 + 32, p.L);
                } else if constexpr (ks == 2) publish_piece(nn, 1);
                DGS_SCHED_FENCE();
            });
        } else publish(nn);
        __syncthreads();
        const int c = cur; cur = nxt; nxt = nn; nn = c;
    };
    const int nplain = p.L / BT;                          // tiles without keys >= L
    for (int t = 0; t < nplain; ++t) tile(t, BoolTag<false>{});
    if (nplain < ntiles) tile(nplain, BoolTag<true>{});
    if (wave_live) {
        bf16_t* orow = p.dq + (row0 + q) * p.ld_d + head * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float c = p.scale;
            *reinterpret_cast<uint2*>(orow + 8 * g + 4 * half) = make_uint2(pack_bf2(c * dq0[4 * g], c * dq0[4 * g + 1]), pack_bf2(c * dq0[4 * g + 2], c * dq0[4 * g + 3]));
            *reinterpret_cast<uint2*>(orow + 32 + 8 * g + 4 * half) = make_uint2(pack_bf2(c * dq1[4 * g], c * dq1[4 * g + 1]), pack_bf2(c * dq1[4 * g + 2], c * dq1[4 * g + 3]));
        }
    }
    if (p.dqT || p.bias_part)                                       // (uniform; the last tile's barrier is behind every wave: the ring is free)
        block_byproducts(dq0, dq1, p.scale, wave_live && q_raw < p.L,
                         p.dqT ? p.dqT + (size_t)b * p.t_batch_stride_qkv + (size_t)head * 64 * p.lpad + qblk * BQ : nullptr, p.lpad, p.lpad - qblk * BQ,
                         p.bias_part ? p.bias_part + ((size_t)b * (p.nmain + p.ntail) + qblk) * p.bias_stride + head * 64 : nullptr,
                         reinterpret_cast<float*>(lds), wave, l31, half);
}

// ------------------------------------------------------------------------------------------------------------------
// dK, dV.  Lane = key (column of every accumulator).  Per 64-query tile:
//   S  = Q . K^T           (A: Q rows from LDS,   B: K fragment registers)      rows of the accumulator = queries
//   dP = dO . V^T          (A: dO rows from LDS,  B: V fragment registers)
//   dV^T += dO^T . P       (A: dO^T permuted tile, B: packed P accumulator registers)
//   dK^T += Q^T . dS       (A: Q^T permuted tile,  B: packed dS accumulator registers)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void attention_bwd_dkv_kernel(AttnBwdParams p) {
    DGS_DYNAMIC_LDS(lds);                                                // [3 stages][Q | dO | Q^T | dO^T | lse[64] | -D[64]] : 97.5 KiB
    constexpr int STAGE = DKV_STAGE;
    int kblk, head, b;
    block_coords(p, kblk, head, b);
    if (kblk >= p.nmain) { tail_key_role(p, kblk, head, b, reinterpret_cast<float*>(lds)); return; }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const size_t row0 = (size_t)b * p.lpad;
    const int key_raw = kblk * BQ + wave * 32 + l31;
    const int key = key_raw < p.lpad ? key_raw : p.lpad - 1;
    const bool wave_live = kblk * BQ + wave * 32 < p.L;
    const bool key_ok = key_raw < p.L;
    const bf16_t* Qg = p.q + row0 * p.ld + head * 64;
    const bf16_t* dOg = p.dO + row0 * p.ld_o + head * 64;
    const bf16_t* QTg = p.qT + (size_t)b * p.t_batch_stride_qkv + (size_t)head * 64 * p.lpad;
    const bf16_t* dOTg = p.dOT + (size_t)b * p.t_batch_stride_do + (size_t)head * 64 * p.lpad;
    const float* lseg = p.lse2 + ((size_t)b * p.heads + head) * p.lpad;
    const float* Dg = p.D + ((size_t)b * p.heads + head) * p.lpad;

    bf16x8 kf[4], vf[4];
    {
        const bf16_t* kp = p.k + (row0 + key) * p.ld + head * 64;
        const bf16_t* vp = p.v + (row0 + key) * p.ld + head * 64;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = *reinterpret_cast<const bf16x8*>(kp + (2 * ks + half) * 8);
            vf[ks] = *reinterpret_cast<const bf16x8*>(vp + (2 * ks + half) * 8);
        }
    }
    f32x16 dk0 = zero_acc(), dk1 = zero_acc(), dv0 = zero_acc(), dv1 = zero_acc();
    const int ntiles = (p.L + BT - 1) / BT;            // queries >= L have dO = 0 and no valid lse: never visited / masked
    const int sr = tid >> 3, sc = tid & 7;
    uint4 qreg, doreg, qtreg, dotreg;
    float statreg = 0.f;                              // threads 0..63: lse of query t*64 + tid; 64..127: -D of query t*64 + tid - 64
    auto issue = [&](int t) {
        if (tid < 128) statreg = (tid < 64 ? lseg : Dg)[t * BT + (tid & 63)];
        qreg = *reinterpret_cast<const uint4*>(Qg + (size_t)(t * BT + sr) * p.ld + sc * 8);
        doreg = *reinterpret_cast<const uint4*>(dOg + (size_t)(t * BT + sr) * p.ld_o + sc * 8);
        qtreg = *reinterpret_cast<const uint4*>(QTg + (size_t)sr * p.lpad + t * BT + sc * 8);
        dotreg = *reinterpret_cast<const uint4*>(dOTg + (size_t)sr * p.lpad + t * BT + sc * 8);
    };
    auto publish_piece = [&](int stage, int piece) {
        char* base = lds + stage * STAGE;
        if (piece == 0) put_rows(base, sr, sc, kKnock == 4 ? qreg : prescaled(qreg, -p.scale_log2e));     // the S operand: bf16(-scale log2(e) q); Q^T stays raw (it feeds dK)
        if (piece == 1) put_rows(base + TILE_B, sr, sc, doreg);
        if (piece == 2) put_perm(base + 2 * TILE_B, sr, sc, qtreg);
        if (piece == 3) {
            put_perm(base + 3 * TILE_B, sr, sc, dotreg);
            if (tid < 128) reinterpret_cast<float*>(base + 4 * TILE_B)[tid] = statreg;
        }
    };
    auto publish = [&](int stage) { publish_piece(stage, 0); publish_piece(stage, 1); publish_piece(stage, 2); publish_piece(stage, 3); };
    // three stages, one barrier per tile, the next tile's first reads issued before it, the publish in the gaps of the last
    // k-steps (see the dQ kernel)
    issue(0);
    publish(0);
    if (ntiles > 1) { issue(1); publish(1); }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    f32x16 s[2], e[2];                                // per 32-query block: s = lse - S', e = dP - D
    bf16x8 fa[4][2];                                  // Q rows | dO rows of score step i in slot i & 3: a ring ACROSS tiles
    bf16x8 fsink[4][2];                               // (knock-out 7: the reads happen, the MFMAs do not depend on them)
    auto read_a = [&](const char* base, int i) {                    // step i = (block, k-step)
        if constexpr (kKnock == 6) return;
        if constexpr (kKnock == 7) {
            fsink[i & 3][0] = get_rows(base, 32 * (i >> 2) + l31, i & 3, half);
            fsink[i & 3][1] = get_rows(base + TILE_B, 32 * (i >> 2) + l31, i & 3, half);
            return;
        }
        fa[i & 3][0] = get_rows(base, 32 * (i >> 2) + l31, i & 3, half);
        fa[i & 3][1] = get_rows(base + TILE_B, 32 * (i >> 2) + l31, i & 3, half);
    };
    // per-query statistics, loaded INTO the accumulators' initial values: register r of block 0 / 1 is query 8 (r >> 2) + 4 half +
    // (r & 3) (+ 32) of the tile (staged with the tiles: an L2 round trip per tile otherwise)
    auto load_stats = [&](const char* base, int blk) {
        if constexpr (kKnock == 3) return;
        const float* lt = reinterpret_cast<const float*>(base + 4 * TILE_B) + 4 * half + 32 * blk;
        const float* dt = lt + 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 l = *reinterpret_cast<const float4*>(lt + 8 * g), d = *reinterpret_cast<const float4*>(dt + 8 * g);
            s[blk][4 * g] = l.x; s[blk][4 * g + 1] = l.y; s[blk][4 * g + 2] = l.z; s[blk][4 * g + 3] = l.w;
            e[blk][4 * g] = d.x; e[blk][4 * g + 1] = d.y; e[blk][4 * g + 2] = d.z; e[blk][4 * g + 3] = d.w;
        }
    };
    int cur = 0, nxt = 1, nn = 2;                     // stages of tiles t, t + 1, t + 2
    if (wave_live) { read_a(lds, 0); load_stats(lds, 0); read_a(lds, 1); read_a(lds, 2); }
    const bool stamped = kInstrumented && (p.dbg & 16) && blockIdx.x == 100 && lane == 0 && (wave == 0 || wave == 5);
    auto stamp = [&](int t, int i) {
        if constexpr (kInstrumented)
            if (stamped && t >= 20 && t < 28) dgs_attn_bwd_dbg[((wave ? 1 : 0) * 8 + (t - 20)) * 8 + i] = cycle_stamp();
    };
    auto tile = [&](int t, auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const bool more2 = t + 2 < ntiles;
        stamp(t, 0);
        if (more2) issue(t + 2);
        if (wave_live) {
            const char* base = lds + cur * STAGE;
            const char* nbase = lds + nxt * STAGE;
            // score fragments three steps ahead in a register ring, product fragments one k-step (four MFMAs) ahead, the
            // exponentials of one query block in the shadow of the other block's MFMAs (see the dQ kernel and softmax_grad_unit);
            // block 1's statistics behind the first MFMA, the next tile's block 0 statistics and first fragments behind the last
            // k-step of this one
            const char* qt = base + 2 * TILE_B;
            const char* dot = base + 3 * TILE_B;
            bf16x8 fg[2][4];
            uint32_t pw[2][8], dw[2][8];
            const int q0 = t * BT + 4 * half;
            auto read_g_half = [&](int ks, int h, int part) {
                if (part == 0) { fg[h][0] = get_rows(dot, l31, ks, half); fg[h][1] = get_rows(dot, 32 + l31, ks, half); }
                else { fg[h][2] = get_rows(qt, l31, ks, half); fg[h][3] = get_rows(qt, 32 + l31, ks, half); }
            };
            auto read_g = [&](int ks, int h) { read_g_half(ks, h, 0); read_g_half(ks, h, 1); };
            static_for<0, 8>([&](auto ic) {
                constexpr int I = decltype(ic)::value, blk = I >> 2, ks = I & 3, h = I & 3;
                s[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][0], kf[ks], s[blk], 0, 0, 0);
                if constexpr (I < 5) read_a(base, I + 3); else if constexpr (I < 7) read_g_half(0, 0, I - 5);
                if constexpr (I == 0) load_stats(base, 1);
                if constexpr (blk == 1) softmax_grad_unit<RAGGED, true, 2 * ks>(s[0], e[0], pw[0], dw[0], q0, p.L);
                DGS_SCHED_FENCE();
                e[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][1], vf[ks], e[blk], 0, 0, 0);
                if constexpr (blk == 1) softmax_grad_unit<RAGGED, true, 2 * ks + 1>(s[0], e[0], pw[0], dw[0], q0, p.L);
                DGS_SCHED_FENCE();
            });
            stamp(t, 1);
            static_for<0, 4>([&](auto kc) {
                constexpr int ks = decltype(kc)::value, h = ks & 1;
                const bf16x8 pf = words4(pw[ks >> 1] + 4 * (ks & 1)), dsf = words4(dw[ks >> 1] + 4 * (ks & 1));
                if constexpr (kKnock != 5) dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg[h][0], pf, dv0, 0, 0, 0);
                else { dv0[ks] += __builtin_bit_cast(float, pw[ks >> 1][4 * (ks & 1)]); dk0[ks] += __builtin_bit_cast(float, dw[ks >> 1][4 * (ks & 1)]); }
                if constexpr (ks < 3) read_g(ks + 1, h ^ 1); else read_a(nbase, 0);        // (past the last tile: unused reads)
                if constexpr (ks < 2) softmax_grad_unit<RAGGED, true, 4 * ks>(s[1], e[1], pw[1], dw[1], q0 + 32, p.L);
                DGS_SCHED_FENCE();
                if constexpr (kKnock != 5) dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg[h][1], pf, dv1, 0, 0, 0);
                if constexpr (ks < 2) softmax_grad_unit<RAGGED, true, 4 * ks + 1>(s[1], e[1], pw[1], dw[1], q0 + 32, p.L);
                if constexpr (ks == 2) publish_piece(nn, 0);
                if constexpr (ks == 3) load_stats(nbase, 0);
                DGS_SCHED_FENCE();
                if constexpr (kKnock != 5) dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg[h][2], dsf, dk0, 0, 0, 0);
                if constexpr (ks < 2) softmax_grad_unit<RAGGED, true, 4 * ks + 2>(s[1], e[1], pw[1], dw[1], q0 + 32, p.L);
                if constexpr (ks == 2) publish_piece(nn, 1);
                if constexpr (ks == 3) { read_a(nbase, 1); publish_piece(nn, 3); }
                DGS_SCHED_FENCE();
                if constexpr (kKnock != 5) dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fg[h][3], dsf, dk1, 0, 0, 0);
                if constexpr (ks < 2) softmax_grad_unit<RAGGED, true, 4 * ks + 3>(s[1], e[1], pw[1], dw[1], q0 + 32, p.L);
                if constexpr (ks == 2) publish_piece(nn, 2);
                if constexpr (ks == 3) read_a(nbase, 2);
                DGS_SCHED_FENCE();
            });
            if constexpr (kKnock == 7) {
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("" ::"v"(fsink[i][0]), "v"(fsink[i][1]));
            }
            stamp(t, 2);
        } else publish(nn);
        stamp(t, 3);
        __syncthreads();
        stamp(t, 4);
        const int c = cur; cur = nxt; nxt = nn; nn = c;
    };
    const int nplain = p.L / BT;                          // tiles without queries >= L
    for (int t = 0; t < nplain; ++t) tile(t, BoolTag<false>{});
    if (nplain < ntiles) tile(nplain, BoolTag<true>{});
    // lane = key: a key >= L only ever polluted its own dK / dV rows -- they are padding rows and receive exact zeros;
    // dK carries the `scale` the loop left out
    const float ck = p.scale;
    if (wave_live) {
        bf16_t* krow = p.dk + (row0 + key) * p.ld_d + head * 64;
        bf16_t* vrow = p.dv + (row0 + key) * p.ld_d + head * 64;
        const uint2 zero2 = make_uint2(0u, 0u);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 k0 = make_uint2(pack_bf2(ck * dk0[4 * g], ck * dk0[4 * g + 1]), pack_bf2(ck * dk0[4 * g + 2], ck * dk0[4 * g + 3]));
            const uint2 k1 = make_uint2(pack_bf2(ck * dk1[4 * g], ck * dk1[4 * g + 1]), pack_bf2(ck * dk1[4 * g + 2], ck * dk1[4 * g + 3]));
            const uint2 v0 = make_uint2(pack_bf2(dv0[4 * g], dv0[4 * g + 1]), pack_bf2(dv0[4 * g + 2], dv0[4 * g + 3]));
            const uint2 v1 = make_uint2(pack_bf2(dv1[4 * g], dv1[4 * g + 1]), pack_bf2(dv1[4 * g + 2], dv1[4 * g + 3]));
            *reinterpret_cast<uint2*>(krow + 8 * g + 4 * half) = key_ok ? k0 : zero2;
            *reinterpret_cast<uint2*>(krow + 32 + 8 * g + 4 * half) = key_ok ? k1 : zero2;
            *reinterpret_cast<uint2*>(vrow + 8 * g + 4 * half) = key_ok ? v0 : zero2;
            *reinterpret_cast<uint2*>(vrow + 32 + 8 * g + 4 * half) = key_ok ? v1 : zero2;
        }
    }
    if (p.dkT || p.bias_part) {                                     // (uniform; the ring is free behind the last tile's barrier)
        const size_t tcol = (size_t)b * p.t_batch_stride_qkv + (size_t)head * 64 * p.lpad + kblk * BQ;
        float* const brow = p.bias_part ? p.bias_part + ((size_t)b * (p.nmain + p.ntail) + kblk) * p.bias_stride + head * 64 : nullptr;
        const int W = p.heads * 64;
        block_byproducts(dk0, dk1, ck, wave_live && key_ok, p.dkT ? p.dkT + tcol : nullptr, p.lpad, p.lpad - kblk * BQ, brow ? brow + W : nullptr, reinterpret_cast<float*>(lds), wave, l31, half);
        block_byproducts(dv0, dv1, 1.f, wave_live && key_ok, p.dvT ? p.dvT + tcol : nullptr, p.lpad, p.lpad - kblk * BQ, brow ? brow + 2 * W : nullptr, reinterpret_cast<float*>(lds), wave, l31, half);
    }
}

}  // namespace dgs

using namespace dgs;

extern "C" int32_t dgs_dit_attention_backward_slots(int32_t L) {
    if (L <= 0) return 0;
    const int full = L / BQ, rest = L % BQ;
    const bool tail = full >= 1 && rest >= 1 && rest <= BWD_TAIL_MAX;
    return tail ? full + rest : (L + BQ - 1) / BQ;
}

extern "C" int dgs_dit_attention_backward(const DgsDitAttentionBackwardArgs* a, dgs_stream_t stream) {
    if (!a || a->B <= 0 || a->heads <= 0 || a->L <= 0 || a->lpad < a->L || a->lpad % 128) return DGS_ERR_INVALID_ARGUMENT;
    if (!a->qkv || !a->qkvT || !a->o || !a->dO || !a->dOT || !a->lse2 || !a->D || !a->dqkv) return DGS_ERR_INVALID_ARGUMENT;
    const int W = a->heads * 64;
    AttnBwdParams p;
    p.B = a->B; p.heads = a->heads; p.L = a->L; p.lpad = a->lpad; p.ld = 3 * W; p.ld_o = W; p.ld_d = 3 * W;
    p.t_batch_stride_qkv = (long long)3 * W * a->lpad; p.t_batch_stride_do = (long long)W * a->lpad;
    p.q = a->qkv; p.k = a->qkv + W; p.v = a->qkv + 2 * W;
    p.qT = a->qkvT; p.kT = a->qkvT + (size_t)W * a->lpad;
    p.o = a->o; p.dO = a->dO; p.dOT = a->dOT; p.lse2 = a->lse2; p.D = a->D;
    p.dq = a->dqkv; p.dk = a->dqkv + W; p.dv = a->dqkv + 2 * W;
    p.dqT = a->dqkvT; p.dkT = a->dqkvT ? a->dqkvT + (size_t)W * a->lpad : nullptr; p.dvT = a->dqkvT ? a->dqkvT + (size_t)2 * W * a->lpad : nullptr;
    p.bias_part = a->bias_part; p.bias_stride = 3 * W;
    p.scale = a->scale; p.scale_log2e = a->scale * 1.44269504088896341f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int full = a->L / BQ, rest = a->L % BQ;
    const bool tail = full >= 1 && rest >= 1 && rest <= BWD_TAIL_MAX;
    p.nmain = tail ? full : (a->L + BQ - 1) / BQ;
    p.ntail = tail ? rest : 0;
    const dim3 grid((p.nmain + p.ntail) * a->heads * a->B);
    constexpr int DQ_LDS = BWD_RING * DQ_STAGE, DKV_LDS = BWD_RING * DKV_STAGE;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dq_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DQ_LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS) != hipSuccess)
            return DGS_ERR_DEVICE;
        attr_set = true;
    }
    p.dbg = 0;
#ifdef DGS_INSTRUMENT
    static const int dbg = getenv("DGS_ATTN_DBG") ? atoi(getenv("DGS_ATTN_DBG")) : 0;
    p.dbg = dbg;
#endif
    hipLaunchKernelGGL(attention_bwd_dq_kernel, grid, dim3(512), DQ_LDS, st, p);
    hipLaunchKernelGGL(attention_bwd_dkv_kernel, grid, dim3(512), DKV_LDS, st, p);
#ifdef DGS_INSTRUMENT
    if (dbg & 16) {
        static long long host[2 * 8 * 8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(dgs_attn_bwd_dbg), sizeof(host));
        for (int w = 0; w < 2; ++w)
            for (int t = 0; t < 8; ++t) {
                const long long* h = host + (w * 8 + t) * 8;
                fprintf(stderr, "[attn bwd dbg] dkv wg 100 wave %d tile %d: scores %lld  products %lld  publish %lld  barrier %lld  | tile %lld cycles\n", w ? 5 : 0, 20 + t,
                        h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[4] - h[0]);
            }
    }
#endif
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
