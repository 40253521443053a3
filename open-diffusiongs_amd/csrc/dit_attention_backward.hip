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
#include "dgs_dit.h"

namespace dgs {

constexpr int BQ = 256, BNW = 8, BT = 64;      // rows per workgroup, waves, rows per tile of the walked dimension
constexpr int TILE_B = BT * 64 * 2;            // 8 KiB: one [64][64] bf16 tile

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
};

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
template <bool V> struct BoolTag { static constexpr bool value = V; };
#define DGS_SCHED_FENCE() sched_fence()

__device__ __forceinline__ f32x16 zero_acc() {
    const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    return z;
}

// ------------------------------------------------------------------------------------------------------------------
// dQ (and D).  Lane = query (column of every accumulator).  Per 64-key tile:
//   S^T  = K . Q^T         (A: K rows from LDS,  B: Q fragment registers)
//   dP^T = V . dO^T        (A: V rows from LDS,  B: dO fragment registers)
//   dQ^T += K^T . dS^T     (A: K^T permuted tile, B: packed dS^T accumulator registers)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void attention_bwd_dq_kernel(AttnBwdParams p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 3 * TILE_B];     // [stage][K | V | K^T]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int qblk = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
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
    auto publish = [&](int stage) {
        char* base = lds + stage * 3 * TILE_B;
        put_rows(base, sr, sc, kreg);
        put_rows(base + TILE_B, sr, sc, vreg);
        put_perm(base + 2 * TILE_B, sr, sc, ktreg);
    };
    issue(0);
    publish(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): retire the fragment loads before the loop (see forward kernel)
    __syncthreads();
    auto tile = [&](int t, auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const bool more = t + 1 < ntiles;
        if (more) issue(t + 1);
        if (wave_live) {
            const char* base = lds + (t & 1) * 3 * TILE_B;
            f32x16 s0, s1, e0, e1;                       // s = lse - S', e = dP - D
            // Every LDS fragment is read into the other half of a register double buffer behind the first MFMA of the step
            // before the one that consumes it (hipcc waits lgkmcnt(0) at first use: a read issued right before its MFMA
            // exposes the whole LDS latency, 24 times per tile); sched_barrier fences keep this order.
            const char* kt = base + 2 * TILE_B;
            bf16x8 fa[2][4], fk[2][2];
            auto read4 = [&](int ks, int h) {
                fa[h][0] = get_rows(base, l31, ks, half); fa[h][1] = get_rows(base, 32 + l31, ks, half);
                fa[h][2] = get_rows(base + TILE_B, l31, ks, half); fa[h][3] = get_rows(base + TILE_B, 32 + l31, ks, half);
            };
            auto read2 = [&](int ks, int h) { fk[h][0] = get_rows(kt, l31, ks, half); fk[h][1] = get_rows(kt, 32 + l31, ks, half); };
            read4(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int h = ks & 1;
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][0], qf[ks], ks == 0 ? lse16 : s0, 0, 0, 0);
                if (ks < 3) read4(ks + 1, h ^ 1); else read2(0, 0);
                DGS_SCHED_FENCE();
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][1], qf[ks], ks == 0 ? lse16 : s1, 0, 0, 0);
                DGS_SCHED_FENCE();
                e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][2], dof[ks], ks == 0 ? negd16 : e0, 0, 0, 0);
                DGS_SCHED_FENCE();
                e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][3], dof[ks], ks == 0 ? negd16 : e1, 0, 0, 0);
                DGS_SCHED_FENCE();
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p0 = fast_exp2(-s0[r]), p1 = fast_exp2(-s1[r]);
                if (RAGGED) {
                    const int key = t * BT + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (key >= p.L) { p0 = 0.f; e0[r] = 0.f; }
                    if (key + 32 >= p.L) { p1 = 0.f; e1[r] = 0.f; }
                }
                s0[r] = p0 * e0[r];                      // dS / scale
                s1[r] = p1 * e1[r];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int h = ks & 1;
                const bf16x8 dsf = pack_rows(ks < 2 ? s0 : s1, 8 * (ks & 1));
                dq0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[h][0], dsf, dq0, 0, 0, 0);
                if (ks < 3) read2(ks + 1, h ^ 1);
                DGS_SCHED_FENCE();
                dq1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk[h][1], dsf, dq1, 0, 0, 0);
                DGS_SCHED_FENCE();
            }
        }
        if (more) publish((t + 1) & 1);
        __syncthreads();
    };
    const int nplain = p.L / BT;                          // tiles without keys >= L
    for (int t = 0; t < nplain; ++t) tile(t, BoolTag<false>{});
    if (nplain < ntiles) tile(nplain, BoolTag<true>{});
    if (!wave_live) return;
    bf16_t* orow = p.dq + (row0 + q) * p.ld_d + head * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float c = p.scale;
        *reinterpret_cast<uint2*>(orow + 8 * g + 4 * half) = make_uint2(pack_bf2(c * dq0[4 * g], c * dq0[4 * g + 1]), pack_bf2(c * dq0[4 * g + 2], c * dq0[4 * g + 3]));
        *reinterpret_cast<uint2*>(orow + 32 + 8 * g + 4 * half) = make_uint2(pack_bf2(c * dq1[4 * g], c * dq1[4 * g + 1]), pack_bf2(c * dq1[4 * g + 2], c * dq1[4 * g + 3]));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dK, dV.  Lane = key (column of every accumulator).  Per 64-query tile:
//   S  = Q . K^T           (A: Q rows from LDS,   B: K fragment registers)      rows of the accumulator = queries
//   dP = dO . V^T          (A: dO rows from LDS,  B: V fragment registers)
//   dV^T += dO^T . P       (A: dO^T permuted tile, B: packed P accumulator registers)
//   dK^T += Q^T . dS       (A: Q^T permuted tile,  B: packed dS accumulator registers)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void attention_bwd_dkv_kernel(AttnBwdParams p) {
    DGS_DYNAMIC_LDS(lds);                                                // [stage][Q | dO | Q^T | dO^T | lse[64] | -D[64]] : 65 KiB
    constexpr int STAGE = 4 * TILE_B + 512;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int kblk = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
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
    auto publish = [&](int stage) {
        char* base = lds + stage * STAGE;
        uint4 qs = qreg;                                  // the S operand: bf16(-scale log2(e) q); Q^T below stays raw (it feeds dK)
        qs.x = scale_bf2(qs.x, -p.scale_log2e); qs.y = scale_bf2(qs.y, -p.scale_log2e);
        qs.z = scale_bf2(qs.z, -p.scale_log2e); qs.w = scale_bf2(qs.w, -p.scale_log2e);
        put_rows(base, sr, sc, qs);
        put_rows(base + TILE_B, sr, sc, doreg);
        put_perm(base + 2 * TILE_B, sr, sc, qtreg);
        put_perm(base + 3 * TILE_B, sr, sc, dotreg);
        if (tid < 128) reinterpret_cast<float*>(base + 4 * TILE_B)[tid] = statreg;
    };
    issue(0);
    publish(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    auto tile = [&](int t, auto ragged_tag) {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        const bool more = t + 1 < ntiles;
        if (more) issue(t + 1);
        if (wave_live) {
            const char* base = lds + (t & 1) * STAGE;
            // per-query statistics, loaded INTO the accumulators' initial values: register r of block 0 / 1 is query 8 (r >> 2) + 4 half + (r & 3) (+ 32) of the tile
            const float* lt = reinterpret_cast<const float*>(base + 4 * TILE_B) + 4 * half;     // staged with the tiles: an L2
            const float* dt = lt + 64;                                                          // round trip per tile otherwise
            f32x16 s0, s1, e0, e1;                       // s = lse - S', e = dP - D
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 l0 = *reinterpret_cast<const float4*>(lt + 8 * g), l1 = *reinterpret_cast<const float4*>(lt + 32 + 8 * g);
                const float4 d0 = *reinterpret_cast<const float4*>(dt + 8 * g), d1 = *reinterpret_cast<const float4*>(dt + 32 + 8 * g);
                s0[4 * g] = l0.x; s0[4 * g + 1] = l0.y; s0[4 * g + 2] = l0.z; s0[4 * g + 3] = l0.w;
                s1[4 * g] = l1.x; s1[4 * g + 1] = l1.y; s1[4 * g + 2] = l1.z; s1[4 * g + 3] = l1.w;
                e0[4 * g] = d0.x; e0[4 * g + 1] = d0.y; e0[4 * g + 2] = d0.z; e0[4 * g + 3] = d0.w;
                e1[4 * g] = d1.x; e1[4 * g + 1] = d1.y; e1[4 * g + 2] = d1.z; e1[4 * g + 3] = d1.w;
            }
            // fragments one step ahead in a register double buffer (see the dQ kernel)
            const char* qt = base + 2 * TILE_B;
            const char* dot = base + 3 * TILE_B;
            bf16x8 fa[2][4];
            auto read_s = [&](int ks, int h) {
                fa[h][0] = get_rows(base, l31, ks, half); fa[h][1] = get_rows(base, 32 + l31, ks, half);
                fa[h][2] = get_rows(base + TILE_B, l31, ks, half); fa[h][3] = get_rows(base + TILE_B, 32 + l31, ks, half);
            };
            auto read_g = [&](int ks, int h) {
                fa[h][0] = get_rows(dot, l31, ks, half); fa[h][1] = get_rows(dot, 32 + l31, ks, half);
                fa[h][2] = get_rows(qt, l31, ks, half); fa[h][3] = get_rows(qt, 32 + l31, ks, half);
            };
            read_s(0, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int h = ks & 1;
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][0], kf[ks], s0, 0, 0, 0);
                if (ks < 3) read_s(ks + 1, h ^ 1); else read_g(0, 0);
                DGS_SCHED_FENCE();
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][1], kf[ks], s1, 0, 0, 0);
                DGS_SCHED_FENCE();
                e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][2], vf[ks], e0, 0, 0, 0);
                DGS_SCHED_FENCE();
                e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][3], vf[ks], e1, 0, 0, 0);
                DGS_SCHED_FENCE();
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p0 = fast_exp2(-s0[r]), p1 = fast_exp2(-s1[r]);
                if (RAGGED) {                            // queries >= L: lse / D of padding rows are not meaningful
                    const int qq = t * BT + 8 * (r >> 2) + 4 * half + (r & 3);
                    if (qq >= p.L) { p0 = 0.f; e0[r] = 0.f; }
                    if (qq + 32 >= p.L) { p1 = 0.f; e1[r] = 0.f; }
                }
                e0[r] = p0 * e0[r]; e1[r] = p1 * e1[r];   // dS / scale
                s0[r] = p0; s1[r] = p1;                    // P
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int h = ks & 1;
                const bf16x8 pf = pack_rows(ks < 2 ? s0 : s1, 8 * (ks & 1));
                const bf16x8 dsf = pack_rows(ks < 2 ? e0 : e1, 8 * (ks & 1));
                dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][0], pf, dv0, 0, 0, 0);
                if (ks < 3) read_g(ks + 1, h ^ 1);
                DGS_SCHED_FENCE();
                dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][1], pf, dv1, 0, 0, 0);
                DGS_SCHED_FENCE();
                dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][2], dsf, dk0, 0, 0, 0);
                DGS_SCHED_FENCE();
                dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[h][3], dsf, dk1, 0, 0, 0);
                DGS_SCHED_FENCE();
            }
        }
        if (more) publish((t + 1) & 1);
        __syncthreads();
    };
    const int nplain = p.L / BT;                          // tiles without queries >= L
    for (int t = 0; t < nplain; ++t) tile(t, BoolTag<false>{});
    if (nplain < ntiles) tile(nplain, BoolTag<true>{});
    if (!wave_live) return;
    // lane = key: a key >= L only ever polluted its own dK / dV rows -- they are padding rows and receive exact zeros;
    // dK carries the `scale` the loop left out
    const float ck = p.scale;
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

}  // namespace dgs

using namespace dgs;

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
    p.scale = a->scale; p.scale_log2e = a->scale * 1.44269504088896341f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((a->L + BQ - 1) / BQ, a->heads, a->B);
    hipLaunchKernelGGL(attention_bwd_dq_kernel, grid, dim3(512), 0, st, p);
    constexpr int DKV_LDS = 2 * (4 * TILE_B + 512);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bwd_dkv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DKV_LDS) != hipSuccess)
            return DGS_ERR_DEVICE;
        attr_set = true;
    }
    hipLaunchKernelGGL(attention_bwd_dkv_kernel, grid, dim3(512), DKV_LDS, st, p);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
