// dit_attention.hip -- flash-style fused attention forward for the DiT blocks (gfx950, wave64, bf16 MFMA).
//
// Replaces F.scaled_dot_product_attention(q, k, v) as called by timm==0.9.16 Attention.forward (non-causal, no mask,
// scale = head_dim^-1/2, head_dim = 64), used by DiTBlock (utils_transformer.py:254-256, 286).  The S = L x L score
// matrix is never materialised: per 256-query workgroup (8 waves x 32 queries) the kernel walks 64-key tiles with an
// online softmax (fp32 running max / sum, fp32 accumulators).
//
// Formulation (cdna_hip_programming.md appendix B, "swapped QK^T"): every MFMA is computed transposed,
//     S^T[key, q] = K . Q^T            A = K fragment (LDS),  B = Q fragment (registers, loaded once)
//     O^T[d,  q]  = V^T . P^T          A = V^T fragment (LDS), B = P^T fragment (registers)
// so the query index is the lane (column of the 32x32 accumulator): softmax statistics and the O rescale are per-lane
// scalars, the row max needs ONE cross-lane exchange (lane ^ 32), and the P^T accumulator registers of a lane are --
// after bf16 packing -- directly the B fragment of the second MFMA (the k-slot <-> key mapping of an MFMA is free as long
// as both operands agree, so the V^T fragment is simply read with the accumulator's row pattern).
// V arrives already transposed ([B, heads*64, lpad]) from the QKV GEMM epilogue, so no transposing LDS access is needed.
//
// VALU diet (the loop is VALU-bound at head_dim 64: 32 scores per lane per tile against 16 MFMAs):
//   * Q is multiplied by scale * log2(e) once, when its fragments are loaded, and the S accumulators START at -m (the
//     running max, a per-lane scalar kept splatted in 16 registers as the MFMA's C operand) -- so the matrix pipe hands
//     back S' - m and the softmax is a bare v_exp_f32 per score: no per-score multiply / subtract;
//   * deferred rescale (T13): m only moves when a score outgrows it by more than 2^RESCALE_THR.
//
// LDS: K tile [64 keys][128 B] and V^T tile [64 d][128 B], both with the 16-byte slot swizzle s ^ ((row >> 1) & 7)
// (conflict-free ds_read_b128); inside a V^T row every 16-key group is stored as [k0-3, k8-11 | k4-7, k12-15] so the 8
// keys a lane contributes to one MFMA k-step are one 16-byte read.  K ring 3 deep, V^T ring 2 deep, register-staged
// prefetch of the next tiles issued at the top of an iteration and written behind its MFMAs (T14), one barrier per tile.
// Every operand fragment is read from LDS a whole phase before the MFMAs that consume it.
#include <stdio.h>
#include <stdlib.h>

#include "dit_common.h"
#include "dgs_dit.h"

namespace dgs {

constexpr int QB = 256;          // queries per workgroup: 8 waves x 32
constexpr int NW = QB / 32;
constexpr int KB = 64;           // keys per tile
constexpr int KV_TILE_BYTES = KB * 64 * 2;   // 8 KiB
constexpr float RESCALE_THR = 6.0f;          // deferred-max threshold in exp2 units: P <= 64

struct AttnParams {
    int B, heads, L, lpad, ld_qk, k_offset, nqb, nfull, nmain, dbg, q_prescaled;
    int tail_r;              // tail queries THIS launch handles inside the main kernel: L % 32, or 0 when they run as a launch of their own (tail_mode)
    long long vt_batch_stride;
    float* lse2;
    float* tail_ws;          // [B * heads][chunks][L % 32][TAIL_REC] partial results of the tail workgroups
    unsigned* tail_cnt;      // [B * heads] arrival counters (zero between launches)
    const bf16_t* qk;
    const bf16_t* vt;
    bf16_t* out;
    float scale_log2e;
};

__device__ __forceinline__ float row_max(const f32x16& s0, const f32x16& s1) {
    float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, fmaxf(s0[r], s1[r]));
    return xor32_max(mx);
}

template <int V> struct Mode { static constexpr int value = V; };


// The L % 32 queries behind the last full 32-query unit (the two learned tokens of the DiT: L = 32 k + 2) would cost a
// whole extra wave per head on the matrix pipe.  Their attention is split over the KEY tiles instead: behind the main loop
// of every workgroup of the head wave w takes key tile qblk + nqb * w (if it exists), computes exp2(S' - tile max), its sum and its P V for the tail queries with the same
// MFMA formulation (fragments straight from global memory, no LDS) and parks one record {max, sum, O[64]} per tail
// query in a library-owned workspace.  The last workgroup of the (sample, head) to finish merges the records.
//
// Cross-workgroup hand-off without fences: a release fence at agent scope writes back the whole L2 (~15 us behind 8 MB of
// attention output); the records are written and read with agent-scope atomics instead (sc1: performed at the memory
// side, coherent across XCDs), ordered by vmcnt(0) + barrier before the arrival counter is bumped.
union PFrag { bf16x8 v; uint32_t u[4]; };

constexpr int TAIL_REC = 68;          // floats per (key tile, query) record: max, sum, 2 unused, O[64] (O 16-byte aligned: 9 stores per lane)
// The tail tiles of a workgroup (one per wave: key tile qblk + nqb * wave) and the tail queries are copied into LDS by the main
// loop's last rounds (LDS-DMA, no registers), so that the tail work behind the loop starts from LDS instead of a round trip of
// fragment gathers at the moment all 256 workgroups leave their loops (DGS_ATTN_DBG stamps: 10.5 k cycles per workgroup, of 162 k).
constexpr int TAIL_LDS_TILES = 5;                         // tiles staged per workgroup (waves 0..4); a wave beyond that reads global memory
constexpr int TAIL_DMAS = 2 * TAIL_LDS_TILES + 1;         // per wave: its 1 KiB piece of every staged K and V^T tile + of the query tile

__device__ __forceinline__ int pi16(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// one key tile of the tail queries, by one wave; rec = this tile's records [r][TAIL_REC]
// `lds_k` != nullptr: the tile's K / V^T images and the tail queries' rows are in LDS (same layout as the ring tiles; foff = the
// lane's fragment offsets of the main loop); otherwise the fragments are gathered from global memory.
// REC_LDS: `rec` is in LDS (the workgroup merges its waves' records before anything leaves for memory: plain stores)
template <bool REC_LDS = false>
__device__ __forceinline__ void tail_tile(const AttnParams& p, int t, int r, const bf16_t* Qg, const bf16_t* Kg, const bf16_t* Vg, float* rec, int lane,
                                          const char* lds_k = nullptr, const char* lds_v = nullptr, const char* lds_q = nullptr, const int* foff = nullptr) {
    const int l31 = lane & 31, half = lane >> 5;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bf16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
    // fragments: Q (B operand) of the tail unit, K rows pi-permuted like the LDS image, V^T rows as they are
    const int qrow = p.nfull * 32 + l31, qld = qrow < p.lpad ? qrow : p.lpad - 1;
    bf16x8 qf[4], kf[8], vf[8];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 raw;
        if (lds_k) raw = *reinterpret_cast<const uint4*>(lds_q + foff[ks]);
        else raw = *reinterpret_cast<const uint4*>(Qg + (size_t)qld * p.ld_qk + (2 * ks + half) * 8);
        if (!p.q_prescaled) {
            raw.x = scale_bf2(raw.x, p.scale_log2e); raw.y = scale_bf2(raw.y, p.scale_log2e);
            raw.z = scale_bf2(raw.z, p.scale_log2e); raw.w = scale_bf2(raw.w, p.scale_log2e);
        }
        qf[ks] = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            if (lds_k) {
                kf[2 * ks + blk] = *reinterpret_cast<const bf16x8*>(lds_k + foff[ks] + blk * 32 * 128);
                vf[2 * ks + blk] = *reinterpret_cast<const bf16x8*>(lds_v + foff[ks] + blk * 32 * 128);
            } else {
                const int key = t * KB + pi16(l31 + 32 * blk);
                kf[2 * ks + blk] = *reinterpret_cast<const bf16x8*>(Kg + (size_t)(key < p.L ? key : p.L - 1) * p.ld_qk + (2 * ks + half) * 8);
                vf[2 * ks + blk] = *reinterpret_cast<const bf16x8*>(Vg + (size_t)(l31 + 32 * blk) * p.lpad + t * KB + 16 * ks + 8 * half);
            }
        }
    }
    f32x16 s0 = zero16, s1 = zero16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * ks], qf[ks], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * ks + 1], qf[ks], s1, 0, 0, 0);
    }
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {          // keys >= L (only the last tile has any)
        const int key = t * KB + (rr & 3) + 4 * ((rr >> 2) & 1) + 8 * half + 16 * (rr >> 3);
        if (key >= p.L) s0[rr] = -__builtin_inff();
        if (key + 32 >= p.L) s1[rr] = -__builtin_inff();
    }
    const float mx = row_max(s0, s1);          // finite: every tile holds a key < L
    PFrag pf[4];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) { s0[rr] = fast_exp2(s0[rr] - mx); s1[rr] = fast_exp2(s1[rr] - mx); }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = 8 * (ks & 1) + 2 * j;
            pf[ks].u[j] = ks < 2 ? pack_bf2(s0[e], s0[e + 1]) : pack_bf2(s1[e], s1[e + 1]);
        }
    f32x16 o0 = zero16, o1 = zero16, la = zero16;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2 * ks], pf[ks].v, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2 * ks + 1], pf[ks].v, o1, 0, 0, 0);
        la = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, pf[ks].v, la, 0, 0, 0);
    }
    if (l31 < r) {                             // lane owns query l31: d = db * 32 + 8 (rr >> 2) + 4 half + (rr & 3)
        float* q = rec + l31 * TAIL_REC;       // [max, sum, -, -, O[64]]: four consecutive d per accumulator group = one 16-byte store
        if constexpr (REC_LDS) {
            if (half == 0) { q[0] = mx; q[1] = la[0]; }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                *reinterpret_cast<float4*>(q + 4 + 8 * g + 4 * half) = make_float4(o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]);
                *reinterpret_cast<float4*>(q + 4 + 32 + 8 * g + 4 * half) = make_float4(o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]);
            }
        } else {
            if (half == 0) st_agent2(q, mx, la[0]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                st_agent4(q + 4 + 8 * g + 4 * half, o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]);
                st_agent4(q + 4 + 32 + 8 * g + 4 * half, o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]);
            }
        }
    }
}

// merge of the nrec per-tile records of one (sample, head), by the whole workgroup (512 threads)
// An item is one (tail query, d) pair: r * 64 of them.  With the DiT's r = 2 that is 128 items for 512 threads: as one thread per
// item walking all 65 records in two dependent LDS passes the merge took 15 k cycles of the LAST workgroup of every head
// (DGS_ATTN_DBG stamps) -- 8 us at the very end of an 87 us kernel.  The records of a batch are therefore cut into G = 512 / items
// slices (a power of two, at most 8); thread (slice g, item i) folds its slice into a running {max, sum, O} with the LDS reads of
// four records in flight, and the G partial states of an item meet in LDS once, at the end.
__device__ void tail_merge(const AttnParams& p, int bh, int nrec, int r, char* lds) {
    const int tid = threadIdx.x, head = bh % p.heads, b = bh / p.heads;
    // records of consecutive tiles are contiguous: pull them into LDS in batches (independent coalesced loads), then
    // every thread folds its slice of the batch into the running {max, sum, O} of its (query, d) item
    float* lbuf = reinterpret_cast<float*>(lds);
    const float* const recs = p.tail_ws + (size_t)bh * nrec * r * TAIL_REC;
    const int per_rec = r * TAIL_REC, nitems = r * 64;
    const int cb = 15360 / per_rec;                            // tiles per batch (60 KiB of LDS)
    int G = 1;
    while (2 * G * nitems <= 512 && G < 8) G *= 2;             // slices of a batch; G == 1: one thread per item, items looped
    float Mr[4], lr[4], orr[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) { Mr[it] = -3.0e38f; lr[it] = 0.0f; orr[it] = 0.0f; }
    for (int c0 = 0; c0 < nrec; c0 += cb) {
        const int n = nrec - c0 < cb ? nrec - c0 : cb;
        __syncthreads();
        const float2* src = reinterpret_cast<const float2*>(recs + (size_t)c0 * per_rec);       // 68 floats per record
        const int n2 = n * per_rec / 2;
        for (int i0 = 0; i0 < n2; i0 += 10 * 512) {             // 10 independent loads in flight per thread
            float2 v[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int i = i0 + u * 512 + tid;
                v[u] = i < n2 ? ld_agent2(src + i) : make_float2(0.0f, 0.0f);
            }
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int i = i0 + u * 512 + tid;
                if (i < n2) reinterpret_cast<float2*>(lbuf)[i] = v[u];
            }
        }
        __syncthreads();
        const int per_slice = (n + G - 1) / G;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 512, g = idx / nitems, i = idx - g * nitems;
            if (g >= G) break;
            const int ca = g * per_slice, ce = ca + per_slice < n ? ca + per_slice : n;
            if (ca >= ce) continue;
            const float* rec = lbuf + (i >> 6) * TAIL_REC;
            float M = Mr[it];
#pragma unroll 4
            for (int c = ca; c < ce; ++c) M = fmaxf(M, rec[c * per_rec]);
            const float w0 = fast_exp2(Mr[it] - M);
            float l = lr[it] * w0, o = orr[it] * w0;
#pragma unroll 4
            for (int c = ca; c < ce; ++c) {
                const float w = fast_exp2(rec[c * per_rec] - M);
                l += w * rec[c * per_rec + 1];
                o += w * rec[c * per_rec + 4 + (i & 63)];
            }
            Mr[it] = M; lr[it] = l; orr[it] = o;
        }
    }
    if (G > 1) {                                               // the G partial states of an item: LDS, then slice 0 folds them in slice order
        __syncthreads();
        const int g = tid / nitems, i = tid - g * nitems;
        float* part = lbuf;                                    // [G][nitems][3]
        if (g < G) { part[(g * nitems + i) * 3] = Mr[0]; part[(g * nitems + i) * 3 + 1] = lr[0]; part[(g * nitems + i) * 3 + 2] = orr[0]; }
        __syncthreads();
        if (g != 0) return;
        float M = Mr[0];
        for (int k = 1; k < G; ++k) M = fmaxf(M, part[(k * nitems + i) * 3]);
        float l = 0.0f, o = 0.0f;
        for (int k = 0; k < G; ++k) {
            const float w = fast_exp2(part[(k * nitems + i) * 3] - M);          // an empty slice: exp2(-3e38 - M) = 0
            l += w * part[(k * nitems + i) * 3 + 1];
            o += w * part[(k * nitems + i) * 3 + 2];
        }
        Mr[0] = M; lr[0] = l; orr[0] = o;
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int i = tid + it * 512;
        if (i >= r * 64) break;
        const int qq = i >> 6, d = i & 63, qrow = p.nfull * 32 + qq;
        p.out[((size_t)b * p.lpad + qrow) * (size_t)(p.heads * 64) + head * 64 + d] = (bf16_t)f2bf_fast(orr[it] / lr[it]);
        if (d == 0 && p.lse2) p.lse2[((size_t)b * p.heads + head) * p.lpad + qrow] = Mr[it] + log2f(lr[it]);
    }
}

// Workgroup = 256 queries of one (sample, head): 8 waves x 32 queries, two waves per SIMD, one workgroup per CU.  At
// L = 4098 that is 16 heads x 16 blocks = 256 workgroups = exactly one round of the 256 CUs (the L % 32 = 2
// learned-token queries go to the tail workgroups above).  The grid is 1-D with head = id % heads: all query blocks of a
// head run on one XCD (dispatch places block b on XCD b % 8) and re-read that head's K / V^T (1 MiB) from its L2.
//
// Staging is LDS-DMA (global_load_lds, 16 B per lane, no registers): K tile rows are stored PERMUTED -- LDS row R holds
// key pi(R) = R with bits 2 and 3 swapped -- so that the 8 accumulator rows a lane owns inside a 16-row group are 8
// CONSECUTIVE keys, and the matching V^T fragment is one plain 16-byte read of the token-contiguous V^T row; both tiles
// use the 16-byte slot swizzle slot ^ ((row >> 1) & 7), applied on the DMA's source address and on the ds_read address.
// Rings are 4 tiles deep; the DMA for K(t+4) / V(t+3) is issued at the top of iteration t and only has to have landed
// by the end of iteration t+1 (counted vmcnt, raw s_barrier): more than a full iteration of flight time.
//
// Issue schedule (tools/ubench/issue_bench: on one SIMD a v_mfma_f32_32x32x16_bf16 occupies the matrix pipe for 32
// cycles and hides ~6 plain VALU issued behind it; v_exp_f32 costs two issue slots, packed-f32 VALU stalls beside MFMAs).
// The loop is VALU-issue bound at head_dim 64, so the goal is that no MFMA ever waits and no VALU slot is wasted: one
// iteration is cut into 16 SLICES of { 1 MFMA, 1 LDS fragment read, 6 VALU }, fenced with sched_barrier so the order
// below is the issue order.  Slices 0-7 carry S'(t+1) = K(t+1) (cQ)^T - m, slices 8-15 carry O^T += V^T(t) P^T(t); the
// VALU list is the softmax of tile t (80 ops: exp2, row-sum add, bf16 pack; pf[g] is complete before the P V slice that
// consumes it) followed by the row max of S'(t+1) (17 ops).
constexpr int RING = 4;
__device__ long long dgs_attn_dbg[4096 + 64];   // DGS_ATTN_DBG & 4: loop cycles (s_memtime) per wave of the first 512 workgroups; & 8: phases of wg 0
__device__ unsigned dgs_attn_tl[512][8];   // DGS_ATTN_DBG & 16: per-workgroup phase stamps (100 MHz clock, low 32 bits)



// flat VALU op list of the softmax of one tile: group g = K / 12 produces pf[g] from 8 scores as (exp2 + row-sum add, exp2 +
// add, pack) x 4.  The row sums l = sum_k P[q][k] are two running fp32 adds per lane (one per key block; the lane pair (l, l ^ 32)
// holds the two halves of a query's keys and is combined ONCE, in the epilogue): a ones fragment beside V^T used to form them on the
// matrix pipe (4 of a tile's 20 MFMAs and 20 registers); the loop is matrix-pipe-bound since the VALU diet, so they went back.
template <int K> __device__ __forceinline__ void softmax_step(f32x16& s0, f32x16& s1, PFrag (&pf)[4], float& l0, float& l1) {
    constexpr int g = K / 12, o = K % 12, j = o / 3, base = 8 * (g & 1);
    f32x16& s = g < 2 ? s0 : s1;
    float& l = g < 2 ? l0 : l1;
    if constexpr (o % 3 < 2) {
        s[base + 2 * j + o % 3] = fast_exp2(s[base + 2 * j + o % 3]);
        l += s[base + 2 * j + o % 3];
    } else pf[g].u[j] = pack_bf2(s[base + 2 * j], s[base + 2 * j + 1]);
}
// VALU steps per slice (cumulative), balanced by issue cycles (tools/ubench/issue_bench: v_exp_f32 8, v_add 4, v_cvt_pk / v_max3 5,
// LDS read ~5, next to the MFMA's 32): softmax steps 0-47 in slices 0-12 (fewer in the slices that also issue LDS reads: 0-1, 4-5,
// 9-10; pf[ks] is complete 2+ slices before P V slice 8 + 2 ks), the row max of the next tile (steps 48-63) in slices 12-15.
__device__ constexpr int SLICE_END[16] = {3, 6, 11, 16, 19, 22, 27, 32, 37, 39, 41, 46, 51, 56, 60, 64};
// row max of the next tile's scores, two per step (one v_max3_f32 each; hipcc fuses only every other pair on its own)
template <int K> __device__ __forceinline__ void max_step(const f32x16& n0, const f32x16& n1, float& mx) {
    const f32x16& n = K < 8 ? n0 : n1;
    constexpr int e = 2 * (K & 7);
    if constexpr (K == 0) mx = fmaxf(n[e], n[e + 1]);
    else mx = __builtin_fmaxf(__builtin_fmaxf(mx, n[e]), n[e + 1]);
}

__global__ __launch_bounds__(512) void attention_fwd_kernel(AttnParams p) {
    DGS_DYNAMIC_LDS(lds);                              // K ring [4] | V^T ring [4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int dbg = kInstrumented ? p.dbg : 0;         // debug switches exist in the instrumented library only (dit_common.h)
    // DGS_ATTN_DBG & 16: every workgroup's {start, loop end, tail tile + fold done, output stores issued, arrived, end} on the constant
    // 100 MHz clock (dgs_attn_tl: the launch's timeline, like the GEMMs' profiles/r05_gemm_timeline.txt)
#define MAIN_STAMP(i) do { if ((dbg & 8) && blockIdx.x == 0 && tid == 0) dgs_attn_dbg[4096 + 16 + (i)] = cycle_stamp(); \
                           if ((dbg & 16) && tid == 0 && blockIdx.x < 512) dgs_attn_tl[blockIdx.x][(i)] = (unsigned)wall_stamp(); } while (0)
    MAIN_STAMP(0);
    // ---- block -> (sample, head, query block) ----
    int id = blockIdx.x;
    const int head = id % p.heads; id /= p.heads;
    const int qblk = id % p.nqb, b = id / p.nqb;
    const int unit = qblk * NW + wave;                 // this wave's 32-query unit
    const bool wave_live = unit < p.nfull;             // other waves only help with staging and barriers
    const size_t row0 = (size_t)b * p.lpad;
    const bf16_t* Qg = p.qk + row0 * p.ld_qk + head * 64;
    const bf16_t* Kg = Qg + p.k_offset;
    const bf16_t* Vg = p.vt + (size_t)b * p.vt_batch_stride + (size_t)head * 64 * p.lpad;

    // Q fragments (B operand), pre-multiplied by scale * log2(e): query = lane & 31, d-chunk = 2 ks + half.
    const int q = unit * 32 + l31;
    const int qld = q < p.lpad ? q : p.lpad - 1;
    bf16x8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        uint4 raw = *reinterpret_cast<const uint4*>(Qg + (size_t)qld * p.ld_qk + (2 * ks + half) * 8);
        if (!p.q_prescaled) {
            raw.x = scale_bf2(raw.x, p.scale_log2e); raw.y = scale_bf2(raw.y, p.scale_log2e);
            raw.z = scale_bf2(raw.z, p.scale_log2e); raw.w = scale_bf2(raw.w, p.scale_log2e);
        }
        qf[ks] = __builtin_bit_cast(bf16x8, raw);
    }

    f32x16 oacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.0f;
    float lsum0 = 0.0f, lsum1 = 0.0f;                   // row sums of this lane's half of the keys, per key block (softmax_step)
    float m_run = 0.0f;                                   // log2 units (scores are pre-scaled)

    const int ntiles = (p.L + KB - 1) / KB;
    const int mask_from = p.L / KB;                       // first tile that contains keys >= L (== ntiles if none)
    // ---- LDS-DMA staging: a tile is 8 pieces of 1 KiB (8 rows); wave w moves piece w of K and of V^T ----
    char* const kring = lds;
    char* const vring = lds + RING * KV_TILE_BYTES;
    const int R = 8 * wave + (lane >> 3);                          // LDS row this lane fills
    const int chunk = (lane & 7) ^ ((R >> 1) & 7);                 // 16-byte source chunk that belongs in LDS slot lane & 7
    const bf16_t* const ksrc = Kg + (size_t)pi16(R) * p.ld_qk + chunk * 8;
    const bf16_t* const vsrc = Vg + (size_t)R * p.lpad + chunk * 8;
    auto stage_k = [&](int tile, int slot) { glds16(ksrc + (size_t)tile * KB * p.ld_qk, kring + slot * KV_TILE_BYTES + wave * 1024); };
    auto stage_v = [&](int tile, int slot) { glds16(vsrc + (size_t)tile * KB, vring + slot * KV_TILE_BYTES + wave * 1024); };
    // tail staging (see TAIL_LDS_TILES): [K j | V^T j] x TAIL_LDS_TILES behind the rings, then the tail queries' rows as one more
    // K-shaped tile (LDS row R = query nfull * 32 + R, unpermuted: the B operand wants query = lane & 31).  Every wave issues
    // exactly TAIL_DMAS pieces (a tile index beyond the last tile is clamped: a harmless duplicate), so the counted waits of the
    // iterations around the issue are compile-time.
    const int tail_r = p.tail_r;
    char* const tail_lds = lds + 2 * RING * KV_TILE_BYTES;
    char* const tail_q = tail_lds + 2 * TAIL_LDS_TILES * KV_TILE_BYTES;
    auto stage_tail = [&]() {
#pragma unroll
        for (int j = 0; j < TAIL_LDS_TILES; ++j) {
            const int tt = min(qblk + p.nqb * j, ntiles - 1);
            glds16(ksrc + (size_t)tt * KB * p.ld_qk, tail_lds + (2 * j) * KV_TILE_BYTES + wave * 1024);
            glds16(vsrc + (size_t)tt * KB, tail_lds + (2 * j + 1) * KV_TILE_BYTES + wave * 1024);
        }
        const int qrow = min(p.nfull * 32 + R, p.lpad - 1);
        glds16(Qg + (size_t)qrow * p.ld_qk + chunk * 8, tail_q + wave * 1024);
    };
    int tail_age = tail_r ? -1 : 99;       // -1: still to be staged; 0 / 1: its pieces may be in flight across this iteration's wait; >= 2: landed
    // per-lane fragment address inside a tile: fragment i = 2 ks + blk -> row l31 + 32 blk, 16-byte slot (2 ks + half) ^ swizzle
    const int kswz = (l31 >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) foff[ks] = l31 * 128 + (((2 * ks + half) ^ kswz) << 4);
    auto frag = [&](const char* tile, int i) { return *reinterpret_cast<const bf16x8*>(tile + foff[i >> 1] + (i & 1) * 32 * 128); };
    // key (inside the tile) of accumulator register r of key block kb: LDS row 32 kb + (r & 3) + 8 (r >> 2) + 4 half, un-permuted
    auto mask_tile = [&](f32x16& s0, f32x16& s1, int key0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * half + 16 * (r >> 3);
            if (key >= p.L) s0[r] = -__builtin_inff();
            if (key + 32 >= p.L) s1[r] = -__builtin_inff();
        }
    };

    // ---- prologue: K(0..3), V(0..2) in flight; wait for all of it (and the Q fragments) ----
#pragma unroll
    for (int j = 0; j < RING; ++j)
        if (j < ntiles) stage_k(j, j);
#pragma unroll
    for (int j = 0; j < RING - 1; ++j)
        if (j < ntiles) stage_v(j, j);
    wait_vmcnt<0>();
    __syncthreads();
    MAIN_STAMP(1);
    // ---- first tile: S'(0), its row max = the initial running max; fragments of K(1) ----
    f32x16 s0, s1, negm;
    bf16x8 kf[8];
    if (wave_live) {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[i] = frag(kring, i);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * ks], qf[ks], ks == 0 ? zero16 : s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[2 * ks + 1], qf[ks], ks == 0 ? zero16 : s1, 0, 0, 0);
        }
        if (mask_from == 0) mask_tile(s0, s1, 0);
        m_run = row_max(s0, s1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { s0[r] -= m_run; s1[r] -= m_run; negm[r] = -m_run; }
#pragma unroll
        for (int i = 0; i < 4; ++i) kf[i] = frag(kring + KV_TILE_BYTES, i);     // K(1), first half; iteration 0 reads the rest
    }
    __syncthreads();   // K(0)'s slot is refilled with K(4) at the top of iteration 0: every wave must be done reading it

    // One iteration = one 64-key tile t.  MODE 0: steady state, 1: S(t+1) is the last tile (ragged: masked), 2: last tile
    // (no S(t+1)).  Straight-line copies instead of in-loop branches.  `cs*` hold P(t) (in: S'(t) - m), `ns*` receive
    // S'(t+1) - m: the caller alternates two register sets, so nothing is copied between iterations.
    // `fast_tag` = 1: the caller guarantees t + RING < ntiles and dbg == 0 -- both DMAs go out unconditionally, the wait is the
    // literal vmcnt(2) and the barrier is unconditional; `live_tag` = 0 / 1 then replaces the run-time test of `wave_live` (2: test
    // it).  The steady-state loop below runs on these copies: tools/ubench/issue_bench prices ONE scalar compare + branch in the
    // MFMA + VALU stream at 16 (untaken) / 27 (taken) cycles of the wave and an s_add at 2.75, and the generic form spends ~10
    // branches and ~30 other SALU per tile on conditions that never change inside the loop.
    auto iteration = [&](int t, const int slot, auto mode_tag, f32x16& cs0, f32x16& cs1, f32x16& ns0, f32x16& ns1, auto fast_tag, auto live_tag) {
        constexpr int MODE = decltype(mode_tag)::value;     // slot == t % RING; a literal at the steady-state call sites
        constexpr bool FAST = decltype(fast_tag)::value != 0;
        constexpr int LIVE = decltype(live_tag)::value;
        static_assert(!FAST || MODE == 0, "the fast form is the steady state");
        int issued = 0;
        if constexpr (FAST) {
            stage_k(t + RING, slot);
            stage_v(t + RING - 1, (slot + RING - 1) & (RING - 1));
        } else if (MODE == 0 && !(dbg & 1)) {
            if (t + RING < ntiles) { stage_k(t + RING, slot); ++issued; }
            if (t + RING - 1 < ntiles) { stage_v(t + RING - 1, (slot + RING - 1) & (RING - 1)); ++issued; }
        }
        // the tail pieces go out BEHIND this iteration's own DMAs, in the first iteration of the generic rounds at the end of the loop
        // (K / V^T are L2-resident by then); they may stay in flight across this wait and the next one -- two iterations, like a ring
        // tile -- because memory operations retire in order: "at most TAIL_DMAS + own outstanding" still means everything older landed
        if constexpr (!FAST && MODE != 2) {
            if (tail_age < 0) { stage_tail(); tail_age = 0; }
        }
        if (LIVE == 1 || (LIVE == 2 && wave_live)) {
            bf16x8 vf[8];
            PFrag pf[4];
            float mx = 0.0f;
            const char* const vtile = vring + slot * KV_TILE_BYTES;
            const char* const ktile1 = kring + ((slot + 1) & (RING - 1)) * KV_TILE_BYTES;      // K(t+1)
            const char* const ktile2 = kring + ((slot + 2) & (RING - 1)) * KV_TILE_BYTES;      // K(t+2)
            static_for<0, 16>([&](auto jc) {
                constexpr int J = decltype(jc)::value;
                // ---- matrix pipe: slices 0-7 S'(t+1); slices 8-15 per k-step { O^T block 0, O^T block 1 } ----
                if constexpr (J < 8) {
                    if constexpr (MODE != 2) {
                        constexpr int ks = J >> 1;
                        if constexpr ((J & 1) == 0) ns0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[J], qf[ks], ks == 0 ? negm : ns0, 0, 0, 0);
                        else ns1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[J], qf[ks], ks == 0 ? negm : ns1, 0, 0, 0);
                    }
                } else {
                    constexpr int ks = (J - 8) / 2, w = (J - 8) % 2;
                    oacc[w] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[2 * ks + w], pf[ks].v, oacc[w], 0, 0, 0);
                }
                // ---- LDS: three bursts -- slices 0-1: K(t+1) fragments 4-7 (used from slice 4); 4-5: V^T(t) fragments 0-3 (from
                //      slice 8); 9-10: V^T(t) fragments 4-7 (from slice 14) and K(t+2) fragments 0-3 (next iteration).  hipcc waits
                //      with lgkmcnt(0) at the first use of a burst, so no read may be in flight shortly before such a point. ----
                if constexpr (J < 2 && MODE != 2) { kf[4 + 2 * J] = frag(ktile1, 4 + 2 * J); kf[5 + 2 * J] = frag(ktile1, 5 + 2 * J); }
                if constexpr (J == 4 || J == 5) { vf[2 * J - 8] = frag(vtile, 2 * J - 8); vf[2 * J - 7] = frag(vtile, 2 * J - 7); }
                if constexpr (J == 9 || J == 10) {
                    vf[2 * J - 14] = frag(vtile, 2 * J - 14); vf[2 * J - 13] = frag(vtile, 2 * J - 13);
                    if constexpr (MODE == 0) { kf[2 * J - 18] = frag(ktile2, 2 * J - 18); kf[2 * J - 17] = frag(ktile2, 2 * J - 17); }
                }
                // ---- VALU: [softmax(t) x 48 | row max of S'(t+1) x 16]; pf[ks] is complete 2+ slices before its P V ----
                constexpr int K0 = J == 0 ? 0 : SLICE_END[J == 0 ? 0 : J - 1], K1 = SLICE_END[J];
                static_for<K0, K1>([&](auto kc) {
                    constexpr int K = decltype(kc)::value;
                    if constexpr (K < 48) softmax_step<K>(cs0, cs1, pf, lsum0, lsum1);
                    else if constexpr (MODE != 2) {
                        if constexpr (MODE == 1 && K == 48) mask_tile(ns0, ns1, (t + 1) * KB);
                        max_step<K - 48>(ns0, ns1, mx);
                    }
                });
                sched_fence();
            });
            if constexpr (MODE != 2) {
                mx = xor32_max(mx);                     // relative to the running max
                // deferred rescale (T13) for tile t+1 -- after ALL of P(t) V(t) has been issued: the running max only moves
                // when some query of the wave outgrew it by more than RESCALE_THR (exp2 units), so P <= 2^THR and the
                // O / l rescale is skipped on almost every tile
                if (__any(mx > RESCALE_THR)) {
                    const float d = fmaxf(mx, 0.0f);
                    const float alpha = fast_exp2(-d);
                    m_run += d;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        oacc[0][r] *= alpha; oacc[1][r] *= alpha;
                        ns0[r] -= d; ns1[r] -= d; negm[r] -= d;
                    }
                    lsum0 *= alpha; lsum1 *= alpha;
                }
            }
        }
        if (MODE != 2) {
            // everything issued BEFORE this iteration (K(t+3), V^T(t+2) and older) has landed once at most this
            // iteration's own DMAs are outstanding; then everybody is also done reading K(t+2)'s and V^T(t)'s slots
            if constexpr (FAST) {
                wait_vmcnt<2>();
                raw_barrier();
            } else {
                if (tail_age == 0 || tail_age == 1) {
                    if (issued == 2) wait_vmcnt<2 + TAIL_DMAS>();
                    else if (issued == 1) wait_vmcnt<1 + TAIL_DMAS>();
                    else wait_vmcnt<TAIL_DMAS>();
                    ++tail_age;
                } else if (issued == 2) wait_vmcnt<2>();
                else wait_vmcnt<0>();
                if (!(dbg & 2)) raw_barrier();
            }
        }
    };
    const bool ragged = mask_from < ntiles && ntiles > 1;       // the last tile holds keys >= L
    const int steady_end = ragged ? ntiles - 2 : ntiles - 1;    // iterations [0, steady_end) use MODE 0
    MAIN_STAMP(2);
    f32x16 u0, u1;                                              // second score register set
    int t = 0;
    const long long dbg_t0 = (dbg & 4) ? cycle_stamp() : 0;
    // steady state, unrolled by the ring depth: ring slots are literals, every LDS address is register + immediate.  While all
    // four iterations of a round still issue both of their DMAs (t + 3 + RING < ntiles) and no debug switch is set, the round runs
    // on the branch-free copies -- one loop for the waves with queries, one for the waves that only stage and synchronise.
    constexpr IC<0> GEN{};
    constexpr IC<1> YES{};
    constexpr IC<2> ASK{};
    if (dbg == 0) {
        if (wave_live) {
            for (; t + 3 + RING < ntiles; t += 4) {
                iteration(t, 0, Mode<0>{}, s0, s1, u0, u1, YES, YES);
                iteration(t + 1, 1, Mode<0>{}, u0, u1, s0, s1, YES, YES);
                iteration(t + 2, 2, Mode<0>{}, s0, s1, u0, u1, YES, YES);
                iteration(t + 3, 3, Mode<0>{}, u0, u1, s0, s1, YES, YES);
            }
        } else {
            for (; t + 3 + RING < ntiles; t += 4) {
                iteration(t, 0, Mode<0>{}, s0, s1, u0, u1, YES, GEN);
                iteration(t + 1, 1, Mode<0>{}, u0, u1, s0, s1, YES, GEN);
                iteration(t + 2, 2, Mode<0>{}, s0, s1, u0, u1, YES, GEN);
                iteration(t + 3, 3, Mode<0>{}, u0, u1, s0, s1, YES, GEN);
            }
        }
    }
    for (; t + 3 < steady_end; t += 4) {
        iteration(t, 0, Mode<0>{}, s0, s1, u0, u1, GEN, ASK);
        iteration(t + 1, 1, Mode<0>{}, u0, u1, s0, s1, GEN, ASK);
        iteration(t + 2, 2, Mode<0>{}, s0, s1, u0, u1, GEN, ASK);
        iteration(t + 3, 3, Mode<0>{}, u0, u1, s0, s1, GEN, ASK);
    }
    if (t + 1 < steady_end) {
        iteration(t, t & 3, Mode<0>{}, s0, s1, u0, u1, GEN, ASK);
        iteration(t + 1, (t + 1) & 3, Mode<0>{}, u0, u1, s0, s1, GEN, ASK);
        t += 2;
    }
    const int tl = ntiles - 1;
    if (t < steady_end) {                                       // odd count: the current scores end up in u
        iteration(t, t & 3, Mode<0>{}, s0, s1, u0, u1, GEN, ASK);
        if (ragged) { iteration(tl - 1, (tl - 1) & 3, Mode<1>{}, u0, u1, s0, s1, GEN, ASK); iteration(tl, tl & 3, Mode<2>{}, s0, s1, u0, u1, GEN, ASK); }
        else iteration(tl, tl & 3, Mode<2>{}, u0, u1, s0, s1, GEN, ASK);
    } else {
        if (ragged) { iteration(tl - 1, (tl - 1) & 3, Mode<1>{}, s0, s1, u0, u1, GEN, ASK); iteration(tl, tl & 3, Mode<2>{}, u0, u1, s0, s1, GEN, ASK); }
        else iteration(tl, tl & 3, Mode<2>{}, s0, s1, u0, u1, GEN, ASK);
    }

    if ((dbg & 4) && lane == 0 && blockIdx.x < 512) dgs_attn_dbg[blockIdx.x * 8 + wave] = cycle_stamp() - dbg_t0;
    MAIN_STAMP(3);
    // ---- finish: O[q, d] = O^T / l, packed to bf16 now (frees the accumulators), stored behind the tail tile ----
    uint2 outv[8];
    float lse_out = 0.0f;
    if (wave_live) {
        const float l_tot = xor32_sum(lsum0 + lsum1);       // the lane pair (l, l ^ 32) holds the two halves of the query's keys
        const float inv = 1.0f / l_tot;
        lse_out = m_run + log2f(l_tot);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                outv[4 * db + g].x = pack_bf2(oacc[db][4 * g] * inv, oacc[db][4 * g + 1] * inv);
                outv[4 * db + g].y = pack_bf2(oacc[db][4 * g + 2] * inv, oacc[db][4 * g + 3] * inv);
            }
    }
    // ---- tail queries: wave w takes key tile qblk + nqb * w (K / V^T are L2-resident by now) and leaves its record {max, sum, O[64]} per
    //      tail query in LDS (the ring is idle); the first r * 64 threads fold the workgroup's records into ONE (round 5: the last
    //      workgroup of the head then merges nqb records instead of one per key tile -- 16 instead of 65 at L = 4,098: its record loads,
    //      folds and combine were 9.7 k cycles at the very end of the kernel, profiles/r03_attn_tail_merge_stamps.txt).  The record's
    //      stores are issued BEFORE the output stores so that the counted wait below covers exactly them. ----
    const int bh = b * p.heads + head;
    if (tail_r) {
        if (tail_age < 0) stage_tail();                 // a single-tile sequence: no iteration with a wait came by
        wait_vmcnt<0>();                                // every wave's pieces of the staged tiles have landed ...
        raw_barrier();                                  // ... and are visible to the wave that reads them
        float* const wrec = reinterpret_cast<float*>(lds) + (size_t)wave * tail_r * TAIL_REC;     // this wave's records: [r][TAIL_REC] in the idle ring
        const int tt = qblk + p.nqb * wave;             // ntiles <= nqb * NW: at most one tile per wave
        if (tt < ntiles) {
            if (wave < TAIL_LDS_TILES) tail_tile<true>(p, tt, tail_r, Qg, Kg, Vg, wrec, lane, tail_lds + (2 * wave) * KV_TILE_BYTES, tail_lds + (2 * wave + 1) * KV_TILE_BYTES, tail_q, foff);
            else tail_tile<true>(p, tt, tail_r, Qg, Kg, Vg, wrec, lane);
        } else if (lane < tail_r) {                     // no tile: an empty record (weight exp2(-3e38 - M) = 0 in the fold)
            wrec[lane * TAIL_REC] = -3.0e38f; wrec[lane * TAIL_REC + 1] = 0.0f;
        }
        lds_barrier();
        for (int item = tid; item < tail_r * 64; item += 64 * NW) {       // item = (tail query, d): fold the NW records in wave order
            const int qq = item >> 6, d = item & 63;
            const float* rw = reinterpret_cast<const float*>(lds) + qq * TAIL_REC;
            float M = rw[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, rw[w * tail_r * TAIL_REC]);
            float l = 0.0f, o = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float* rr = rw + w * tail_r * TAIL_REC;
                const float wgt = fast_exp2(rr[0] - M);
                l += wgt * rr[1];
                o += (rr[1] > 0.0f ? wgt * rr[4 + d] : 0.0f);          // an empty record's O is not initialised
            }
            float* const rec = p.tail_ws + ((size_t)bh * p.nqb + qblk) * tail_r * TAIL_REC + qq * TAIL_REC;
            st_agent(rec + 4 + d, o);
            if (d == 0) st_agent2(rec, M, l);
        }
    }
    MAIN_STAMP(7);
    if (wave_live) {                                    // lane owns query q, d = db*32 + 8 g + 4 half + (0..3)
        if (p.lse2 && half == 0) p.lse2[((size_t)b * p.heads + head) * p.lpad + q] = lse_out;
        bf16_t* orow = p.out + (row0 + q) * (size_t)(p.heads * 64) + head * 64;
        // a lane pair (l31, l31 + 32) holds 8 consecutive d of two neighbouring 8-column groups: one exchange per dword makes
        // that 16 contiguous bytes per lane -- four 16-byte stores instead of eight 8-byte ones (the store tail is issue-bound)
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            half_swap(outv[i].x, outv[i + 1].x);
            half_swap(outv[i].y, outv[i + 1].y);
            *reinterpret_cast<uint4*>(orow + (i >> 2) * 32 + 8 * ((i & 3) + half)) = make_uint4(outv[i].x, outv[i].y, outv[i + 1].x, outv[i + 1].y);
        }
    }
    MAIN_STAMP(4);
    if (!tail_r) return;
    // the last workgroup of this (sample, head) to get here merges the per-tile records
    // stores retire in order: with at most the 4 output stores outstanding, every record store of this wave has been
    // performed (sc1: at the memory side).  A wave without live queries issued no output stores behind its records,
    // so it has to drain completely.
    if (wave_live) wait_vmcnt<4>();
    else wait_vmcnt<0>();
    int* flag = reinterpret_cast<int*>(lds);
    raw_barrier();                                     // raw barriers: __syncthreads() would also drain the output stores
    if (tid == 0) {
        const unsigned prev = atomicAdd(p.tail_cnt + bh, 1u);
        *flag = prev + 1 == (unsigned)p.nqb;
        if (*flag) p.tail_cnt[bh] = 0;                 // leave the counter ready for the next launch
    }
    wait_lgkmcnt0();
    raw_barrier();
    MAIN_STAMP(5);
    if (*flag) tail_merge(p, bh, p.nqb, tail_r, lds);        // one record per workgroup of the head
    MAIN_STAMP(6);
}

}  // namespace dgs

using namespace dgs;

static size_t tail_counter_bytes(int B, int heads) { return ((size_t)B * heads * sizeof(unsigned) + 255) / 256 * 256; }

extern "C" size_t dgs_dit_attention_tail_bytes(int32_t B, int32_t heads, int32_t L) {
    const int r = L % 32;
    if (B <= 0 || heads <= 0 || L <= 0 || !r) return 0;
    const size_t ntiles = (size_t)(L + KB - 1) / KB;
    return tail_counter_bytes(B, heads) + (size_t)B * heads * ntiles * r * TAIL_REC * sizeof(float);
}

extern "C" int dgs_dit_attention(const DgsDitAttentionArgs* a, dgs_stream_t stream) {
    if (!a || a->B <= 0 || a->heads <= 0 || a->L <= 0 || a->lpad < a->L || a->lpad % 128 || !a->qk || !a->vt || !a->out)
        return DGS_ERR_INVALID_ARGUMENT;
    AttnParams p;
    p.B = a->B; p.heads = a->heads; p.L = a->L; p.lpad = a->lpad;
    p.ld_qk = a->ld_qk > 0 ? a->ld_qk : 2 * a->heads * 64;
    p.k_offset = a->k_offset > 0 ? a->k_offset : a->heads * 64;
    p.vt_batch_stride = a->vt_batch_stride > 0 ? a->vt_batch_stride : (long long)a->heads * 64 * a->lpad;
    p.lse2 = a->lse2;
    if (a->tail_mode != 0) return DGS_ERR_INVALID_ARGUMENT;        // reserved (dgs_dit.h): the tail queries run inside the main kernel
    p.nfull = a->L / 32;                       // full 32-query wave units; the L % 32 rest goes to the tail workgroups
    p.nqb = p.nfull ? (p.nfull + NW - 1) / NW : 1;        // L < 32: one workgroup per head, tail path only
    p.nmain = a->B * a->heads * p.nqb;
    static const int dbg = kInstrumented && getenv("DGS_ATTN_DBG") ? atoi(getenv("DGS_ATTN_DBG")) : 0;     // instrumented library only
    p.dbg = dbg;
    const int r = a->L % 32;                           // tail queries: split over the key tiles behind the main loop
    p.tail_r = r;
    p.tail_ws = nullptr; p.tail_cnt = nullptr;
    if (r) {
        // caller-owned scratch (per call site / per stream): [counters | records]; the counters are zero between launches
        const size_t need = dgs_dit_attention_tail_bytes(a->B, a->heads, a->L);
        if (!a->tail_ws || a->tail_ws_bytes < need) return DGS_ERR_INVALID_ARGUMENT;
        p.tail_cnt = static_cast<unsigned*>(a->tail_ws);
        p.tail_ws = reinterpret_cast<float*>(static_cast<char*>(a->tail_ws) + tail_counter_bytes(a->B, a->heads));
    }
    p.qk = a->qk; p.vt = a->vt; p.out = a->out; p.q_prescaled = a->q_prescaled;
    p.scale_log2e = a->scale * 1.44269504088896341f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // 64 KiB of rings; DGS_ATTN_LDS_PAD (bytes) adds unused LDS to cap the workgroups per CU (measurement aid)
    static const int lds_pad = getenv("DGS_ATTN_LDS_PAD") ? atoi(getenv("DGS_ATTN_LDS_PAD")) : 0;
    // + the staged tail tiles and the tail queries' tile when L % 32 != 0: 64 + 80 + 8 = 152 KiB (one workgroup per CU either way)
    const int lds_bytes = 2 * RING * KV_TILE_BYTES + (r ? (2 * TAIL_LDS_TILES + 1) * KV_TILE_BYTES : 0) + lds_pad;
    static int lds_attr = 0;
    if (lds_attr != lds_bytes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(attention_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess)
            return DGS_ERR_DEVICE;
        lds_attr = lds_bytes;
    }
    hipLaunchKernelGGL(attention_fwd_kernel, dim3(p.nmain), dim3(512), lds_bytes, st, p);
#ifdef DGS_INSTRUMENT
    if (dbg & 16) {
        static unsigned tl[512][8];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(tl, HIP_SYMBOL(dgs_attn_tl), sizeof(tl));
        const int n = p.nmain < 512 ? p.nmain : 512;
        unsigned t0 = 0xffffffffu;
        for (int i = 0; i < n; ++i) t0 = tl[i][0] < t0 ? tl[i][0] : t0;
        // stamps: 0 start, 3 loop end, 7 tail tile + fold, 4 output stores issued, 5 arrived (records drained, counter), 6 end (merge if last)
        const int order[6] = {0, 3, 7, 4, 5, 6};
        const char* names[6] = {"start", "loop end", "tail tile + fold", "stores issued", "arrived", "end"};
        fprintf(stderr, "[attn timeline] L=%d, %d workgroups, us since the first start (min/mean/max):", a->L, n);
        for (int k = 0; k < 6; ++k) {
            double mn = 1e9, mx = 0, sum = 0;
            for (int i = 0; i < n; ++i) { const double v = (tl[i][order[k]] - t0) / 100.0; mn = v < mn ? v : mn; mx = v > mx ? v : mx; sum += v; }
            fprintf(stderr, "  %s %.2f/%.2f/%.2f", names[k], mn, sum / n, mx);
        }
        fprintf(stderr, "\n");
    }
    if (dbg & 4) {
        static long long host[4096 + 64];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(host, HIP_SYMBOL(dgs_attn_dbg), sizeof(host));
        const int n = (p.nmain < 512 ? p.nmain : 512) * 8;
        long long mn = host[0], mx = host[0]; double sum = 0;
        for (int i = 0; i < n; ++i) { mn = host[i] < mn ? host[i] : mn; mx = host[i] > mx ? host[i] : mx; sum += host[i]; }
        const int ntiles = (a->L + KB - 1) / KB;
        if (dbg & 8) fprintf(stderr, "[attn dbg] main wg0 stamps: prologue (load wait) %lld, first tile %lld, loop %lld, pack + tail tile %lld, output stores issued %lld, "
                             "record drain + barrier + counter %lld, merge (if last) %lld\n", host[4113] - host[4112],
                             host[4114] - host[4113], host[4115] - host[4114], host[4119] - host[4115], host[4116] - host[4119], host[4117] - host[4116], host[4118] - host[4117]);
        fprintf(stderr, "[attn dbg] L=%d loop cycles per wave: min %lld avg %.0f max %lld  -> per tile %.0f / %.0f / %.0f\n", a->L, mn, sum / n, mx,
                (double)mn / ntiles, sum / n / ntiles, (double)mx / ntiles);
    }
#endif
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
