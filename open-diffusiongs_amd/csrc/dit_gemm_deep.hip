// dit_gemm_deep.hip -- the "sliced" DiT GEMM: 256-row tiles on a 4-stage LDS-DMA ring with an explicit issue schedule (same
// contract and epilogues as dit_gemm.hip), its split-K form for the weight gradients, and the side jobs for tile rows with a
// single live 32-row block (the DiT's two learned-token rows).
//
// Why a second kernel: the 128-wide two-stage kernel of dit_gemm.hip drains its one slab in flight (`__syncthreads()` =
// vmcnt(0) + barrier) every K slab and stages 0.5-0.75 KiB per MFMA; it runs at 0.4-0.65 PFLOP/s.  This one keeps three slabs
// in flight per workgroup (COUNTED `s_waitcnt vmcnt(N)` + ONE raw `s_barrier` per slab), gives a wave 64 x 128 (or 128 x 128)
// of a 256 x 256 tile so that one staged KiB feeds 4 MFMAs, and fixes the issue order by hand.
// History (measured on MI355X, removed because they lost to the kernels kept): a 128 x N/8-tile deep-ring kernel, a two-stage
// 256 x 256 kernel, and split-K with the reduction inside the kernel for the few-tile / long-K shapes (fc2 at one sample:
// 66 -> 86 us, the partial tiles do not stay in L2).
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

#include "dit_common.h"
#include "dgs_dit.h"
#include "dit_gemm_epilogue.h"

namespace dgs {

struct DeepParams {
    int M, N, K, lda, ldw, ldo, gate_stride, rows_per_batch, valid_rows, tiles_n, ntiles, dbg;
    int rows_ps, full_rows, tail_rows, nfull_items;             // sliced kernel: 256-row tile rows per sample (all / ring path / one live block), full items
    int ntail, tail_mode;                                       // sliced kernel: side jobs of the single-live-block tile rows (1: MFMA items, 2: two-row GEMV items)
    int tail_wgs;                                               // > 0: that many workgroups BEHIND the tiles do nothing but the side jobs (idle CUs)
    int map2d;                                                  // tile <- workgroup id: 4 x 2 blocks of the tile grid per XCD (launch_sliced decides)
    int rows_external;                                          // the two-row GEMV side jobs are somebody else's work (layernorm_rows_gemv_kernel): none here
    int nsplit, splits_per_batch;                               // sliced kernel, split-K: items = nsplit x tiles, K = k_per_batch
    long long a_batch_stride, w_batch_stride, out_split_stride;
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    void* out;
    const float* gate;
    const float* resid;
    bf16_t* vt;
    void* aux;
    float q_scale;
};

__device__ __forceinline__ float gelu_tanh_d(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x / (1.0f + __expf(-2.0f * u));
}
__device__ __forceinline__ float dgelu_tanh_d(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float sg = 1.0f / (1.0f + __expf(-2.0f * u));
    return sg + x * sg * (1.0f - sg) * (2.0f * 0.7978845608028654f) * (1.0f + 3.0f * 0.044715f * x * x);
}

// ---- LDS image of a [rows][BK] bf16 slab: row-major, 16-byte chunk c of row r in slot
//        BK = 64: c ^ ((r >> 1) & 7)   (128-byte rows)        BK = 32: c ^ ((r >> 2) & 3)   (64-byte rows)
//      both conflict-free for ds_read_b128 when a 16-lane group reads 16 rows that are distinct mod 16.
template <int BK>
__device__ __forceinline__ int slab_off(int r, int c) {
    return BK == 64 ? r * 128 + ((c ^ ((r >> 1) & 7)) << 4) : r * 64 + ((c ^ ((r >> 2) & 3)) << 4);
}

template <int EPI, int NI>
__device__ __forceinline__ void store_block(const DeepParams& p, const f32x16 (&acc)[NI], int mbase, int nbase, int lane) {
    // D fragment: col (n) = lane & 31, row (m) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5); mbase already includes 4 * (lane >> 5)
    const int b = mbase / p.rows_per_batch;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = nbase + ni * 32 + (lane & 31);
        const float bias = p.bias ? p.bias[n] : 0.0f;
        const float qs = (EPI == DGS_EPI_QKV && n < p.N / 3) ? p.q_scale : 1.0f;      // pre-scaled queries
        const bool qkv_v = EPI == DGS_EPI_QKV && n >= (p.N / 3) * 2;
        bf16_t* tdst = nullptr;
        if (EPI == DGS_EPI_QKV) {
            if (qkv_v) tdst = p.vt + ((size_t)b * (p.N / 3) + (n - (p.N / 3) * 2)) * p.rows_per_batch + (mbase - b * p.rows_per_batch);
        } else if ((EPI == DGS_EPI_BF16 || EPI == DGS_EPI_GELU_BF16 || EPI == DGS_EPI_DGELU_BF16) && p.vt) {
            tdst = p.vt + ((size_t)b * p.N + n) * p.rows_per_batch + (mbase - b * p.rows_per_batch);
        }
        float gate = 0.0f;
        if (EPI == DGS_EPI_GATE_RESIDUAL) gate = p.gate[(size_t)b * p.gate_stride + n];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float o4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int m = mbase + q + 8 * g;
                const float v = EPI == DGS_EPI_QKV ? (acc[ni][4 * g + q] + bias) * qs : acc[ni][4 * g + q] + bias;
                const size_t o = (size_t)m * p.ldo + n;
                if (EPI == DGS_EPI_BF16) {
                    o4[q] = v;
                    reinterpret_cast<bf16_t*>(p.out)[o] = (bf16_t)f2bf_fast(v);
                } else if (EPI == DGS_EPI_QKV) {
                    o4[q] = v;
                    if (!qkv_v) reinterpret_cast<bf16_t*>(p.out)[o] = (bf16_t)f2bf_fast(v);
                } else if (EPI == DGS_EPI_GELU_BF16) {
                    o4[q] = gelu_tanh_d(v);
                    reinterpret_cast<bf16_t*>(p.out)[o] = (bf16_t)f2bf_fast(o4[q]);
                    if (p.aux) reinterpret_cast<bf16_t*>(p.aux)[o] = (bf16_t)f2bf_fast(v);
                } else if (EPI == DGS_EPI_DGELU_BF16) {
                    o4[q] = v * dgelu_tanh_d(bf2f(reinterpret_cast<const bf16_t*>(p.aux)[o]));
                    reinterpret_cast<bf16_t*>(p.out)[o] = (bf16_t)f2bf_fast(o4[q]);
                } else if (EPI == DGS_EPI_GATE_RESIDUAL) {
                    reinterpret_cast<float*>(p.out)[o] = p.resid[o] + gate * v;
                    if (p.aux) reinterpret_cast<bf16_t*>(p.aux)[o] = (bf16_t)f2bf_fast(v);
                } else {
                    reinterpret_cast<float*>(p.out)[o] = v;
                }
            }
            if (tdst) *reinterpret_cast<uint2*>(tdst + 8 * g) = make_uint2(pack_bf2(o4[0], o4[1]), pack_bf2(o4[2], o4[3]));
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// "Sliced" kernel: 256 x BN x 32 tiles (BN = 256 or 128), 8 waves as 4 (M) x 2 (N) -- a wave owns 64 x BN/2 --, 4-stage
// LDS-DMA ring, counted vmcnt + one raw barrier per K slab, and an explicit issue schedule.
// tools/ubench/issue_bench (MI355X): one SIMD retires { v_mfma_f32_32x32x16_bf16 + 1 ds_read_b128 } at 32-33 cycles and
// still ~36 with an LDS-DMA every 4th MFMA, but ONLY if no MFMA has to wait for an operand: hipcc waits with lgkmcnt(0)
// at the first use of a fragment, so a ds_read issued just before an MFMA exposes the whole LDS latency.  Here every
// fragment is read one k-substep (>= 4 MFMAs) before it is used, into the other half of a register double buffer:
//   slab t, substep 0:  MFMAs on fa/fb[0]   while reading fa/fb[1] <- slab t,     k 16..31
//   slab t, substep 1:  MFMAs on fa/fb[1]   while reading fa/fb[0] <- slab t + 1, k 0..15   (landed one barrier ago)
// with sched_barrier fences between { 1 MFMA, <= 1 LDS read } slices so the order in the source is the issue order.
// The slab loop is unrolled by the ring depth: ring slots are literals and every LDS address is register + immediate.
// ------------------------------------------------------------------------------------------------------------------
template <int I> struct SIC { static constexpr int value = I; };
template <int I, int N, class F> __device__ __forceinline__ void sliced_for(F&& f) {
    if constexpr (I < N) { f(SIC<I>{}); sliced_for<I + 1, N>(f); }
}

__device__ long long dgs_gemm_dbg[16];   // DGS_GEMM_DBG: cycle stamps of workgroup 0, wave 0 (loop total, wait + barrier share)
__device__ unsigned dgs_gemm_tl[1024][4]; // DGS_GEMM_DBG: per workgroup {start, loop start, loop end, end} on the constant 100 MHz clock (low 32 bits)

// NW = 4 ("quad"): the same ring and schedule with 4 waves as 2 x 2, a wave owning 128 x 128 (4 x 4 accumulators = 256
// registers, which hipcc keeps in AGPRs: MFMA reads and writes them there, nothing else touches them before the epilogue).
// Why: LDS bandwidth, not L2, bounds these kernels (tools/ubench/l2_tile_bench: the L2 -> LDS DMA alone streams GEMM tiles at
// 30 TB/s = 60 B/clk/CU).  Per 32 x 32 x 16 MFMA (32 cycles) a wave with an I x J block tile reads (I + J) / (I J) KiB of
// fragments; four SIMDs at full MFMA rate therefore pull 128 (I + J) / (I J) B/clk from a 128 B/clk LDS that also absorbs the
// DMA writes:   2 x 1 (the 128 x 64 tile): 192 + 48,   2 x 2: 128 + 32,   2 x 4 (NW = 8): 96 + 32,   4 x 4 (NW = 4): 64 + 32.
// EXP (measurement only, wrong results): 1 = no DMA refills in the loop, 2 = no fragment reads in the loop
// BM = 128 (NW = 4, BN = 128: a wave owns 64 x 64): the N = 1024 GEMMs at one sample.  256-row tiles give 64-128 of them for 256
// CUs and the two-stage 128-wide kernel of dit_gemm.hip runs them at 0.4-0.57 PFLOP/s; 128 x 128 tiles are 32 x 8 = 256, one
// per CU, on this kernel's ring and schedule (0.5 DMA pieces per MFMA instead of 0.25: the loop is DMA-issue-bound, but with
// three slabs in flight and one barrier per slab).
template <int EPI, int BN, int NW, int EXP = 0, int BM = 256>
__global__ __launch_bounds__(NW == 8 ? 512 : 256) void gemm_sliced_kernel(DeepParams p) {
    // KW = 2 (BM = BN = 128 with EIGHT waves): two K groups of 2 x 2 waves.  A ring stage holds TWO 32-wide slabs and group kg works on
    // slab kg of every stage: the 128 x 128 tile keeps its 64 x 64 wave tiles (the 4-wave form's) but every SIMD has two waves instead
    // of one to cover each other's LDS and issue latency (the 4-wave loop ran at 47 % of its MFMA floor); the groups' partial sums meet
    // once, in LDS behind the loop, and each group finishes one of the wave tile's two row blocks.
    constexpr int KW = (BM == 128 && NW == 8) ? 2 : 1, MW = NW / KW;
    constexpr int BK = 32, NS = (BM == 128 && KW == 1) ? 8 : 4;   // ring depth (LDS: 128 KiB for every tile shape)
    constexpr int WMB = BM / (16 * MW), WROWS = 32 * WMB;         // A blocks per wave: a K group's waves are (MW / 2) x 2
    constexpr int WN = BN / 2, NI = WN / 32;                      // wave tile WROWS x WN: WMB x NI accumulators
    constexpr int A_BYTES = BM * BK * 2, W_BYTES = BN * BK * 2, SUB = A_BYTES + W_BYTES, STAGE = KW * SUB;
    // LDS-DMA instructions per wave per slab: GA pieces of A, GW of W.  BN = 192 (QKV at one sample: 16 x 16 = 256 tiles, one per CU,
    // where 256-wide tiles are 192): 12 W pieces for 8 waves -- the second piece of waves 4..7 is piece 8..11 AGAIN (the same bytes to
    // the same LDS place as waves 0..3 put there), so that every wave issues, and counts, the same G operations per slab
    constexpr int W_PIECES = KW * W_BYTES / 1024, GW = (W_PIECES + NW - 1) / NW, DUP = NW * GW - W_PIECES;
    constexpr int G = KW * A_BYTES / 1024 / NW + GW;
    static_assert(KW == 1 || (A_BYTES / 1024 == NW && W_BYTES / 1024 == NW && WMB == 2 && NI == 2), "K groups: piece q of an operand is slab q's, one per wave");
    constexpr int NF = WMB + NI, MF = WMB * NI;                   // per k-substep: fragments (WMB of A, NI of W), MFMAs
    static_assert(NW == 8 || BN == 256 || (BM == 128 && BN == 128), "4 waves: 256 x 256 tiles (128 x 128 per wave) or 128 x 128 tiles (64 x 64)");
    static_assert(DUP == 0 || (DUP < NW && GW >= 1 && KW == 1), "a doubled piece only in the last W round");
    static_assert(WMB * NI >= G, "one DMA piece behind each MFMA of the spread half");
    DGS_DYNAMIC_LDS(lds);
    const int dbg = kInstrumented ? p.dbg : 0;                        // the product library carries no instrumentation
    const long long dbg_k0 = dbg == 1 ? cycle_stamp() : 0;
    const long long dbg_w0 = dbg == 1 ? wall_stamp() : 0;             // constant 100 MHz: gives the shader clock the cycle stamps ran at
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = KW > 1 ? wave % MW : wave, kg = KW > 1 ? wave / MW : 0;      // place in the tile, K group
    const int wm = wq >> 1, wn = wq & 1;
    const int frow = lane & 31, fhalf = lane >> 5;
    // Work items: every full tile, XCD-aware order.  (The tile rows with a single live 32-row block -- the two learned-token
    // rows of the DiT -- are side jobs of the first workgroups: `side_jobs`, called from the prologue.)
    const int bid = (int)blockIdx.x;
    const bf16_t* const A_all = p.A;                            // the split-K modes narrow p.A / p.W / p.K to this workgroup's range
    const bf16_t* const W_all = p.W;
    const int K_all = p.K;
    auto side_jobs = [&](const int first, const int stride) {
        if (p.tail_mode == 2) {
            // At most two live rows behind the last full tile row (the DiT's learned tokens): a 2-row GEMV on the vector pipe.  An
            // item is 8 output columns of one sample and ONE memory round trip: the NW waves split K, every wave issues all of its
            // loads (8 columns x its K range, 16 bytes per lane; a wave-instruction = 1 KiB of one row = 8 cache lines, against the
            // 32 lines of a fragment-layout gather) before the first v_dot2c_f32_bf16; the partial sums meet in LDS and 16 lanes
            // apply the epilogue element-wise (tail_store).  While the chip streams GEMM tiles a round trip costs microseconds:
            // the earlier form (a wave per 32 / NW columns, all of K, two columns per trip) held its workgroup back by 13 us at K = 4096.
            // The NW waves are KS ranges of K x CS groups of 8 columns: KS = min(NW, K / 512) keeps every lane of a wave busy
            // (a range is 512 or 1024 elements = one or two 16-byte loads per lane and row).
            const int KS = min(NW, K_all / 512), CS = NW / KS, CPI = 8 * CS;     // p.tail_cpi == CPI (launch_sliced)
            const int nblk = p.N / CPI;
            const int kq = wave % KS, cq = wave / KS;
            const int kw = K_all / KS, k_lo = kq * kw;                   // 512 or 1024
            const bool two = kw > 512;
            float* const part = reinterpret_cast<float*>(lds + (NS - 1) * STAGE);     // [k range][row][column of the item]
            const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
            for (int j = first; j < p.ntail; j += stride) {
                const int blk = j % nblk, trr = j / nblk;
                const int tm0 = ((trr / p.tail_rows) * p.rows_ps + p.full_rows + trr % p.tail_rows) * BM, tn0 = blk * CPI + cq * 8;
                const bf16_t* a_row0 = A_all + (size_t)tm0 * p.lda + k_lo + lane * 8;
                const bf16_t* w_col0 = W_all + (size_t)tn0 * p.ldw + k_lo + lane * 8;
                const int er = tid / CPI, ec = tid - er * CPI;           // the element thread `tid` finishes (tid < 2 CPI)
                const int erow = tm0 + er;
                const bool finisher = tid < 2 * CPI && erow - (erow / p.rows_per_batch) * p.rows_per_batch < p.valid_rows;
                TailOperands ops{0.f, 0.f, 0.f};
                if (finisher) ops = tail_prefetch<EPI>(p, erow, blk * CPI + ec);
                uint4 a[2][2], w[8][2];
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
                    const bool on = ch == 0 || two;
#pragma unroll
                    for (int r = 0; r < 2; ++r) a[r][ch] = on ? *reinterpret_cast<const uint4*>(a_row0 + (size_t)r * p.lda + ch * 512) : zero;
#pragma unroll
                    for (int c = 0; c < 8; ++c) w[c][ch] = on ? *reinterpret_cast<const uint4*>(w_col0 + (size_t)c * p.ldw + ch * 512) : zero;
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    float s0 = dot8_bf16(a[0][0], w[c][0], 0.f), s1 = dot8_bf16(a[1][0], w[c][0], 0.f);
                    s0 = dot8_bf16(a[0][1], w[c][1], s0); s1 = dot8_bf16(a[1][1], w[c][1], s1);
                    s0 = wave_sum_lane63(s0); s1 = wave_sum_lane63(s1);
                    if (lane == 63) { part[(kq * 2 + 0) * CPI + cq * 8 + c] = s0; part[(kq * 2 + 1) * CPI + cq * 8 + c] = s1; }
                }
                __syncthreads();
                if (finisher) {
                    float v = 0.f;
                    for (int q = 0; q < KS; ++q) v += part[(q * 2 + er) * CPI + ec];
                    tail_store<EPI>(p, erow, blk * CPI + ec, v, ops);
                }
                __syncthreads();
            }
            return;
        }
        for (int j = p.nsplit > 1 ? p.ntail : first; j < p.ntail; j += stride) {
            constexpr int CBW = BN == 256 ? 1 : 2, KQ = NW / CBW, TPC = BN / (32 * CBW);   // column blocks, K ranges per item; items per tile column
            const int sub = j % TPC, ttn = (j / TPC) % p.tiles_n, trr = j / (TPC * p.tiles_n);
            const int tm0 = ((trr / p.tail_rows) * p.rows_ps + p.full_rows + trr % p.tail_rows) * BM, tn0 = ttn * BN + sub * 32 * CBW;
            const int cbl = wave % CBW, kq = wave / CBW, kper = K_all / KQ;  // kper % 128 == 0 (launch_sliced)
            const bf16_t* arow = A_all + (size_t)(tm0 + frow) * p.lda + kq * kper + fhalf * 8;
            const bf16_t* wrow = W_all + (size_t)(tn0 + cbl * 32 + frow) * p.ldw + kq * kper + fhalf * 8;
            f32x16 acc1[1];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[0][r] = 0.f;
            if (kper % 256 == 0)
                for (int k = 0; k < kper; k += 256) direct_block_mfma<16>(arow + k, wrow + k, acc1[0]);
            else
                for (int k = 0; k < kper; k += 128) direct_block_mfma<8>(arow + k, wrow + k, acc1[0]);
            float* red = reinterpret_cast<float*>(lds + (NS - 1) * STAGE);          // (KQ - 1) * CBW * 4 KiB <= STAGE
            if (kq > 0)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[((kq - 1) * CBW + cbl) * 1024 + r * 64 + lane] = acc1[0][r];
            __syncthreads();
            if (kq == 0) {
#pragma unroll
                for (int q = 1; q < KQ; ++q)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[0][r] += red[((q - 1) * CBW + cbl) * 1024 + r * 64 + lane];
                store_block<EPI, 1>(p, acc1, tm0 + 4 * fhalf, tn0 + cbl * 32, lane);
            }
            __syncthreads();                                            // `red` is rewritten by the next item / the ring
        }
    };
    if (p.nfull_items == 0) { side_jobs(bid, (int)gridDim.x); return; }           // no full tile at all (<= 32 valid rows per sample)
    // One-round grids that leave CUs idle (QKV at one sample: 204 tiles of 256 x 256 on 256 CUs): the side jobs go to workgroups of
    // their own behind the tiles -- the dispatcher puts them on the idle CUs at once, and no tile waits for a GEMV (as side jobs
    // of the first 96 tile workgroups they held the whole launch back by ~9 us: 37 -> 46)
    if (p.tail_wgs > 0 && bid >= p.nfull_items) { side_jobs(bid - p.nfull_items, p.tail_wgs); return; }
    int tile;
    if (p.nsplit > 1) {
        // split-K (weight gradients): item = (split, tile), split-major so that an XCD's tiles share operand panels; a split is
        // a range of 128-wide K units of one sample; its partial product goes to its own [M, N] f32 plane
        const int all = xcd_remap(bid, p.nfull_items * p.nsplit);
        const int sp = all / p.nfull_items, b = sp / p.splits_per_batch, part = sp % p.splits_per_batch;
        tile = all - sp * p.nfull_items;
        const int units = p.K / 128, u0 = part * units / p.splits_per_batch, u1 = (part + 1) * units / p.splits_per_batch;
        p.A += (size_t)b * p.a_batch_stride + u0 * 128;
        p.W += (size_t)b * p.w_batch_stride + u0 * 128;
        p.out = static_cast<float*>(p.out) + (size_t)sp * p.out_split_stride;
        p.K = (u1 - u0) * 128;
    } else if (p.map2d) {
        // An XCD's tiles as a 4 x 2 arrangement of blocks of the tile grid instead of a run of whole tile rows: with one tile per CU
        // (16 x 16 tiles: fc1, QKV on 192-wide tiles) an XCD then reads 4 tile rows of A and HALF of W -- 6.2 MB instead of 9.4 MB
        // (fc1) through its own L2 -- and the launch fetches a third less from HBM.  (The ring hides the misses either way: the launch
        // is no faster, profiles/r05_gemm_xcd_map_ab.txt; it is the same time on less traffic.)
        const int xcd = bid & 7, slot = bid >> 3, rows_all = p.nfull_items / p.tiles_n;
        const int rper = rows_all >> 2, cper = p.tiles_n >> 1;
        tile = ((xcd >> 1) * rper + slot / cper) * p.tiles_n + (xcd & 1) * cper + slot % cper;
    } else {
        tile = xcd_remap(bid, p.nfull_items);
    }
    const int tn = tile % p.tiles_n, rr = tile / p.tiles_n;
    const int m0 = ((rr / p.full_rows) * p.rows_ps + rr % p.full_rows) * BM, n0 = tn * BN;
    f32x16 acc[WMB][NI];
#pragma unroll
    for (int i = 0; i < WMB; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // fragment f of substep ks: f < WMB: A rows wm*WROWS + 32 f + frow;  f >= WMB: W rows wn*WN + 32 (f - WMB) + frow; chunk 2 ks + fhalf
    int foff[2][NF];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int f = 0; f < NF; ++f)
            foff[ks][f] = kg * SUB + (f < WMB ? slab_off<BK>(wm * WROWS + 32 * f + frow, 2 * ks + fhalf) : A_BYTES + slab_off<BK>(wn * WN + 32 * (f - WMB) + frow, 2 * ks + fhalf));
    auto frag = [&](int slot, int ks, int f) { return *reinterpret_cast<const bf16x8*>(lds + slot * STAGE + foff[ks][f]); };
    const int nk = p.K / (BK * KW);                               // ring stages to walk: a multiple of NS
    constexpr int GA = KW * A_BYTES / 1024 / NW;                  // DMA pieces per wave per stage: GA of A, G - GA of W
    // The DMA of a piece, all-scalar addressing (lds_dma_scalar): global address = SGPR pair (the operand's tile row 0 at the slab's
    // k) + one 32-bit VGPR offset per piece (the lane's row and 16-byte chunk: loop-invariant), LDS address = M0 = SGPR + literal.
    constexpr int RPI = 1024 / (2 * BK), CPR = BK / 8;
    char* const lds_mine = lds + wave * 1024;
    const uint32_t lds_wave = lds_address(lds) + (uint32_t)wave * 1024u;
    const char* sb_a = reinterpret_cast<const char*>(p.A + (size_t)m0 * p.lda);        // + 2 BK bytes per slab staged
    const char* sb_w = reinterpret_cast<const char*>(p.W + (size_t)n0 * p.ldw);
    uint32_t vo_a[GA], vo_w[G - GA];
    const int back = (DUP > 0 && wave + NW * (GW - 1) >= W_PIECES) ? DUP : 0;      // pieces this wave's last W piece steps back (a doubled piece)
    {
        const int r0 = lane / CPR, chunk = (lane % CPR) ^ (BK == 64 ? (r0 >> 1) & 7 : (r0 >> 2) & 3);     // piece rows are multiples of 16
        if constexpr (KW > 1) {                                   // piece q of an operand: rows of piece `wave`, slab q of the stage (+ BK elements)
#pragma unroll
            for (int q = 0; q < GA; ++q) vo_a[q] = (uint32_t)(((wave * RPI + r0) * p.lda + chunk * 8 + q * BK) * 2);
#pragma unroll
            for (int q = 0; q < G - GA; ++q) vo_w[q] = (uint32_t)(((wave * RPI + r0) * p.ldw + chunk * 8 + q * BK) * 2);
        } else {
#pragma unroll
            for (int q = 0; q < GA; ++q) vo_a[q] = (uint32_t)((((wave + NW * q) * RPI + r0) * p.lda + chunk * 8) * 2);
#pragma unroll
            for (int q = 0; q < G - GA; ++q) vo_w[q] = (uint32_t)((((wave + NW * q - (q == GW - 1 ? back : 0)) * RPI + r0) * p.ldw + chunk * 8) * 2);
        }
    }
    char* const lds_mine_last = lds_mine - back * 1024;
    const uint32_t lds_wave_last = lds_wave - (uint32_t)back * 1024u;
    auto dma_piece = [&](auto slotc, auto qc) {                   // piece q (A pieces first) of the NEXT slab into ring slot `slot`
        constexpr int slot = decltype(slotc)::value, q = decltype(qc)::value;
        if constexpr (KW > 1 && q < GA) lds_dma_scalar<slot * STAGE + q * SUB>(sb_a, vo_a[q], lds_wave, lds_mine);
        else if constexpr (KW > 1) lds_dma_scalar<slot * STAGE + (q - GA) * SUB + A_BYTES>(sb_w, vo_w[q - GA], lds_wave, lds_mine);
        else if constexpr (q < GA) lds_dma_scalar<slot * STAGE + NW * q * 1024>(sb_a, vo_a[q], lds_wave, lds_mine);
        else if constexpr (DUP > 0 && q == G - 1) lds_dma_scalar<slot * STAGE + A_BYTES + NW * (q - GA) * 1024>(sb_w, vo_w[q - GA], lds_wave_last, lds_mine_last);
        else lds_dma_scalar<slot * STAGE + A_BYTES + NW * (q - GA) * 1024>(sb_w, vo_w[q - GA], lds_wave, lds_mine);
        if constexpr (q == G - 1) { sb_a += 2 * BK * KW; sb_w += 2 * BK * KW; }
    };
    auto stage_next = [&](auto slotc) { sliced_for<0, G>([&](auto qc) { dma_piece(slotc, qc); }); };
    // ---- prologue: slabs 0 .. NS-2 in flight, then the fragments of slab 0, substep 0.  A workgroup with a side job issues only
    //      slabs 0 and 1 first: vector memory returns in order, and behind all NS-1 slabs (112 KiB with the 8-stage ring) the side
    //      job's one round trip became ~10 k cycles; the other slabs go out when the side job is done ----
    //      (8-stage ring only: fc2 +2.4 -> +1.0 us for the two learned-token rows; with the 4-stage ring there is one slab to hold
    //      back, and holding it back costs more than it saves: fc1 +4.1 -> +5.4)
    const bool has_side_job = NS > 4 && p.tail_wgs == 0 && bid < p.ntail && p.nsplit == 1;
    sliced_for<0, 2>([&](auto sc) { stage_next(sc); });
    if (!has_side_job) sliced_for<2, NS - 1>([&](auto sc) { stage_next(sc); });
    // ---- tiles with a single live 32-row block (the learned-token rows of every sample), folded into the first workgroups ----
    // An item is 32 rows x 32 columns (64 for BN = 128): the NW waves are ranges of K, every wave pulls its fragments straight from
    // L2 in one round trip per 128 / 256 columns of K (one trip for K = 1024), the ranges meet in the ring stage that iteration 0
    // refills.  The loads are gathers -- 32 rows, 32 cache lines per wave-instruction -- and the CU's address path, not latency,
    // sets their price (~2.5 cycles per line): the smaller the item, the more workgroups share that cost.  It runs here, while this workgroup's first slabs are in flight, because a memory round trip costs
    // 2-4 us when the chip is streaming GEMM tiles and nothing else can hide it: as workgroups of their own (256 full tiles on
    // 256 CUs, so they start when the first full tiles retire) these items added 36 us when a wave walked all of K for one
    // column block, and still 14 us in this one-trip form.
    if (p.tail_wgs == 0) side_jobs(bid, (int)gridDim.x);
    if (has_side_job) sliced_for<2, NS - 1>([&](auto sc) { stage_next(sc); });
    // Slabs 0 and 1 have to be there before iteration 0 (its second half prefetches fragments of slab 1); slabs 2 .. NS-2 may stay
    // in flight -- with the 8-stage ring of the 128 x 128 tiles, waiting for all seven (112 KiB per CU, every CU at once) cost
    // ~10 k cycles of every tile.  Stores a side job left in flight are older than those slabs: "at most (NS-3) G operations
    // outstanding" then still means slabs 0 and 1 have landed (loads return in order; a lingering store only makes the wait longer).
    wait_vmcnt<(NS - 3) * G>();
    raw_barrier();
    bf16x8 fr[2][NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) fr[0][f] = frag(0, 0, f);

    long long dbg_wait = 0, dbg_bar = 0;
    // NW = 4: the wave is alone on its SIMD, so a burst of G DMA issues at the top of the iteration is a bubble in the MFMA
    // stream; one DMA goes out behind each MFMA of the second half of substep 0 instead (those slices have no LDS read)
    constexpr bool SPREAD = NW == 4;
    // The gated-residual epilogue of the 128 x 128 tiles reads 64 KiB of the residual stream per workgroup; every workgroup of the
    // one-round grid reaches its epilogue at the same time and a strip's loads cannot move above the previous strip's stores
    // (out may alias resid), so the tile paid two exposed round trips (~16 k of a 90 k-cycle tile).  Here the whole tile's
    // residual (64 registers) is requested behind the LAST DMA, eight slabs before the loop ends, and the counted waits of the
    // remaining iterations let those loads stay in flight.
    constexpr bool PREFETCH_RESID = EPI == DGS_EPI_GATE_RESIDUAL && BM == 128;
    constexpr int PRE_BLOCKS = KW > 1 ? 1 : WMB;                            // row blocks of the wave tile this wave finishes (K groups: block kg)
    constexpr int PRE_LOADS = PREFETCH_RESID ? PRE_BLOCKS * (NI / 2) * 8 : 0;       // float4 loads per lane
    float4 pre[PREFETCH_RESID ? PRE_BLOCKS * (NI / 2) : 1][8];
    // `dbg_tag`: the cycle stamps of DGS_GEMM_DBG=1 exist only in the copy of the loop that a debug run takes -- as run-time tests of
    // p.dbg they were three compare + branch pairs per slab in every run (tools/ubench/issue_bench: ~16 cycles each beside MFMAs)
    auto iteration = [&](int t, auto slot_tag, auto refill_tag, auto pre_tag, auto left_tag, auto dbg_tag) {    // slot == t % NS, a literal at the call sites
        constexpr bool DBG = decltype(dbg_tag)::value;
        constexpr int slot = decltype(slot_tag)::value;
        constexpr int PRE_N = decltype(pre_tag)::value;                // residual loads in flight at the end of this iteration
        constexpr bool refill = decltype(refill_tag)::value;           // slab t + NS - 1 exists; it goes into the stage of slab t - 1
        if constexpr (refill && !SPREAD && EXP != 1) stage_next(SIC<(slot + NS - 1) % NS>{});
        sliced_for<0, 2 * MF>([&](auto jc) {
            constexpr int J = decltype(jc)::value, ks = J / MF, i = (J % MF) / NI, j = J % NI;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[ks][i], fr[ks][WMB + j], acc[i][j], 0, 0, 0);
            sched_fence();                                         // the MFMA leads its slice: what follows issues in its shadow
            // the other register half: substep 1 of this slab, then substep 0 of the next one (stale data behind the last slab)
            constexpr int f = J % MF;
            if constexpr (f < NF && EXP != 2) fr[ks ^ 1][f] = frag(ks == 0 ? slot : (slot + 1) % NS, ks ^ 1, f);
            if constexpr (SPREAD && refill && ks == 0 && f >= MF - G && EXP != 1) {
                constexpr int q = f - (MF - G);
                dma_piece(SIC<(slot + NS - 1) % NS>{}, SIC<q>{});
            }
            sched_fence();
        });
        if constexpr (PRE_N > 0 && refill) {                       // the tail's first iteration: its DMAs were the last ones
            if constexpr (KW > 1) {
#pragma unroll
                for (int j = 0; j < NI; j += 2) residual_prefetch<2>(p, m0 + wm * WROWS + 32 * kg, n0 + wn * WN + 32 * j, lane, pre[j / 2]);
            } else {
#pragma unroll
                for (int i = 0; i < WMB; ++i)
#pragma unroll
                    for (int j = 0; j < NI; j += 2) residual_prefetch<2>(p, m0 + wm * WROWS + 32 * i, n0 + wn * WN + 32 * j, lane, pre[i * (NI / 2) + j / 2]);
            }
        }
        // slab t+1 (and older) has landed once at most this iteration's own DMAs are outstanding; then everybody is also
        // done reading slab t
        long long w0 = 0;
        if constexpr (DBG) w0 = cycle_stamp();
        // The second half of iteration t+1 already prefetches fragments of slab t+2, so slabs <= t+2 must have landed by the end of
        // iteration t: slabs t+3 .. t+NS-1 stay in flight across the barrier (DMAs complete in issue order).  NS = 4: only the slab
        // issued in this iteration; NS = 8 (128 x 128 tiles, whose A operand streams from beyond L2): five slabs, ~3 k cycles of
        // latency tolerance.
        if constexpr (PRE_N > 0) {
            // the residual loads went out behind the last DMA (below) and stay in flight to the epilogue: slabs t+3 .. nk-1 plus them
            constexpr int left = decltype(left_tag)::value;       // slabs behind t+2 that exist
            wait_vmcnt<left * G + PRE_N>();
        } else if constexpr (refill) wait_vmcnt<(NS - 3) * G>();
        else wait_vmcnt<0>();
        long long w1 = 0;
        if constexpr (DBG) w1 = cycle_stamp();
        raw_barrier();
        if constexpr (DBG) { dbg_wait += w1 - w0; dbg_bar += cycle_stamp() - w1; }
        (void)t;
    };
    const long long dbg_t0 = dbg == 1 ? cycle_stamp() : 0;
    const long long dbg_w1 = dbg == 1 ? wall_stamp() : 0;
    auto k_loop = [&](auto dbg_tag) {
        for (int t = 0; t < nk - NS; t += NS)                      // unrolled by the ring depth: slots are literals
            sliced_for<0, NS>([&](auto sc) { iteration(t + decltype(sc)::value, sc, std::true_type{}, SIC<0>{}, SIC<0>{}, dbg_tag); });
        sliced_for<0, NS>([&](auto sc) {                           // the last slab goes out in the first of the last NS iterations
            constexpr int S = decltype(sc)::value;
            iteration(nk - NS + S, sc, std::integral_constant<bool, S == 0>{}, SIC<PRE_LOADS>{}, SIC<(NS - S - 3 > 0 ? NS - S - 3 : 0)>{}, dbg_tag);
        });
    };
    if constexpr (kInstrumented) {
        if (dbg == 1) k_loop(std::true_type{});
        else k_loop(std::false_type{});
    } else {
        k_loop(std::false_type{});
    }
    const long long dbg_t1 = dbg == 1 ? cycle_stamp() : 0;
    const long long dbg_w2 = dbg == 1 ? wall_stamp() : 0;
    if constexpr (KW > 1) {
        // The two K groups hold partial sums of the same 64 x 64 wave tiles.  Wave (kg, wq) keeps row block kg and hands the other one
        // to its partner (same wq, other group) through the idle ring: [wq][owner][block j][register][lane], lane-contiguous floats.
        float* const xch = reinterpret_cast<float*>(lds);
        auto region = [&](int owner) { return xch + ((wq * 2 + owner) * NI * 16) * 64 + lane; };
        auto give = [&](const f32x16 (&blk)[NI], float* dst) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) dst[(j * 16 + r) * 64] = blk[j][r];
        };
        auto take = [&](f32x16 (&blk)[NI], const float* src) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) blk[j][r] += src[(j * 16 + r) * 64];
        };
        if (kg == 0) give(acc[1], region(1)); else give(acc[0], region(0));
        __syncthreads();
        if (kg == 0) take(acc[0], region(0)); else take(acc[1], region(1));
        __syncthreads();                                           // the patches below overwrite the exchange regions
        char* patch = lds + wave * epi_strip_bytes(2);
        const int mb = m0 + wm * WROWS + 32 * kg, nb = n0 + wn * WN;
        if (kg == 0) {
            if constexpr (PREFETCH_RESID) store_strip<EPI, 2>(p, &acc[0][0], mb, nb, lane, patch, pre[0]);
            else store_strip<EPI, 2>(p, &acc[0][0], mb, nb, lane, patch);
        } else {
            if constexpr (PREFETCH_RESID) store_strip<EPI, 2>(p, &acc[1][0], mb, nb, lane, patch, pre[0]);
            else store_strip<EPI, 2>(p, &acc[1][0], mb, nb, lane, patch);
        }
    } else
    if (epi_staged<EPI>(p)) {          // the ring is idle now (everybody passed the last barrier): a private LDS patch per wave
        char* patch = lds + wave * epi_strip_bytes(2);
        if constexpr (NI % 2 == 1) {
            // three column blocks (BN = 192): a two-block strip and a single one; store_strip decides "V feature" per strip, so the one
            // strip that would straddle the K | V boundary of the QKV output leaves as two single blocks
            const int nb = n0 + wn * WN, v0 = (p.N / 3) * 2;
            const bool straddles = EPI == DGS_EPI_QKV && nb < v0 && nb + 64 > v0;
#pragma unroll
            for (int i = 0; i < WMB; ++i) {
                const int mb = m0 + wm * WROWS + 32 * i;
                if (straddles) { store_strip<EPI, 1>(p, &acc[i][0], mb, nb, lane, patch); store_strip<EPI, 1>(p, &acc[i][1], mb, nb + 32, lane, patch); }
                else store_strip<EPI, 2>(p, &acc[i][0], mb, nb, lane, patch);
                store_strip<EPI, 1>(p, &acc[i][NI - 1], mb, nb + 32 * (NI - 1), lane, patch);
            }
        } else
#pragma unroll
        for (int i = 0; i < WMB; ++i)
#pragma unroll
            for (int j = 0; j < NI; j += 2) {
                if constexpr (PREFETCH_RESID) store_strip<EPI, 2>(p, &acc[i][j], m0 + wm * WROWS + 32 * i, n0 + wn * WN + 32 * j, lane, patch, pre[i * (NI / 2) + j / 2]);
                else store_strip<EPI, 2>(p, &acc[i][j], m0 + wm * WROWS + 32 * i, n0 + wn * WN + 32 * j, lane, patch);
            }
    } else {
#pragma unroll
        for (int i = 0; i < WMB; ++i) store_block<EPI, NI>(p, acc[i], m0 + wm * WROWS + 32 * i + 4 * fhalf, n0 + wn * WN, lane);
    }
    if constexpr (kInstrumented) if (dbg == 1) {
        wait_vmcnt<0>();
        if (tid == 0 && bid < 1024) {
            dgs_gemm_tl[bid][0] = (unsigned)dbg_w0; dgs_gemm_tl[bid][1] = (unsigned)dbg_w1; dgs_gemm_tl[bid][2] = (unsigned)dbg_w2;
            dgs_gemm_tl[bid][3] = (unsigned)wall_stamp();
        }
        if (tid == 0) {
            atomicMax(reinterpret_cast<unsigned long long*>(&dgs_gemm_dbg[6]), (unsigned long long)(cycle_stamp() - dbg_k0));
            atomicMax(reinterpret_cast<unsigned long long*>(&dgs_gemm_dbg[7]), (unsigned long long)dbg_wait);
            atomicMin(reinterpret_cast<unsigned long long*>(&dgs_gemm_dbg[8]), (unsigned long long)dbg_k0);
            atomicMax(reinterpret_cast<unsigned long long*>(&dgs_gemm_dbg[9]), (unsigned long long)cycle_stamp());
        }
        if (blockIdx.x == 0 && tid == 0) {
            dgs_gemm_dbg[0] = dbg_t1 - dbg_t0; dgs_gemm_dbg[1] = dbg_wait; dgs_gemm_dbg[2] = dbg_bar; dgs_gemm_dbg[3] = nk;
            dgs_gemm_dbg[4] = dbg_t0 - dbg_k0; dgs_gemm_dbg[5] = cycle_stamp() - dbg_t1;
            dgs_gemm_dbg[10] = cycle_stamp() - dbg_k0; dgs_gemm_dbg[11] = wall_stamp() - dbg_w0;
        }
    }
}

template <int EPI, int BN, int NW = 8, int BM = 256>
static int launch_sliced(DeepParams p, hipStream_t st) {
    constexpr int LDS = (BM == 128 ? 8 : 4) * (BM * 32 * 2 + BN * 32 * 2);   // 128 KiB (256 x 256) / 96 KiB (256 x 128) / 128 KiB (128 x 128: 8 slabs, as 8 or 4 x 2)
    p.tiles_n = p.N / BN;
    p.rows_ps = p.rows_per_batch / BM;
    p.full_rows = 0; p.tail_rows = 0;
    // A tile row with a single live 32-row block is not a tile: its blocks are side jobs of the first workgroups -- a two-row
    // GEMV on the vector pipe when at most 2 rows are live (the DiT's learned tokens), MFMA items otherwise; if the shape fits
    // neither, it runs the ring like a full row.
    const int last_live = p.valid_rows - (p.valid_rows - 1) / BM * BM;                       // live rows of the last tile row that has any
    const bool gemv_ok = last_live <= 2 && p.K >= 512 && p.K <= 4096 && (p.K & (p.K - 1)) == 0 && p.N % 64 == 0;
    const bool mfma_ok = p.K % (NW / (BN == 256 ? 1 : 2) * 128) == 0;
    for (int i = 0; i < p.rows_ps; ++i) {                          // per sample: tile rows with >= 2 / exactly 1 live 32-row blocks
        const int live = (p.valid_rows - i * BM + 31) / 32;
        if (live > 1 || (live == 1 && !gemv_ok && !mfma_ok)) ++p.full_rows; else if (live == 1) ++p.tail_rows;
    }
    if (p.rows_external) {
        if (!gemv_ok || p.nsplit > 1) return DGS_ERR_INVALID_ARGUMENT;
        p.tail_rows = 0;
    }
    const int samples = p.M / p.rows_per_batch;
    p.nfull_items = samples * p.full_rows * p.tiles_n;
    p.tail_mode = gemv_ok ? 2 : 1;
    const int gemv_ks = NW < p.K / 512 ? NW : p.K / 512, gemv_cpi = 8 * (NW / (gemv_ks > 0 ? gemv_ks : 1));
    p.ntail = samples * p.tail_rows * (gemv_ok ? p.N / gemv_cpi : p.tiles_n * (BN / (BN == 256 ? 32 : 64)));
    p.ntiles = p.nfull_items ? p.nfull_items : p.ntail;
    p.tail_wgs = 0;
    if (p.nsplit <= 1 && p.ntail > 0 && p.nfull_items > 0) {
        static const int ncu = compute_unit_count();               // (the emulator reads DGS_EMU_CUS: tests exercise the tail-only workgroups with it)
        const int idle = ncu - p.nfull_items;                      // a tile workgroup owns its CU's LDS
        if (idle > 0) p.tail_wgs = idle < p.ntail ? idle : p.ntail;
        p.ntiles += p.tail_wgs;
    }
    if (p.nsplit > 1) {
        if (p.tail_rows || samples != 1) return DGS_ERR_INVALID_ARGUMENT;
        p.ntiles = p.nfull_items * p.nsplit;
    }
    static const int no_map2d = getenv("DGS_GEMM_NO_MAP2D") ? atoi(getenv("DGS_GEMM_NO_MAP2D")) : 0;    // measurement aid
    const int rows_all = samples * p.full_rows;
    p.map2d = !no_map2d && BM == 256 && p.nsplit <= 1 && p.tail_wgs == 0 && rows_all % 4 == 0 && p.tiles_n % 2 == 0 && p.nfull_items % 8 == 0 &&
              p.nfull_items <= compute_unit_count_cached() ? 1 : 0;
    auto kern = gemm_sliced_kernel<EPI, BN, NW, 0, BM>;
    if constexpr (kInstrumented && EPI == DGS_EPI_F32 && BN == 256 && BM == 256) {   // DGS_GEMM_EXP=1|2: the measurement variants (instrumented library only)
        static const int exp = getenv("DGS_GEMM_EXP") ? atoi(getenv("DGS_GEMM_EXP")) : 0;
        if (exp == 1) kern = gemm_sliced_kernel<EPI, BN, NW, 1>;
        if (exp == 2) kern = gemm_sliced_kernel<EPI, BN, NW, 2>;
        if (exp && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return DGS_ERR_DEVICE;
    }
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return DGS_ERR_DEVICE;
        attr_set = true;
    }
    static const int dbg = kInstrumented && getenv("DGS_GEMM_DBG") ? atoi(getenv("DGS_GEMM_DBG")) : 0;   // 1: cycle stamps (instrumented library only)
    p.dbg = dbg;
#ifdef DGS_INSTRUMENT
    if (dbg == 1) {
        const long long z[4] = {0, 0, 0x7fffffffffffffffLL, 0};
        (void)hipStreamSynchronize(st);
        if (hipMemcpyToSymbol(HIP_SYMBOL(dgs_gemm_dbg), z, sizeof(z), 6 * sizeof(long long), hipMemcpyHostToDevice) != hipSuccess) fprintf(stderr, "[gemm dbg] reset failed\n");
    }
#endif
    hipLaunchKernelGGL(kern, dim3(p.ntiles), dim3(64 * NW), LDS, st, p);
#ifdef DGS_INSTRUMENT
    if (dbg == 1) {
        long long h[12];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(dgs_gemm_dbg), sizeof(h));
        {
            // the launch's timeline: per phase min / mean / max over the tile workgroups, in us from the first workgroup's start
            static unsigned tl[1024][4];
            (void)hipMemcpyFromSymbol(tl, HIP_SYMBOL(dgs_gemm_tl), sizeof(tl));
            const int n = p.ntiles < 1024 ? p.ntiles : 1024;
            unsigned t0 = 0xffffffffu;
            for (int i = 0; i < n; ++i) t0 = tl[i][0] < t0 ? tl[i][0] : t0;
            double acc[4] = {0, 0, 0, 0}, mn[4] = {1e9, 1e9, 1e9, 1e9}, mx[4] = {0, 0, 0, 0};
            const int nt = p.nfull_items < n ? p.nfull_items : n;
            for (int i = 0; i < nt; ++i)
                for (int k = 0; k < 4; ++k) { const double v = (tl[i][k] - t0) / 100.0; acc[k] += v; mn[k] = v < mn[k] ? v : mn[k]; mx[k] = v > mx[k] ? v : mx[k]; }
            fprintf(stderr, "[gemm timeline] %d tile workgroups (of %d), us since the first start: start %.2f/%.2f/%.2f  loop start %.2f/%.2f/%.2f  loop end %.2f/%.2f/%.2f  end %.2f/%.2f/%.2f (min/mean/max)\n",
                    nt, n, mn[0], acc[0] / nt, mx[0], mn[1], acc[1] / nt, mx[1], mn[2], acc[2] / nt, mx[2], mn[3], acc[3] / nt, mx[3]);
            double xe[8] = {0}, xl[8] = {0}; int xn[8] = {0};
            for (int i = 0; i < nt; ++i) { xe[i & 7] += (tl[i][3] - t0) / 100.0; xl[i & 7] += (tl[i][2] - tl[i][1]) / 100.0; ++xn[i & 7]; }
            fprintf(stderr, "[gemm timeline] by block id %% 8 (XCD): mean loop us / mean end us:");
            for (int x = 0; x < 8; ++x) fprintf(stderr, " %.2f/%.2f", xn[x] ? xl[x] / xn[x] : 0.0, xn[x] ? xe[x] / xn[x] : 0.0);
            double se = 0, s2 = 0;      // workgroups with a side job (the first ntail) against the rest: time to the loop
            int c1 = 0, c2 = 0;
            for (int i = 0; i < nt; ++i) { if (p.tail_wgs == 0 && i < p.ntail) { se += (tl[i][1] - tl[i][0]) / 100.0; ++c1; } else { s2 += (tl[i][1] - tl[i][0]) / 100.0; ++c2; } }
            fprintf(stderr, "\n[gemm timeline] prologue us: %d workgroups with a side job %.2f, %d without %.2f\n", c1, c1 ? se / c1 : 0.0, c2, c2 ? s2 / c2 : 0.0);
        }
        fprintf(stderr, "[gemm dbg] M=%d N=%d K=%d BN=%d: prologue %lld, loop %lld cycles (%lld per slab), vmcnt wait %lld, barrier %lld, epilogue %lld | slowest wg %lld cycles, max vmcnt wait %lld per slab, workgroup 0: %lld cycles in %.2f us = %.0f MHz\n", p.M,
                p.N, p.K, BN, h[4], h[0], h[0] / h[3], h[1] / h[3], h[2] / h[3], h[5], h[6], h[7] / h[3], h[10], h[11] / 100.0, h[10] / (h[11] / 100.0));
    }
#endif
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

// Tile width (256 / 128) of the sliced kernel for this shape; 0 when it is not eligible.
int sliced_gemm_tile(int M, int N, int K, int epilogue, int k_per_batch, int rows_per_batch, int valid_rows) {
    if (k_per_batch != K || K % 128 || M % 256 || rows_per_batch % 256) return 0;
    if (epilogue == DGS_EPI_QKV && N % 3) return 0;
    int full_rows = 0;
    for (int i = 0; i < rows_per_batch / 256; ++i)
        if ((valid_rows - i * 256 + 31) / 32 > 1) ++full_rows;
    // QKV when 256-wide tiles leave CUs without one and 192-wide ones do not (one sample: 12 x 16 = 192 tiles -> 16 x 16 = 256, a tile
    // 25 % shorter on every CU; the learned tokens' rows are not side jobs of this launch any more, so it needs no idle CU for them)
    static const int no192 = getenv("DGS_GEMM_NO_BN192") ? atoi(getenv("DGS_GEMM_NO_BN192")) : 0;      // measurement aid
    const int ncu = compute_unit_count_cached();
    const int rows_all = (M / rows_per_batch) * full_rows;
    // (small token counts keep their kernel: a handful of 192-wide tiles on 256 CUs is not what this is for)
    const bool fills = rows_all * (N / 192) >= 160 || ncu < 160;                      // (the emulator's chip has 6 CUs)
    if (!no192 && epilogue == DGS_EPI_QKV && N % 192 == 0 && N % 256 == 0 && rows_all * (N / 256) < ncu && rows_all * (N / 192) <= ncu && fills) return 192;
    if (N % 256 == 0 && rows_all * (N / 256) >= 160) return 256;
    return N % 128 ? 0 : 128;
}

// 128 x 128 tiles (bn = -128 in launch_sliced_gemm): eligibility
bool sliced128_eligible(int M, int N, int K, int epilogue, int k_per_batch, int rows_per_batch) {
    return k_per_batch == K && K % 256 == 0 && M % 128 == 0 && rows_per_batch % 128 == 0 && N % 128 == 0 && epilogue != DGS_EPI_QKV;
}

// two-row GEMV side jobs: the condition of launch_sliced, for callers that take those rows out of the launch
bool sliced_rows_are_gemv(int K, int N, int valid_rows) {
    const int last_live = valid_rows - (valid_rows - 1) / 256 * 256;
    return last_live <= 2 && K >= 512 && K <= 4096 && (K & (K - 1)) == 0 && N % 64 == 0;
}

int launch_sliced_gemm(const DgsDitGemmArgs* a, int bn, int rows_per_batch, int valid_rows, hipStream_t st, bool quad, bool rows_external) {
    DeepParams p;
    p.rows_external = rows_external ? 1 : 0;
    p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->ldo; p.gate_stride = a->gate_stride;
    p.rows_per_batch = rows_per_batch; p.valid_rows = valid_rows; p.dbg = 0; p.nsplit = 1; p.splits_per_batch = 1;
    p.a_batch_stride = p.w_batch_stride = p.out_split_stride = 0;
    p.A = a->A; p.W = a->W; p.bias = a->bias; p.out = a->out; p.gate = a->gate; p.vt = a->vt; p.aux = a->aux;
    p.q_scale = a->q_scale != 0.0f ? a->q_scale : 1.0f;
    p.resid = a->resid ? a->resid : static_cast<const float*>(a->out);
    static const int s128_nw4 = getenv("DGS_GEMM_S128_NW4") ? atoi(getenv("DGS_GEMM_S128_NW4")) : 0;   // measurement aid: the 4-wave form of the 128 x 128 tile
#define DGS_SLICED_CASE(E) case E: return bn == 256 ? (quad ? launch_sliced<E, 256, 4>(p, st) : launch_sliced<E, 256>(p, st)) : \
                                   bn == 128 ? launch_sliced<E, 128>(p, st) : s128_nw4 ? launch_sliced<E, 128, 4, 128>(p, st) : launch_sliced<E, 128, 8, 128>(p, st)
    if (bn == 192) return a->epilogue == DGS_EPI_QKV && !quad ? launch_sliced<DGS_EPI_QKV, 192>(p, st) : DGS_ERR_INVALID_ARGUMENT;
    switch (a->epilogue) {
        DGS_SLICED_CASE(DGS_EPI_BF16);
        DGS_SLICED_CASE(DGS_EPI_GELU_BF16);
        DGS_SLICED_CASE(DGS_EPI_GATE_RESIDUAL);
        DGS_SLICED_CASE(DGS_EPI_F32);
        DGS_SLICED_CASE(DGS_EPI_QKV);
        DGS_SLICED_CASE(DGS_EPI_DGELU_BF16);
        default: return DGS_ERR_INVALID_ARGUMENT;
    }
#undef DGS_SLICED_CASE
}

// ---- split-K on the sliced kernel (weight gradients: [N_out, N_in] outputs = 16 .. 64 tiles of 256 x 256, K = all tokens) ----
// Splits never straddle samples (the operands of a sample are contiguous in K, samples are a batch stride apart); a sample
// is cut further while that keeps the item count within one round of the 256 CUs and the partial planes at 8 or fewer.
// Measured on MI355X (tools/wgrad_bench.py, lpad 4224): at 4 samples every DiT shape gains (183 -> 130, 201 -> 147, 182 -> 144,
// 154 -> 64 us); at 1 sample only the 1024 x 1024 gradient does (41 -> 31 us) -- splits of fewer than 16 K units pay more
// for prologue, epilogue and the reduction than they win, unless the single-pass grid is tiny (<= 16 tiles).
int splitk_plan(int M, int N, int K, int k_per_batch, int* splits_per_batch) {
    if (M <= 0 || N <= 0 || k_per_batch <= 0 || M % 256 || N % 256 || k_per_batch % 128 || K % k_per_batch) return 0;
    const int nb = K / k_per_batch, tiles = (M / 256) * (N / 256), units = k_per_batch / 128;
    if (nb > 8) return 0;
    const bool tiny = tiles <= 16;
    int s = 256 / (nb * tiles);
    if (s > 8 / nb) s = 8 / nb;
    if (s > units / (tiny ? 4 : 16)) s = units / (tiny ? 4 : 16);
    if (s < 1) s = 1;
    const char* e = getenv("DGS_SPLITK_MIN_ITEMS");            // tests lower it to reach the path at emulator-sized shapes
    if (nb * s < 2 || nb * s * tiles < (e ? atoi(e) : tiny ? 96 : 160)) return 0;
    *splits_per_batch = s;
    return nb * s;
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int nsplit, int N, int ldo,
                                                            size_t plane) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;     // element of the [M, N] plane
    if (i >= plane) return;
    float4 acc = *reinterpret_cast<const float4*>(part + i);
    for (int s = 1; s < nsplit; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(part + s * plane + i);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    const size_t m = i / N, n = i - m * N;
    *reinterpret_cast<float4*>(out + m * ldo + n) = acc;
}

int launch_splitk_gemm(const DgsDitGemmArgs* a, int k_per_batch, hipStream_t st) {
    int spb = 1;
    const int nsplit = splitk_plan(a->M, a->N, a->K, k_per_batch, &spb);
    if (!nsplit || a->epilogue != DGS_EPI_F32 || a->bias || !a->splitk_ws || a->ldo % 4) return DGS_ERR_INVALID_ARGUMENT;
    DeepParams p;
    p.M = a->M; p.N = a->N; p.K = k_per_batch; p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->N; p.gate_stride = 0;
    p.rows_per_batch = a->M; p.valid_rows = a->M; p.dbg = 0; p.nsplit = nsplit; p.splits_per_batch = spb;
    p.a_batch_stride = a->a_batch_stride; p.w_batch_stride = a->w_batch_stride; p.out_split_stride = (long long)a->M * a->N;
    p.A = a->A; p.W = a->W; p.bias = nullptr; p.out = a->splitk_ws; p.gate = nullptr; p.vt = nullptr; p.aux = nullptr; p.q_scale = 1.0f;
    p.resid = nullptr; p.rows_external = 0;
    const int rc = launch_sliced<DGS_EPI_F32, 256>(p, st);
    if (rc != DGS_OK) return rc;
    const size_t plane = (size_t)a->M * a->N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((plane / 4 + 255) / 256)), dim3(256), 0, st, a->splitk_ws, static_cast<float*>(a->out), nsplit,
                       a->N, a->ldo, plane);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

}  // namespace dgs
