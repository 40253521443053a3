// dit_gemm.hip -- bf16 MFMA GEMM with fused epilogues for the DiT linears (gfx950, wave64).
//
//   C[M,N] = A[M,K] . W[N,K]^T   (A: activations bf16 row-major, W: nn.Linear weight bf16 [out,in]; fp32 accumulate)
//
// Replaces the reference's nn.Linear calls inside timm Attention / Mlp (utils_transformer.py:254-265), the tokenizer
// (denoiser.py:216-221) and the decoder head (denoiser.py:148-164); the elementwise ops that follow each Linear in the
// reference (bias, GELU-tanh :259, gate * y + residual :286-289, the q/k/v split + V transpose of timm Attention) are
// fused into the epilogue so no intermediate makes an extra HBM round trip.
//
// Structure (cdna_hip_programming.md section 5): 128x128 output tile per 256-thread workgroup (4 waves, 2x2, each
// 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 accumulators), BK = 64.  Both operands are K-contiguous, so both tiles are
// [128 rows][128 B] images staged with 16-byte LDS-DMA (global_load_lds_dwordx4), double-buffered.  The LDS image is
// lane-linear (DMA constraint), so the bank-conflict swizzle lives on the SOURCE address and on the ds_read_b128
// address (rule 21): 16-byte chunk c of row r sits in slot c ^ ((r >> 1) & 7).
#include <stdlib.h>

#include "dit_common.h"
#include "dgs_dit.h"
#include "dit_gemm_epilogue.h"

namespace dgs {

constexpr int BM = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB: A tile of one stage (the W tile is BN/128 of that)

struct GemmParams {
    int M, N, K, lda, ldw, ldo, gate_stride, rows_per_batch, valid_rows, tiles_n, ntiles, tiles_m, map_mode;
    int rows_ps, full_rows;          // per sample: 128-row tile rows, and how many of them hold a live 32-row block (a prefix)
    int ntail, tail_row0;            // GEMV items (one 32-column block of one sample each) for a last tile row with <= 2 live rows,
                                     // taken by the first `ntail` workgroups; tail_row0 = that row's first token (row in sample)
    int k_per_batch;                 // reduction elements per sample (K when the reduction dimension is not batched)
    long long a_batch_stride, w_batch_stride;   // element stride between samples along the reduction (weight-gradient GEMMs)
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    void* out;
    const float* gate;
    const float* resid;              // GATE_RESIDUAL input stream (== out for the in-place inference form)
    bf16_t* vt;                      // transposed bf16 copy [batch, N, rows_per_batch] (QKV: V only)
    void* aux;                       // GELU: u out; GATE_RESIDUAL: y out; DGELU: u in   (bf16 [M, ldo])
    float q_scale;                   // QKV: factor on the q features
};

__device__ __forceinline__ float gelu_tanh(float x) {
    // nn.GELU(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) == x * sigmoid(2u)
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x / (1.0f + __expf(-2.0f * u));
}

// d/dx of gelu_tanh: with s = sigmoid(2u), u = c (x + a x^3):  s + x s (1 - s) 2 c (1 + 3 a x^2)
__device__ __forceinline__ float dgelu_tanh(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    const float sg = 1.0f / (1.0f + __expf(-2.0f * u));
    return sg + x * sg * (1.0f - sg) * (2.0f * 0.7978845608028654f) * (1.0f + 3.0f * 0.044715f * x * x);
}

// Stage one [ROWS][64] bf16 tile: ROWS/8 wave-instructions of 1 KiB (8 rows each); wave w issues pieces (ROWS/32) w ...
template <int ROWS>
__device__ __forceinline__ void stage_tile(const bf16_t* g, int ld, int row0, int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < ROWS / 32; ++q) {
        const int piece = wave * (ROWS / 32) + q;
        const int row = piece * 8 + (lane >> 3);
        const int slot = lane & 7;
        const int chunk = slot ^ ((row >> 1) & 7);
        const bf16_t* src = g + (size_t)(row0 + row) * ld + k0 + chunk * 8;   // k0 already carries the sample offset
        glds16(src, lds_tile + piece * 1024);
    }
}

// One BK = 64 slab of a wave's accumulators: MI live 32-row blocks x NI 32-column blocks, 4 k-substeps of 16.
// Issue order (tools/ubench/issue_bench; hipcc waits with lgkmcnt(0) at the first use of a fragment, so a read issued
// right before an MFMA exposes the whole LDS latency): the fragments of substep ks + 1 are read into the other half of a
// register double buffer BEHIND the first MFMA of substep ks, i.e. MI * NI - 1 MFMAs before they are needed; only the
// first substep of a slab waits for its reads.  sched_barrier fences keep the source order.
template <int NI, int MI>
__device__ __forceinline__ void mma_tile(const char* la, const char* lb, int fhalf, int swz, f32x16 (&acc)[2][NI]) {
    bf16x8 a[2][MI], b[2][NI];
    auto read = [&](int ks, int h) {
        const int off = ((2 * ks + fhalf) ^ swz) << 4;
#pragma unroll
        for (int i = 0; i < MI; ++i) a[h][i] = *reinterpret_cast<const bf16x8*>(la + i * 32 * 128 + off);
#pragma unroll
        for (int j = 0; j < NI; ++j) b[h][j] = *reinterpret_cast<const bf16x8*>(lb + j * 32 * 128 + off);
    };
    read(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int h = ks & 1;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][i], b[h][j], acc[i][j], 0, 0, 0);
                if (i == 0 && j == 0 && ks < 3) read(ks + 1, h ^ 1);
                sched_fence();
            }
    }
}

// ONE_BATCH: the reduction does not cross samples (k_per_batch == K: every forward GEMM) -- the slab's sample index is then the
// literal 0 instead of an integer division per slab (~20 SALU instructions in the loop of every wave).
template <int BN, int NI, int MI, bool ONE_BATCH>
__device__ __forceinline__ void main_loop(const GemmParams& p, char* lds, int m0, int n0, int wave, int lane, int wm, int wn,
                                          f32x16 (&acc)[2][NI]) {
    constexpr int STAGE_BYTES = TILE_BYTES + BN * BK * 2;
    const int nk = p.K / BK;
    const int spb = p.k_per_batch / BK;                      // K slabs per sample
    stage_tile<BM>(p.A, p.lda, m0, 0, lds, wave, lane);
    stage_tile<BN>(p.W, p.ldw, n0, 0, lds + TILE_BYTES, wave, lane);
    __syncthreads();   // drains the DMA (vmcnt(0)) and publishes the tile
    // per-lane fragment addressing: row (lane & 31) of a 32-row block, k-chunk (lane >> 5) + 2*ks
    const int frow = lane & 31, fhalf = lane >> 5;
    const int swz = (frow >> 1) & 7;   // block row offsets are multiples of 32 -> do not change (row >> 1) & 7
    for (int t = 0; t < nk; ++t) {
        char* cur = lds + (t & 1) * STAGE_BYTES;
        if (t + 1 < nk) {
            char* nxt = lds + ((t + 1) & 1) * STAGE_BYTES;
            const int bb = ONE_BATCH ? 0 : (t + 1) / spb, kk = ((t + 1) - bb * spb) * BK;
            stage_tile<BM>(p.A + bb * p.a_batch_stride, p.lda, m0, kk, nxt, wave, lane);
            stage_tile<BN>(p.W + bb * p.w_batch_stride, p.ldw, n0, kk, nxt + TILE_BYTES, wave, lane);
        }
        if (MI > 0) {
            const char* la = cur + (wm * 64 + frow) * 128;
            const char* lb = cur + TILE_BYTES + (wn * (BN / 2) + frow) * 128;
            mma_tile<NI, (MI > 0 ? MI : 1)>(la, lb, fhalf, swz, acc);
        }
        __syncthreads();   // next tile landed; everyone is done reading `cur`
    }
}

// The <= 2 live rows behind a sample's last full 128-row tile (the DiT's two learned tokens: L = 4098 = 32 x 128 + 2) as
// a GEMV on the vector pipe.  As a 33rd tile row these two rows cost a whole extra tile per tile column -- 528 tiles for
// 512 two-per-CU slots: the N = 1024 GEMMs took 71 us instead of 52 (fc2) and 24 instead of 20 (proj).  An item = 8 output
// columns of one sample, a small workgroup BEHIND the tiles in the grid: the 512 tiles are placed first, two per CU, and the
// items take the third (LDS-limited) slot of some CUs right away.  (Items placed first shifted the tile placement: some CUs got
// three tiles, +10 us.)  While the chip
// streams GEMM tiles an L2 round trip costs microseconds, so an item is ONE trip: the four waves split K, every wave
// issues all of its loads (8 columns x its K quarter, 16 bytes per lane) before the first v_dot2c_f32_bf16; the partial
// sums meet in LDS and 16 lanes apply the epilogue element-wise (same arithmetic as store_strip).
template <int EPI>
__device__ __forceinline__ void gemv_tail_rows(const GemmParams& p, char* lds, int item, int wave, int lane) {
    constexpr int CPI = 8;                                   // columns per item
    const int nblk = p.N / CPI, b = item / nblk, tn0 = (item - b * nblk) * CPI;
    const int tm0 = b * p.rows_per_batch + p.tail_row0;
    const int kw = p.K / 4, k_lo = wave * kw;                // this wave's K quarter (K % 512 == 0: a multiple of 128)
    const int nch = (kw + 511) / 512;                        // 512-element chunks (64 lanes x 8), at most 2 (K <= 4096)
    const bool on0 = lane * 8 < kw, on1 = 512 + lane * 8 < kw;
    const bf16_t* a_row0 = p.A + (size_t)tm0 * p.lda + k_lo + lane * 8;
    const bf16_t* w_col0 = p.W + (size_t)tn0 * p.ldw + k_lo + lane * 8;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    const int er = lane / CPI, ec = lane - er * CPI;         // the element lane `lane` of wave 0 finishes (lanes 0 .. 2 CPI - 1)
    const bool finisher = wave == 0 && lane < 2 * CPI && p.tail_row0 + er < p.valid_rows;
    TailOperands ops{0.f, 0.f, 0.f};
    if (finisher) ops = tail_prefetch<EPI>(p, tm0 + er, tn0 + ec);
    uint4 a[2][2], w[CPI][2];
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
        const bool on = ch == 0 ? on0 : (on1 && nch > 1);
#pragma unroll
        for (int r = 0; r < 2; ++r) a[r][ch] = on ? *reinterpret_cast<const uint4*>(a_row0 + (size_t)r * p.lda + ch * 512) : zero;
#pragma unroll
        for (int c = 0; c < CPI; ++c) w[c][ch] = on ? *reinterpret_cast<const uint4*>(w_col0 + (size_t)c * p.ldw + ch * 512) : zero;
    }
    float* const part = reinterpret_cast<float*>(lds);       // [wave][row][column]
#pragma unroll
    for (int c = 0; c < CPI; ++c) {
        float s0 = dot8_bf16(a[0][0], w[c][0], 0.f), s1 = dot8_bf16(a[1][0], w[c][0], 0.f);
        s0 = dot8_bf16(a[0][1], w[c][1], s0); s1 = dot8_bf16(a[1][1], w[c][1], s1);
        s0 = wave_sum_lane63(s0); s1 = wave_sum_lane63(s1);
        if (lane == 63) { part[(wave * 2 + 0) * CPI + c] = s0; part[(wave * 2 + 1) * CPI + c] = s1; }
    }
    __syncthreads();
    if (finisher) {
        const int r = er, c = ec;
        const float v = (part[(0 * 2 + r) * CPI + c] + part[(1 * 2 + r) * CPI + c]) + (part[(2 * 2 + r) * CPI + c] + part[(3 * 2 + r) * CPI + c]);
        tail_store<EPI>(p, tm0 + r, tn0 + c, v, ops);
    }
}

// BN = 128: waves 2(M) x 2(N), each 64 x 64 (2 x 2 accumulators).  BN = 64 (used when N / 128 tiles would not fill the
// chip, e.g. the N = 1024 projections at batch 1): waves 2 x 2, each 64 x 32 (2 x 1 accumulators), half the LDS.
template <int EPI, int BN>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    constexpr int NI = BN / 64;                              // 32-column accumulator blocks per wave
    constexpr int STAGE_BYTES = TILE_BYTES + BN * BK * 2;
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE_BYTES];   // [stage][A | W] : 64 KiB (BN=128) / 48 KiB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: live0/live1 become scalar branches
    const int wm = wave >> 1, wn = wave & 1;
    // Work items: the tile rows of every sample that hold at least one live 32-row block (rows made only of padding are not
    // launched).  A direct, unstaged path for the rows with a single live block -- the two learned-token rows; what the sliced
    // kernel does -- was measured here and lost: +4..7 us on the N = 1024 GEMMs at batch 1.  Such a block is a chain of L2 round trips
    // (4 waves cannot hold K = 4096 worth of fragments in flight); inside the grid a half-empty tile costs less than that because
    // it overlaps with its CU's other workgroup.
    if ((int)blockIdx.x >= p.ntiles) {                           // block-uniform: the GEMV items sit BEHIND the tiles in the grid
        gemv_tail_rows<EPI>(p, lds, (int)blockIdx.x - p.ntiles, wave, lane);
        return;
    }
    const int logical = xcd_remap((int)blockIdx.x, p.ntiles);
    // map_mode 0: an XCD's contiguous id range walks tn fastest (A row panels stay in that XCD's L2, W streams through);
    // map_mode 1: tm fastest (a W column panel stays resident, A streams through)
    int tn, tr;                                                  // tr: index among the full tile rows of all samples
    if (p.map_mode >= 2) {
        // grouped order: ids walk GM = map_mode tile rows (tr fastest) before moving to the next tile column, so the ~64
        // tiles an XCD runs at once form a compact GM x (64 / GM) block whose A and W panels fit that XCD's 4 MiB L2
        const int gsz = p.map_mode * p.tiles_n, grp = logical / gsz, in = logical - grp * gsz;
        const int first = grp * p.map_mode, gm = min(p.tiles_m - first, p.map_mode);
        tr = first + in % gm;
        tn = in / gm;
    } else {
        tn = p.map_mode ? logical / p.tiles_m : logical % p.tiles_n;
        tr = p.map_mode ? logical % p.tiles_m : logical / p.tiles_n;
    }
    const int tm = (tr / p.full_rows) * p.rows_ps + tr % p.full_rows;
    const int m0 = tm * BM, n0 = tn * BN;
    // Padding rows (row-in-sample >= valid_rows) are never observed: a 32-row accumulator block made only of padding
    // skips its MFMAs and its stores (its output rows keep the finite values they had), which makes the one
    // mostly-padding M tile of every sample cost a fraction of a full tile.
    const int mrow = m0 - (m0 / p.rows_per_batch) * p.rows_per_batch + wm * 64;
    const bool live0 = mrow < p.valid_rows, live1 = mrow + 32 < p.valid_rows;

    f32x16 acc[2][NI];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // The K loop exists in three branch-free copies selected per wave (2 / 1 / 0 live 32-row blocks); every copy
    // stages and synchronises identically.  A branch INSIDE the loop makes hipcc carry the accumulators in VGPRs and copy
    // them to AGPRs and back around every slab.
    if (p.k_per_batch == p.K) {
        if (live1) main_loop<BN, NI, 2, true>(p, lds, m0, n0, wave, lane, wm, wn, acc);
        else if (live0) main_loop<BN, NI, 1, true>(p, lds, m0, n0, wave, lane, wm, wn, acc);
        else main_loop<BN, NI, 0, true>(p, lds, m0, n0, wave, lane, wm, wn, acc);
    } else {
        if (live1) main_loop<BN, NI, 2, false>(p, lds, m0, n0, wave, lane, wm, wn, acc);
        else if (live0) main_loop<BN, NI, 1, false>(p, lds, m0, n0, wave, lane, wm, wn, acc);
        else main_loop<BN, NI, 0, false>(p, lds, m0, n0, wave, lane, wm, wn, acc);
    }
    const int fhalf = lane >> 5;

    // ---- epilogue, staged through LDS (dit_gemm_epilogue.h) for everything the inference sequence launches; the stages are
    //      idle (main_loop ends with a barrier), every wave takes a private patch ----
    if (epi_staged<EPI>(p)) {
        char* patch = lds + wave * epi_strip_bytes(NI);
        if constexpr (EPI == DGS_EPI_GATE_RESIDUAL) {
            // both strips' residual values in ONE round trip: out may alias resid, so the second strip's loads cannot move above
            // the first strip's stores by themselves
            float4 pre[2][4 * NI];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                if (mi == 0 ? live0 : live1) residual_prefetch<NI>(p, m0 + wm * 64 + mi * 32, n0 + wn * (BN / 2), lane, pre[mi]);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                if (mi == 0 ? live0 : live1) store_strip<EPI, NI>(p, acc[mi], m0 + wm * 64 + mi * 32, n0 + wn * (BN / 2), lane, patch, pre[mi]);
            return;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
            if (mi == 0 ? live0 : live1) store_strip<EPI, NI>(p, acc[mi], m0 + wm * 64 + mi * 32, n0 + wn * (BN / 2), lane, patch);
        return;
    }

    // ---- epilogue (training variants).  D fragment: col (n) = lane & 31, row (m) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): a lane holds, for ONE
    //      output feature, four groups of four consecutive rows -> row-major stores are 2/4-byte per row, the optional
    //      transposed copy ([batch, N, rows_per_batch], wanted by the attention and weight-gradient kernels) is one 8-byte
    //      store per group.
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int n = n0 + wn * (BN / 2) + ni * 32 + (lane & 31);
        const float bias = p.bias ? p.bias[n] : 0.0f;
        const float qs = (EPI == DGS_EPI_QKV && n < p.N / 3) ? p.q_scale : 1.0f;      // pre-scaled queries
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            if (!(mi == 0 ? live0 : live1)) continue;
            const int mbase = m0 + wm * 64 + mi * 32 + 4 * fhalf;
            const int b = mbase / p.rows_per_batch;          // a 32-row block never straddles samples (rows_per_batch % 128 == 0)
            const bool qkv_v = EPI == DGS_EPI_QKV && n >= (p.N / 3) * 2;
            bf16_t* tdst = nullptr;                           // transposed destination of this lane's feature
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
                    const float v = EPI == DGS_EPI_QKV ? (acc[mi][ni][4 * g + q] + bias) * qs : acc[mi][ni][4 * g + q] + bias;
                    const size_t o = (size_t)m * p.ldo + n;
                    if (EPI == DGS_EPI_BF16) {
                        o4[q] = v;
                        reinterpret_cast<bf16_t*>(p.out)[o] = (bf16_t)f2bf_fast(v);
                    } else if (EPI == DGS_EPI_QKV) {
                        o4[q] = v;
                        if (!qkv_v) reinterpret_cast<bf16_t*>(p.out)[o] = (bf16_t)f2bf_fast(v);
                    } else if (EPI == DGS_EPI_GELU_BF16) {
                        o4[q] = gelu_tanh(v);
                        reinterpret_cast<bf16_t*>(p.out)[o] = (bf16_t)f2bf_fast(o4[q]);
                        if (p.aux) reinterpret_cast<bf16_t*>(p.aux)[o] = (bf16_t)f2bf_fast(v);
                    } else if (EPI == DGS_EPI_DGELU_BF16) {
                        o4[q] = v * dgelu_tanh(bf2f(reinterpret_cast<const bf16_t*>(p.aux)[o]));
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
}

int sliced_gemm_tile(int M, int N, int K, int epilogue, int k_per_batch, int rows_per_batch, int valid_rows);   // dit_gemm_deep.hip
int launch_sliced_gemm(const DgsDitGemmArgs* a, int bn, int rows_per_batch, int valid_rows, hipStream_t st, bool quad, bool rows_external);
bool sliced_rows_are_gemv(int K, int N, int valid_rows);
bool sliced128_eligible(int M, int N, int K, int epilogue, int k_per_batch, int rows_per_batch);
int splitk_plan(int M, int N, int K, int k_per_batch, int* splits_per_batch);
int launch_splitk_gemm(const DgsDitGemmArgs* a, int k_per_batch, hipStream_t st);

}  // namespace dgs

using namespace dgs;

template <int EPI>
static void launch_gemm(const GemmParams& p0, int bn, hipStream_t st) {
    GemmParams p = p0;
    p.tiles_n = p.N / bn;
    p.rows_ps = p.rows_per_batch / BM;
    p.full_rows = (p.valid_rows + BM - 1) / BM;               // valid rows are a prefix of every sample
    // a last tile row with one or two live rows is not a tile row: its 32-column blocks are GEMV items (gemv_tail_rows)
    const int last_live = p.valid_rows - (p.full_rows - 1) * BM;
    static const int no_tail = getenv("DGS_GEMM_NO_GEMV_TAIL") ? atoi(getenv("DGS_GEMM_NO_GEMV_TAIL")) : 0;   // measurement aid
    p.ntail = 0; p.tail_row0 = 0;
    if (!no_tail && p.full_rows > 1 && last_live <= 2 && p.K % 512 == 0 && p.K <= 4096 && p.k_per_batch == p.K && p.N % 8 == 0 &&
        !(EPI == DGS_EPI_QKV)) {
        --p.full_rows;
        p.tail_row0 = p.full_rows * BM;
        p.ntail = (p.M / p.rows_per_batch) * (p.N / 8);
    }
    p.tiles_m = (p.M / p.rows_per_batch) * p.full_rows;
    p.ntiles = p.tiles_m * p.tiles_n;
    static const int map_env = getenv("DGS_GEMM_MAP") ? atoi(getenv("DGS_GEMM_MAP")) : 0;
    p.map_mode = map_env;
    if (bn == 128) hipLaunchKernelGGL((gemm_bf16_kernel<EPI, 128>), dim3(p.ntail + p.ntiles), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((gemm_bf16_kernel<EPI, 64>), dim3(p.ntail + p.ntiles), dim3(256), 0, st, p);
}

// mode 3: no launch, the tile width of the sliced 256-row kernel this call runs on (256 / 192 / 128), 0 for the other kernels.
// mode 0: dgs_dit_gemm.  mode 1: the same launch without the two-row GEMV side jobs of the sliced 256-row kernel (the caller
// produces those rows: layernorm_rows_gemv_kernel); an error if the shape does not run there.  mode 2: no launch, 1 if mode 1 applies.
static int gemm_dispatch(const DgsDitGemmArgs* a, dgs_stream_t stream, int mode) {
    if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->M % BM || a->N % 64 || a->K % BK) return DGS_ERR_INVALID_ARGUMENT;
    if (!a->A || !a->W || !a->out || (a->lda & 7) || (a->ldw & 7)) return DGS_ERR_INVALID_ARGUMENT;
    const int kpb = a->k_per_batch > 0 ? a->k_per_batch : a->K;
    if (kpb % BK || a->K % kpb || a->lda < kpb || a->ldw < kpb) return DGS_ERR_INVALID_ARGUMENT;
    if (a->epilogue == DGS_EPI_GATE_RESIDUAL && (!a->gate || a->rows_per_batch <= 0)) return DGS_ERR_INVALID_ARGUMENT;
    if (a->epilogue == DGS_EPI_QKV && (!a->vt || a->N % 3 || (a->N / 3) % 128)) return DGS_ERR_INVALID_ARGUMENT;
    if (a->epilogue == DGS_EPI_DGELU_BF16 && !a->aux) return DGS_ERR_INVALID_ARGUMENT;
    if ((a->vt || a->epilogue == DGS_EPI_QKV) && a->rows_per_batch <= 0) return DGS_ERR_INVALID_ARGUMENT;
    if (a->rows_per_batch > 0 && (a->rows_per_batch % BM || a->M % a->rows_per_batch)) return DGS_ERR_INVALID_ARGUMENT;
    GemmParams p;
    p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->ldo;
    p.gate_stride = a->gate_stride; p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : a->M;
    p.valid_rows = (a->valid_rows > 0 && a->valid_rows < p.rows_per_batch) ? a->valid_rows : p.rows_per_batch;
    p.k_per_batch = kpb; p.a_batch_stride = a->a_batch_stride; p.w_batch_stride = a->w_batch_stride;
    p.A = a->A; p.W = a->W; p.bias = a->bias; p.out = a->out; p.gate = a->gate; p.vt = a->vt; p.aux = a->aux;
    p.q_scale = a->q_scale != 0.0f ? a->q_scale : 1.0f;
    p.resid = a->resid ? a->resid : static_cast<const float*>(a->out);
    hipStream_t st0 = static_cast<hipStream_t>(stream);
    // Kernel choice.  The default is the 128-wide two-stage kernel below; the sliced 256-row kernel (dit_gemm_deep.hip) takes over
    // where its tile count fits the chip (see AUTO).
    static const int env_algo = getenv("DGS_GEMM_ALGO") ? atoi(getenv("DGS_GEMM_ALGO")) : 0;
    const int algo = a->algo ? a->algo : env_algo;
    // weight-gradient shapes with a scratch buffer: split-K on the sliced kernel (any algo but an explicit SIMPLE128)
    if (mode == 0 && (algo == DGS_GEMM_AUTO || algo == DGS_GEMM_SLICED) && a->splitk_ws && a->epilogue == DGS_EPI_F32 && !a->bias && a->ldo % 4 == 0 &&
        p.rows_per_batch == a->M && p.valid_rows == a->M) {
        int spb = 0;
        if (splitk_plan(a->M, a->N, a->K, kpb, &spb)) return launch_splitk_gemm(a, kpb, st0);
    }
    // AUTO (measured on MI355X, tools/gemm_check.py; 128-wide kernel -> sliced kernel):
    //   1 sample  (M = 4352):  QKV 41 -> 39 us, fc1 + GELU 58 -> 44 us on 256 x 256 tiles (one round of the chip); the N = 1024
    //                          GEMMs (64 tiles of 256 x 256) stay on the 128-wide kernel (fc2 67, proj 23 us)
    //   4 samples (M = 17408): QKV 137 -> 119, fc2 219 -> 144, proj 68 -> 54, fc1 172 -> 178, f32 230 -> 171 us: everything eligible
    const int sbn_auto = algo == DGS_GEMM_AUTO ? sliced_gemm_tile(a->M, a->N, a->K, a->epilogue, kpb, p.rows_per_batch, p.valid_rows) : 0;
    const bool auto_sliced = sbn_auto != 0 && (a->M > 8192 || ((sbn_auto == 256 || sbn_auto == 192) && (a->epilogue == DGS_EPI_QKV ||
                                                                                  (a->epilogue == DGS_EPI_GELU_BF16 && a->N >= 4096))));
    if (algo == DGS_GEMM_SLICED || algo == DGS_GEMM_QUAD || auto_sliced) {
        const int sbn = sliced_gemm_tile(a->M, a->N, a->K, a->epilogue, kpb, p.rows_per_batch, p.valid_rows);
        if (mode == 3) return sbn;
        if (mode == 2) return sbn && sliced_rows_are_gemv(a->K, a->N, p.valid_rows) && p.valid_rows < p.rows_per_batch ? 1 : 0;
        if (sbn) return launch_sliced_gemm(a, sbn, p.rows_per_batch, p.valid_rows, st0, algo == DGS_GEMM_QUAD, mode == 1);
    }
    if (mode == 2 || mode == 3) return 0;
    if (mode == 1) return DGS_ERR_INVALID_ARGUMENT;
    // few tiles (the N = 1024 GEMMs at one sample: fc2, K = 4096, and proj, K = 1024): 128 x 128 tiles on the sliced kernel's ring, one
    // per CU, eight waves as two K groups (dit_gemm_deep.hip KW).  With FOUR waves (one per SIMD: rounds 3-5) fc2 ran 44.8 us and proj
    // lost to the 128-wide kernel below, 21 vs 23 us; with the K groups fc2 is 39.9 us and proj 16.4-18.0 against 19.1-20.8 us
    // (profiles/r06_gemm_kgroups_ab.txt).  DGS_GEMM_S128_MINK: measurement aid (2048: proj back on the 128-wide kernel)
    static const int no_s128 = getenv("DGS_GEMM_NO_SLICED128") ? atoi(getenv("DGS_GEMM_NO_SLICED128")) : 0;   // measurement aid
    static const int s128_mink = getenv("DGS_GEMM_S128_MINK") ? atoi(getenv("DGS_GEMM_S128_MINK")) : 1024;
    const bool few_tiles = (a->M / BM) * (a->N / 128) < 512 && a->N % 128 == 0;
    if ((algo == DGS_GEMM_SLICED128 || (algo == DGS_GEMM_AUTO && few_tiles && !no_s128 && a->K >= s128_mink)) &&
        sliced128_eligible(a->M, a->N, a->K, a->epilogue, kpb, p.rows_per_batch))
        return launch_sliced_gemm(a, -128, p.rows_per_batch, p.valid_rows, st0, false, false);
    // 128 x 64 tiles when 128 x 128 would leave the 256 CUs with fewer than two workgroups each
    if (a->epilogue == DGS_EPI_QKV && a->N % 128) return DGS_ERR_INVALID_ARGUMENT;
    static const int env_bn = getenv("DGS_GEMM_BN") ? atoi(getenv("DGS_GEMM_BN")) : 0;      // measurement aid
    int bn = (a->N % 128 || ((a->M / BM) * (a->N / 128) < 512 && a->epilogue != DGS_EPI_QKV)) ? 64 : 128;
    if (env_bn == 128 && a->N % 128 == 0) bn = 128;
    if (env_bn == 64) bn = 64;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (a->epilogue) {
        case DGS_EPI_BF16: launch_gemm<DGS_EPI_BF16>(p, bn, st); break;
        case DGS_EPI_GELU_BF16: launch_gemm<DGS_EPI_GELU_BF16>(p, bn, st); break;
        case DGS_EPI_GATE_RESIDUAL: launch_gemm<DGS_EPI_GATE_RESIDUAL>(p, bn, st); break;
        case DGS_EPI_F32: launch_gemm<DGS_EPI_F32>(p, bn, st); break;
        case DGS_EPI_QKV: launch_gemm<DGS_EPI_QKV>(p, 128, st); break;
        case DGS_EPI_DGELU_BF16: launch_gemm<DGS_EPI_DGELU_BF16>(p, bn, st); break;
        default: return DGS_ERR_INVALID_ARGUMENT;
    }
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

extern "C" int dgs_dit_gemm(const DgsDitGemmArgs* a, dgs_stream_t stream) { return gemm_dispatch(a, stream, 0); }
extern "C" int32_t dgs_dit_gemm_sliced_tile(const DgsDitGemmArgs* a) { const int w = gemm_dispatch(a, nullptr, 3); return w > 0 ? w : 0; }

namespace dgs {
bool gemm_leaves_rows_out(const DgsDitGemmArgs* a) { return gemm_dispatch(a, nullptr, 2) == 1; }
int launch_gemm_external_rows(const DgsDitGemmArgs* a, dgs_stream_t stream) { return gemm_dispatch(a, stream, 1); }
}  // namespace dgs

extern "C" size_t dgs_dit_gemm_splitk_bytes(int32_t M, int32_t N, int32_t K, int32_t k_per_batch) {
    int spb = 0;
    const int nsplit = dgs::splitk_plan(M, N, K, k_per_batch > 0 ? k_per_batch : K, &spb);
    return (size_t)nsplit * M * N * sizeof(float);
}

