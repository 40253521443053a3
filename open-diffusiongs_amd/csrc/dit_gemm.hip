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
#include "dit_common.h"
#include "dgs_dit.h"

namespace dgs {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;   // 16 KiB per operand per stage

struct GemmParams {
    int M, N, K, lda, ldw, ldo, gate_stride, rows_per_batch, tiles_n, ntiles;
    const bf16_t* A;
    const bf16_t* W;
    const float* bias;
    void* out;
    const float* gate;
    bf16_t* vt;
};

__device__ __forceinline__ float gelu_tanh(float x) {
    // nn.GELU(approximate="tanh"): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) == x * sigmoid(2u)
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return x / (1.0f + __expf(-2.0f * u));
}

// Stage one [128][64] bf16 tile: 16 wave-instructions of 1 KiB (8 rows each); wave w issues instructions 4w..4w+3.
__device__ __forceinline__ void stage_tile(const bf16_t* g, int ld, int row0, int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int piece = wave * 4 + q;
        const int row = piece * 8 + (lane >> 3);
        const int slot = lane & 7;
        const int chunk = slot ^ ((row >> 1) & 7);
        const bf16_t* src = g + (size_t)(row0 + row) * ld + k0 + chunk * 8;
        glds16(src, lds_tile + piece * 1024);
    }
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) char lds[2 * 2 * TILE_BYTES];   // [stage][A,B] : 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int logical = xcd_remap((int)blockIdx.x, p.ntiles);
    const int tn = logical % p.tiles_n, tm = logical / p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = p.K / BK;
    stage_tile(p.A, p.lda, m0, 0, lds, wave, lane);
    stage_tile(p.W, p.ldw, n0, 0, lds + TILE_BYTES, wave, lane);
    __syncthreads();   // drains the DMA (vmcnt(0)) and publishes the tile

    // per-lane fragment addressing: row (lane & 31) of a 32-row block, k-chunk (lane >> 5) + 2*ks
    const int frow = lane & 31, fhalf = lane >> 5;
    const int swz = (frow >> 1) & 7;   // block row offsets are multiples of 32 -> do not change (row >> 1) & 7
    for (int t = 0; t < nk; ++t) {
        char* cur = lds + (t & 1) * 2 * TILE_BYTES;
        if (t + 1 < nk) {
            char* nxt = lds + ((t + 1) & 1) * 2 * TILE_BYTES;
            stage_tile(p.A, p.lda, m0, (t + 1) * BK, nxt, wave, lane);
            stage_tile(p.W, p.ldw, n0, (t + 1) * BK, nxt + TILE_BYTES, wave, lane);
        }
        const char* la = cur + (wm * 64 + frow) * 128;
        const char* lb = cur + TILE_BYTES + (wn * 64 + frow) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int off = ((2 * ks + fhalf) ^ swz) << 4;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(la + off);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(la + 32 * 128 + off);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(lb + off);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(lb + 32 * 128 + off);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();   // next tile landed; everyone is done reading `cur`
    }

    // ---- epilogue.  D fragment: col (n) = lane & 31, row (m) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) ----
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + wn * 64 + ni * 32 + (lane & 31);
        const float bias = p.bias ? p.bias[n] : 0.0f;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
            const int mbase = m0 + wm * 64 + mi * 32 + 4 * fhalf;
            if (EPI == DGS_EPI_QKV && n >= (p.N / 3) * 2) {
                // V^T: 4 consecutive tokens of one feature = one 8-byte store
                const int f = n - (p.N / 3) * 2;
                const int b = mbase / p.rows_per_batch;      // a 32-row block never straddles samples (lpad % 128 == 0)
                bf16_t* dst = p.vt + ((size_t)b * (p.N / 3) + f) * p.rows_per_batch + (mbase - b * p.rows_per_batch);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 v;
                    v.x = pack_bf2(acc[mi][ni][4 * g] + bias, acc[mi][ni][4 * g + 1] + bias);
                    v.y = pack_bf2(acc[mi][ni][4 * g + 2] + bias, acc[mi][ni][4 * g + 3] + bias);
                    *reinterpret_cast<uint2*>(dst + 8 * g) = v;
                }
                continue;
            }
            float gate = 0.0f;
            if (EPI == DGS_EPI_GATE_RESIDUAL) gate = p.gate[(size_t)(mbase / p.rows_per_batch) * p.gate_stride + n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                const float v = acc[mi][ni][r] + bias;
                const size_t o = (size_t)m * p.ldo + n;
                if (EPI == DGS_EPI_BF16 || EPI == DGS_EPI_QKV) reinterpret_cast<bf16_t*>(p.out)[o] = (bf16_t)f2bf(v);
                else if (EPI == DGS_EPI_GELU_BF16) reinterpret_cast<bf16_t*>(p.out)[o] = (bf16_t)f2bf(gelu_tanh(v));
                else if (EPI == DGS_EPI_GATE_RESIDUAL) { float* x = reinterpret_cast<float*>(p.out) + o; *x = *x + gate * v; }
                else reinterpret_cast<float*>(p.out)[o] = v;
            }
        }
    }
}

}  // namespace dgs

using namespace dgs;

extern "C" int dgs_dit_gemm(const DgsDitGemmArgs* a, dgs_stream_t stream) {
    if (!a || a->M <= 0 || a->N <= 0 || a->K <= 0 || a->M % BM || a->N % BN || a->K % BK) return DGS_ERR_INVALID_ARGUMENT;
    if (!a->A || !a->W || !a->out || a->lda < a->K || a->ldw < a->K || (a->lda & 7) || (a->ldw & 7)) return DGS_ERR_INVALID_ARGUMENT;
    if (a->epilogue == DGS_EPI_GATE_RESIDUAL && (!a->gate || a->rows_per_batch <= 0)) return DGS_ERR_INVALID_ARGUMENT;
    if (a->epilogue == DGS_EPI_QKV && (!a->vt || a->rows_per_batch <= 0 || a->rows_per_batch % BM || a->N % 3 || (a->N / 3) % BN))
        return DGS_ERR_INVALID_ARGUMENT;
    GemmParams p;
    p.M = a->M; p.N = a->N; p.K = a->K; p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->ldo;
    p.gate_stride = a->gate_stride; p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : a->M;
    p.tiles_n = a->N / BN; p.ntiles = p.tiles_n * (a->M / BM);
    p.A = a->A; p.W = a->W; p.bias = a->bias; p.out = a->out; p.gate = a->gate; p.vt = a->vt;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid(p.ntiles), block(256);
    switch (a->epilogue) {
        case DGS_EPI_BF16: hipLaunchKernelGGL((gemm_bf16_kernel<DGS_EPI_BF16>), grid, block, 0, st, p); break;
        case DGS_EPI_GELU_BF16: hipLaunchKernelGGL((gemm_bf16_kernel<DGS_EPI_GELU_BF16>), grid, block, 0, st, p); break;
        case DGS_EPI_GATE_RESIDUAL: hipLaunchKernelGGL((gemm_bf16_kernel<DGS_EPI_GATE_RESIDUAL>), grid, block, 0, st, p); break;
        case DGS_EPI_F32: hipLaunchKernelGGL((gemm_bf16_kernel<DGS_EPI_F32>), grid, block, 0, st, p); break;
        case DGS_EPI_QKV: hipLaunchKernelGGL((gemm_bf16_kernel<DGS_EPI_QKV>), grid, block, 0, st, p); break;
        default: return DGS_ERR_INVALID_ARGUMENT;
    }
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
