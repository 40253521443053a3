// dit_tail_rows.hip -- the DiT's learned-token rows as a chain of their own (inference).  AN EXPERIMENT THAT LOST: compiled into the
// tools' library and the emulator build only (DGS_EXPERIMENTS), selected there with DGS_TAIL_CHAIN=1|2; the product keeps the rows as
// side jobs of the main kernels.  Measured (profiles/r05_tail_chain_ab.txt): the step 6.15 -> 6.70 ms with the GEMM / LN rows on the
// side stream (the main GEMMs do not get faster without their side jobs -- fc1 42.1, fc2 46.0 us -- and run beside four 17 us
// launches per block), 6.16 -> 7.37 ms with the tail queries there too (attention 81 -> 73 us, but the one-workgroup-per-head tail
// kernel streams 1 MiB per CU for 77 us beside proj / LN2 / fc1 and displaces their workgroups: fc1 40 -> 62 us).  What prompted it:
//
// A sample has L = 2 + 4 (res / 8)^2 tokens: 16 (or 64) full 256-row tiles of image tokens and TWO rows behind them.  Inside the
// one-round GEMM kernels those two rows were "side jobs" of the tile workgroups (a two-row GEMV per 32 columns, dit_gemm_deep.hip):
// the in-kernel timeline of all 256 workgroups (profiles/r05_gemm_timeline.txt) shows what that costs -- the workgroups that carry
// an item enter their K loop 3.3 us (fc1), 2-4.7 us (fc2), 1.8 us (proj) behind the others and are the launch's tail, because a
// 2 x N GEMV reads all of W once more (80 KiB per item through the CU's 64 B/clk vector-memory path, as much as the ring's first
// three slabs) -- and the attention kernel pays ~9 of 80 us for the two extra queries' merge chain at its very end.
//
// Here the two rows take the block's operators through kernels of their own, launched on a SECOND stream (csrc/dit_forward.hip
// run_blocks): per block  attention(tail queries) -> proj -> LN2 + fc1 -> fc2 -> LN1' + QKV'  for 2 rows, while the main stream
// runs the same operators over the 4,096 image-token rows with nothing bolted on.  The kernels are built to run BESIDE the main
// kernels, not instead of them: <= 32 VGPRs (the sliced GEMMs leave 512 - 2 x 240 registers per SIMD lane), <= 16 KiB of LDS (they
// leave 32), 256-thread workgroups.  tools/coreside_run.py / profiles/r05_coreside_probe.txt: such a kernel streaming a whole
// weight matrix beside fc1 costs fc1 +0.7 us and beside QKV nothing.  (Round 4's attempt to move only the attention tail failed on
// exactly this point: 512-thread workgroups of 101 VGPRs took whole CUs first and the main workgroups they displaced ran as a second
// round, profiles/r04_attention_tail_stream_ab.txt.)
//
// tail_rows_kernel: out[r, n] = epilogue( sum_k A[r, k] W[n, k] + bias[n] ) for R <= 4 rows per sample,
//   A = the rows themselves (bf16), or LayerNorm(x rows, eps) * (1 + scale) + shift rounded to bf16 -- what layernorm_kernel hands
//   the main GEMM -- staged ONCE per workgroup in LDS; a wave owns whole output columns (lanes span K, 16-byte loads, v_dot2c), so a
//   weight byte is read exactly once; epilogues as in dit_gemm_epilogue.h (tail_store): QKV split (pre-scaled q | k | V^T), GELU,
//   gated residual.  Reference semantics: utils_transformer.py:246-290 (DiTBlock), timm Attention.qkv / proj, Mlp.fc1 / fc2.
#include "dit_kernels.h"
#include "dit_gemm_epilogue.h"

#ifdef DGS_EXPERIMENTS

namespace dgs {


// 32 registers: what a SIMD lane has left beside two waves of the 240-register sliced GEMMs (the kernel has to fit BESIDE them)
#ifndef HIPEMU
#define DGS_TAIL_VGPRS __attribute__((amdgpu_num_vgpr(32)))
#else
#define DGS_TAIL_VGPRS
#endif
template <int R>
__global__ __launch_bounds__(256) DGS_TAIL_VGPRS void tail_rows_kernel(TailRowsParams p) {
    DGS_DYNAMIC_LDS(smem);
    bf16_t* As = reinterpret_cast<bf16_t*>(smem);                 // [R][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const size_t row_first = (size_t)b * p.lpad + p.row0;
    // ---- the R operand rows into LDS ----
    if (p.x) {
        // LayerNorm + modulate, one wave per row, two-pass statistics like layernorm_kernel (mean, then centred variance); the row is
        // re-read from L2 per pass instead of kept in registers (the register budget of a kernel that has to fit beside the GEMMs)
        for (int r = wave; r < R; r += 4) {
            const float4* xr = reinterpret_cast<const float4*>(p.x + (row_first + r) * (size_t)p.K);
            const int n4 = p.K / 4;
            float sum = 0.f;
#pragma unroll 1
            for (int i = lane; i < n4; i += 64) { const float4 v = xr[i]; sum += (v.x + v.y) + (v.z + v.w); }
            const float mean = wave_sum(sum) / (float)p.K;
            float sq = 0.f;
#pragma unroll 1
            for (int i = lane; i < n4; i += 64) {
                float4 v = xr[i];
                v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
                sq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
            const float rstd = rsqrtf(wave_sum(sq) / (float)p.K + p.eps);
            const float4* sh = reinterpret_cast<const float4*>(p.shift + (size_t)b * p.mod_stride);
            const float4* sc = reinterpret_cast<const float4*>(p.scale + (size_t)b * p.mod_stride);
#pragma unroll 1
            for (int i = lane; i < n4; i += 64) {
                const float4 v = xr[i], s = sh[i], c = sc[i];
                const float y0 = ((v.x - mean) * rstd) * (1.0f + c.x) + s.x, y1 = ((v.y - mean) * rstd) * (1.0f + c.y) + s.y;
                const float y2 = ((v.z - mean) * rstd) * (1.0f + c.z) + s.z, y3 = ((v.w - mean) * rstd) * (1.0f + c.w) + s.w;
                reinterpret_cast<uint2*>(As + (size_t)r * p.K)[i] = make_uint2(pack_bf2(y0, y1), pack_bf2(y2, y3));
            }
        }
    } else {
        const int n8 = p.K / 8;
        for (int i = tid; i < R * n8; i += 256) {
            const int r = i / n8, c = i - r * n8;
            reinterpret_cast<uint4*>(As + (size_t)r * p.K)[c] = reinterpret_cast<const uint4*>(p.A + (row_first + r) * (size_t)p.lda)[c];
        }
    }
    __syncthreads();
    // ---- a wave owns `cols_per_wave` output columns; lanes span K in 512-element chunks (16 bytes of W per lane and chunk) ----
    const int n_first = ((int)blockIdx.x * 4 + wave) * p.cols_per_wave;
    for (int c = 0; c < p.cols_per_wave; ++c) {
        const int n = n_first + c;
        if (n >= p.N) break;
        const bf16_t* wr = p.W + (size_t)n * p.ldw + lane * 8;
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
#pragma unroll 1
        for (int k = lane * 8; k < p.K; k += 512) {               // one 16-byte piece in flight: the register budget
            const uint4 w0 = *reinterpret_cast<const uint4*>(wr + k - lane * 8);
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = dot8_bf16(*reinterpret_cast<const uint4*>(As + (size_t)r * p.K + k), w0, acc[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = wave_sum_lane63(acc[r]);
        if (lane == 63) {
            const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const size_t m = row_first + r;
                const float v = acc[r] + bias;
                if (p.epilogue == DGS_EPI_QKV) {
                    const int third = p.N / 3;
                    if (n >= 2 * third) p.vt[((size_t)b * third + (n - 2 * third)) * p.lpad + (p.row0 + r)] = (bf16_t)(pack_bf2(v, 0.f) & 0xffffu);
                    else reinterpret_cast<bf16_t*>(p.out)[m * p.ldo + n] = (bf16_t)(pack_bf2(n < third ? v * p.q_scale : v, 0.f) & 0xffffu);
                } else if (p.epilogue == DGS_EPI_GELU_BF16) {
                    reinterpret_cast<bf16_t*>(p.out)[m * p.ldo + n] = (bf16_t)(pack_bf2(epi_gelu_tanh(v), 0.f) & 0xffffu);
                } else {                                          // DGS_EPI_GATE_RESIDUAL: x += gate * y on the fp32 residual stream
                    float* o = reinterpret_cast<float*>(p.out) + m * p.ldo + n;
                    *o = *o + p.gate[(size_t)b * p.gate_stride + n] * v;
                }
            }
        }
    }
}

int launch_tail_rows(const TailRowsParams& p, hipStream_t st) {
    if (p.B <= 0 || p.R < 1 || p.R > 4 || p.N <= 0 || p.K <= 0 || p.K % 8 || p.K > 4096 || (p.ldw & 7) || !p.W || !p.out) return DGS_ERR_INVALID_ARGUMENT;
    if ((p.x == nullptr) == (p.A == nullptr) || (p.x && (!p.shift || !p.scale)) || (p.A && (p.lda & 7))) return DGS_ERR_INVALID_ARGUMENT;
    if (p.epilogue == DGS_EPI_QKV ? (p.N % 3 || !p.vt) : p.epilogue == DGS_EPI_GATE_RESIDUAL ? !p.gate : p.epilogue != DGS_EPI_GELU_BF16) return DGS_ERR_INVALID_ARGUMENT;
    TailRowsParams q = p;
    q.cols_per_wave = p.N > 2048 ? 2 : 1;                         // ~2,000 waves a launch: every free wave slot of the chip gets one
    const int per_wg = 4 * q.cols_per_wave;
    const dim3 grid((p.N + per_wg - 1) / per_wg, p.B), block(256);
    const size_t lds = (size_t)p.R * p.K * sizeof(bf16_t);
    switch (p.R) {
        case 1: hipLaunchKernelGGL(tail_rows_kernel<1>, grid, block, lds, st, q); break;
        case 2: hipLaunchKernelGGL(tail_rows_kernel<2>, grid, block, lds, st, q); break;
        case 3: hipLaunchKernelGGL(tail_rows_kernel<3>, grid, block, lds, st, q); break;
        default: hipLaunchKernelGGL(tail_rows_kernel<4>, grid, block, lds, st, q); break;
    }
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

}  // namespace dgs

#endif  // DGS_EXPERIMENTS
