// dit_backward_elementwise.hip -- the HBM-bound kernels of the DiT backward pass (gfx950, wave64).
//
// The reference gets these from torch autograd (LayerNorm / modulate / gated residual / Linear bias / SiLU-Linear
// backward nodes of DiTBlock, utils_transformer.py:271-290, and of the heads, denoiser.py:122-164); here each is one pass
// over its operands:
//   transpose_kernel            bf16 [B*lpad, F] -> [B, F, lpad]: token-contiguous operands for the weight-gradient GEMMs
//   gate_mul_kernel             dy = gate * dx (bf16, row-major + transposed)  and  dgate += sum_t dx * y
//   layernorm_backward_kernel   dx += LN'(dh (1+scale) w);  dshift += sum_t dh;  dscale += sum_t dh * n w;  dw += sum dh (1+scale) n
//   colsum_kernel               bias gradients  db[n] = sum_m dY[m, n]
//   rowlinear_backward_kernel   Linear on <= 16 rows (adaLN, TimestepEmbedder, upsampler): dW (outer products), db, dx
//   gaussians_backward_kernel   to_gs + hard pixel alignment (denoiser.py:103-120,370-413) -> d(decoder output), d(upsampler output)
// Column sums over tokens (bias / gate / shift / scale / LayerNorm-weight gradients, dx of the adaLN Linear) are ORDER-DETERMINISTIC:
// every workgroup accumulates in registers, combines its waves in a fixed order through LDS and WRITES one partial row per column
// block into a scratch slab; col_reduce_kernel then sums the slab rows in slot order (one launch per DiT block finishes all of the
// block's column sums).  No fp32 atomics: two identical backward passes give identical bits (tools/train_determinism.py), and the
// recompute mode reproduces the save-all gradients exactly.
#include "dit_kernels.h"

namespace dgs {

// ------------------------------------------------------------------------------------------------
// bf16 [B*rows, F] (row stride ld) -> [B, F, rows].  64 x 64 tiles through LDS; 256 threads.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* in, int ld, bf16_t* out, int rows, int F) {
    __shared__ bf16_t tile[64][66];
    const int b = blockIdx.z, t0 = blockIdx.y * 64, f0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const bf16_t* src = in + ((size_t)b * rows + t0) * ld + f0;
#pragma unroll
    for (int r = ty; r < 64; r += 4) tile[r][tx] = src[(size_t)r * ld + tx];
    __syncthreads();
    bf16_t* dst = out + ((size_t)b * F + f0) * rows + t0;
#pragma unroll
    for (int r = ty; r < 64; r += 4) dst[(size_t)r * rows + tx] = tile[tx][r];
}

// ------------------------------------------------------------------------------------------------
// dy[m, n] = gate[b, n] * dx[m, n]  (bf16 row-major + transposed [B, W, rows]).  One workgroup = 64 tokens x 64 features of one
// sample; its two column sums go to part[(b * rows / 64 + token block)][0 .. 2W):
//   [0, W)   sum_t dx[m, n] * y[m, n]      (the gate's gradient)
//   [W, 2W)  sum_t dy[m, n] as rounded     (the bias gradient of the Linear whose output the gate multiplies)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_mul_kernel(const float* dx, const bf16_t* y, const float* gate, int gate_stride,
                                                      bf16_t* dy, bf16_t* dyT, float* part, int rows, int W) {
    __shared__ bf16_t tile[64][66];
    __shared__ float red[2][4][64];
    const int b = blockIdx.z, t0 = blockIdx.y * 64, f0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const size_t base = ((size_t)b * rows + t0) * W + f0;
    const float gt = gate[(size_t)b * gate_stride + f0 + tx];
    float acc = 0.f, accb = 0.f;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const size_t o = base + (size_t)r * W + tx;
        const float d = dx[o];
        const uint32_t v = f2bf_fast(gt * d);
        dy[o] = (bf16_t)v;
        tile[r][tx] = (bf16_t)v;
        acc += d * bf2f(y[o]);
        accb += bf2f(v);
    }
    red[0][ty][tx] = acc;
    red[1][ty][tx] = accb;
    __syncthreads();
    if (ty < 2) {
        float* dst = part + ((size_t)b * (rows >> 6) + blockIdx.y) * (size_t)(2 * W) + (size_t)ty * W + f0 + tx;
        *dst = (red[ty][0][tx] + red[ty][1][tx]) + (red[ty][2][tx] + red[ty][3][tx]);
    }
    bf16_t* dst = dyT + ((size_t)b * W + f0) * rows + t0;
#pragma unroll
    for (int r = ty; r < 64; r += 4) dst[(size_t)r * rows + tx] = tile[tx][r];
}

// ------------------------------------------------------------------------------------------------
// LayerNorm(+weight)+modulate backward.  h = LN(x) * w * (1 + scale) + shift.  One wave per row; a workgroup of 8 waves
// walks `rows_per_block` (32) consecutive rows of ONE sample so the per-column sums stay in registers until the end.
// (A row is a chain of dependent loads and wave reductions: with 4 waves x 32 rows and 136 workgroups the kernel was
// latency-bound at 1.1 TB/s; 8 waves x 4 rows on 544 workgroups keep every CU busy;
// 16-wave workgroups would cap the kernel at 128 VGPRs and spill at width 1024.)
// ------------------------------------------------------------------------------------------------

template <int VPL>
__global__ __launch_bounds__(512) void layernorm_backward_kernel(LnBwdParams p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.x * p.rows_per_block;
    const int b = row0 / p.rows_per_batch;
    float4 a_shift[VPL], a_scale[VPL], a_w[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) a_shift[i] = a_scale[i] = a_w[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const float inv_w = 1.0f / (float)p.width;
    // One workgroup per CU (160+ VGPRs), 8-9 rows per wave one after the other: every load of a row is in flight before its first
    // wave sum (x, dh, the incoming dx), and the NEXT row's x -- the head of its chain of dependent sums -- behind them.
    const int last = min(row0 + p.rows_per_block, p.rows);
    float4 nx[VPL];
    if (row0 + wave < last) {
        const float4* xr = reinterpret_cast<const float4*>(p.x + (size_t)(row0 + wave) * p.width);
#pragma unroll
        for (int i = 0; i < VPL; ++i) nx[i] = xr[i * 64 + lane];
    }
    for (int row = row0 + wave; row < last; row += 8) {       // (no barrier inside the loop)
        float4 n[VPL], dn[VPL], g[VPL], rin[VPL];
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c4 = i * 64 + lane;
            n[i] = nx[i];
            if (p.dh_f32) g[i] = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.dh) + (size_t)row * p.width)[c4];
            else {
                const uint2 u = reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(p.dh) + (size_t)row * p.width)[c4];
                g[i] = make_float4(bf2f(u.x & 0xffffu), bf2f(u.x >> 16), bf2f(u.y & 0xffffu), bf2f(u.y >> 16));
            }
            if (p.dx_in) rin[i] = reinterpret_cast<const float4*>(p.dx_in + (size_t)row * p.width)[c4];
        }
        if (row + 8 < last) {
            const float4* xr = reinterpret_cast<const float4*>(p.x + (size_t)(row + 8) * p.width);
#pragma unroll
            for (int i = 0; i < VPL; ++i) nx[i] = xr[i * 64 + lane];
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) sum += (n[i].x + n[i].y) + (n[i].z + n[i].w);
        const float mean = wave_sum(sum) * inv_w;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            n[i].x -= mean; n[i].y -= mean; n[i].z -= mean; n[i].w -= mean;
            sq += (n[i].x * n[i].x + n[i].y * n[i].y) + (n[i].z * n[i].z + n[i].w * n[i].w);
        }
        const float rstd = rsqrtf(wave_sum(sq) * inv_w + p.eps);
        float s1 = 0.f, s2 = 0.f;     // sum(dn), sum(dn * n)
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c4 = i * 64 + lane;
            n[i].x *= rstd; n[i].y *= rstd; n[i].z *= rstd; n[i].w *= rstd;
            float4 w = make_float4(1.f, 1.f, 1.f, 1.f), m1 = make_float4(1.f, 1.f, 1.f, 1.f);
            if (p.weight) w = reinterpret_cast<const float4*>(p.weight)[c4];
            if (p.scale) {
                const float4 sc = reinterpret_cast<const float4*>(p.scale + (size_t)b * p.mod_stride)[c4];
                m1 = make_float4(1.f + sc.x, 1.f + sc.y, 1.f + sc.z, 1.f + sc.w);
            }
            a_shift[i].x += g[i].x; a_shift[i].y += g[i].y; a_shift[i].z += g[i].z; a_shift[i].w += g[i].w;
            a_scale[i].x += g[i].x * n[i].x * w.x; a_scale[i].y += g[i].y * n[i].y * w.y; a_scale[i].z += g[i].z * n[i].z * w.z; a_scale[i].w += g[i].w * n[i].w * w.w;
            a_w[i].x += g[i].x * m1.x * n[i].x; a_w[i].y += g[i].y * m1.y * n[i].y; a_w[i].z += g[i].z * m1.z * n[i].z; a_w[i].w += g[i].w * m1.w * n[i].w;
            dn[i] = make_float4(g[i].x * m1.x * w.x, g[i].y * m1.y * w.y, g[i].z * m1.z * w.z, g[i].w * m1.w * w.w);
            s1 += (dn[i].x + dn[i].y) + (dn[i].z + dn[i].w);
            s2 += (dn[i].x * n[i].x + dn[i].y * n[i].y) + (dn[i].z * n[i].z + dn[i].w * n[i].w);
        }
        const float m1s = wave_sum(s1) * inv_w, m2s = wave_sum(s2) * inv_w;
        float4* out = reinterpret_cast<float4*>(p.dx_out + (size_t)row * p.width);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c4 = i * 64 + lane;
            float4 d = make_float4(rstd * (dn[i].x - m1s - n[i].x * m2s), rstd * (dn[i].y - m1s - n[i].y * m2s),
                                   rstd * (dn[i].z - m1s - n[i].z * m2s), rstd * (dn[i].w - m1s - n[i].w * m2s));
            if (p.dx_in) { d.x += rin[i].x; d.y += rin[i].y; d.z += rin[i].z; d.w += rin[i].w; }
            out[c4] = d;
        }
    }
    // column sums: 8 waves -> LDS -> summed in wave order, one quantity at a time (32 KiB of LDS).  With a slab (`part`) the
    // workgroup WRITES its row [shift | scale | weight]; without (single-workgroup launches: the learned-token rows of the
    // upsampler head) it adds to the destinations -- plain adds, launches are stream-ordered.
    __shared__ float red[8][VPL * 256];
    float* const dst[3] = {p.dshift ? p.dshift + (size_t)b * p.mod_stride : nullptr, p.dscale ? p.dscale + (size_t)b * p.mod_stride : nullptr, p.dweight};
#pragma unroll
    for (int qn = 0; qn < 3; ++qn) {
        if (!dst[qn]) continue;                        // uniform
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VPL; ++i)
            *reinterpret_cast<float4*>(&red[wave][(i * 64 + lane) * 4]) = qn == 0 ? a_shift[i] : qn == 1 ? a_scale[i] : a_w[i];
        __syncthreads();
        for (int c = threadIdx.x; c < p.width; c += 512) {
            float acc = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) acc += red[w][c];
            if (p.part) p.part[(size_t)blockIdx.x * p.part_stride + qn * p.width + c] = acc;
            else dst[qn][c] += acc;
        }
    }
}

// part[row block][n] = sum over the block's rows of dY[m, n]   (bf16 [M, ld]); workgroup = 256 rows x 64 columns (any N % 64 == 0)
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* dy, int ld, int M, float* part, int N) {
    __shared__ float sm[4][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + tx, m0 = blockIdx.y * 256;
    float acc = 0.f;
    for (int r = ty; r < 256 && m0 + r < M; r += 4) acc += bf2f(dy[(size_t)(m0 + r) * ld + n]);
    sm[ty][tx] = acc;
    __syncthreads();
    if (ty == 0) part[(size_t)blockIdx.y * N + n] = (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]);
}

// The same for N % 512 == 0 (every bias of the DiT blocks): workgroup = 128 rows x 512 columns, a wave reads 1 KiB of a row per
// instruction (16 bytes per lane; the 2-byte loads above ran at 1.5 TB/s), 8 rows in flight per wave.
__global__ __launch_bounds__(256) void colsum_wide_kernel(const bf16_t* dy, int ld, int M, float* part, int N) {
    __shared__ float sm[4][512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 512 + lane * 8, m0 = blockIdx.y * 128;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r0 = wave; r0 < 128; r0 += 32) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + r0 + 4 * u;
            v[u] = m < M ? *reinterpret_cast<const uint4*>(dy + (size_t)m * ld + n0) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[0] += __uint_as_float(v[u].x << 16); acc[1] += __uint_as_float(v[u].x & 0xffff0000u);
            acc[2] += __uint_as_float(v[u].y << 16); acc[3] += __uint_as_float(v[u].y & 0xffff0000u);
            acc[4] += __uint_as_float(v[u].z << 16); acc[5] += __uint_as_float(v[u].z & 0xffff0000u);
            acc[6] += __uint_as_float(v[u].w << 16); acc[7] += __uint_as_float(v[u].w & 0xffff0000u);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sm[wave][lane * 8 + i] = acc[i];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256)
        part[(size_t)blockIdx.y * N + blockIdx.x * 512 + c] = (sm[0][c] + sm[1][c]) + (sm[2][c] + sm[3][c]);
}

// ------------------------------------------------------------------------------------------------
// out[b * out_bstride + c] = sum_{s < slots} part[(b * slots + s) * part_stride + c]  for up to 8 independent jobs in one launch
// (blockIdx.z = job, blockIdx.y = b, 64 columns per workgroup).  Thread (column, q) adds slots q, q + 4, ... in order, the four
// sums are combined as (0 + 1) + (2 + 3): the same order on every run.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void col_reduce_kernel(ColReduceParams p) {
    __shared__ float sm[4][64];
    const ColReduceJob j = p.job[blockIdx.z];
    const int tx = threadIdx.x & 63, q = threadIdx.x >> 6, c = blockIdx.x * 64 + tx, b = blockIdx.y;
    if (b >= j.batches || (int)blockIdx.x * 64 >= j.cols) return;      // uniform
    float acc = 0.f;
    if (c < j.cols) {
        const float* src = j.part + (size_t)b * j.slots * j.part_stride + c;
#pragma unroll 8
        for (int s = q; s < j.slots; s += 4) acc += src[(size_t)s * j.part_stride];
    }
    sm[q][tx] = acc;
    __syncthreads();
    if (q == 0 && c < j.cols) j.out[(size_t)b * j.out_bstride + c] = (sm[0][tx] + sm[1][tx]) + (sm[2][tx] + sm[3][tx]);
}

// ------------------------------------------------------------------------------------------------
// Backward of  y[m, n] = sum_k act(x[m, k]) W[n, k] + b[n]  on M <= 16 rows (weight-streaming, one wave per output row n):
//   dW[n, :] = sum_m dy[m, n] act(x[m, :])        (written, not accumulated)
//   db[n]    = sum_m dy[m, n]
//   dx[m, k]  = act'(x[m, k]) sum_n dy[m, n] W[n, k]      (per workgroup: a slab row part[blockIdx.x][M * K] that col_reduce sums,
//                                                          or -- one workgroup only -- added to dx)
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + __expf(-v)); }
__device__ __forceinline__ float dsilu_f(float v) { const float s = 1.0f / (1.0f + __expf(-v)); return s * (1.0f + v * (1.0f - s)); }

template <int MR>
__global__ __launch_bounds__(256) void rowlinear_backward_kernel(RowLinBwdParams p, int rows_per_block) {
    DGS_DYNAMIC_LDS(smem);
    float* xs = reinterpret_cast<float*>(smem);           // [MR][K] activated input
    float* dxs = xs + (size_t)MR * p.K;                   // [MR][K] partial dx of this workgroup
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < MR * p.K; i += 256) {
        const int m = i / p.K;
        float v = (m < p.M) ? p.x[i] : 0.0f;
        xs[i] = p.silu_in ? silu_f(v) : v;
        dxs[i] = 0.f;
    }
    __syncthreads();
    // lane owns the 8-column chunks {lane, lane + 64, ...} (K <= 1024 -> at most 2 chunks): dx partials stay in registers
    constexpr int MAXC = 2;
    float dxr[MR][MAXC][8];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) dxr[m][c][j] = 0.f;
    const int n0 = blockIdx.x * rows_per_block;
    for (int n = n0 + wave; n < n0 + rows_per_block && n < p.N; n += 4) {
        float dyv[MR];
        float bsum = 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) { dyv[m] = (m < p.M) ? p.dy[(size_t)m * (p.ldy ? p.ldy : p.N) + n] : 0.f; bsum += dyv[m]; }
        if (p.db && lane == 0) p.db[n] = bsum;
        const bf16_t* wr = p.W + (size_t)n * p.K;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            const int k0 = (lane + 64 * c) * 8;
            if (k0 >= p.K) continue;
            if (p.dx) {                                          // uniform: the weight row is only needed for dx
                const uint4 w = *reinterpret_cast<const uint4*>(wr + k0);
                const float wf[8] = {bf2f(w.x & 0xffffu), bf2f(w.x >> 16), bf2f(w.y & 0xffffu), bf2f(w.y >> 16),
                                     bf2f(w.z & 0xffffu), bf2f(w.z >> 16), bf2f(w.w & 0xffffu), bf2f(w.w >> 16)};
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int j = 0; j < 8; ++j) dxr[m][c][j] += dyv[m] * wf[j];
            }
            if (p.dW) {
                float g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const float4 xa = *reinterpret_cast<const float4*>(xs + m * p.K + k0);
                    const float4 xb = *reinterpret_cast<const float4*>(xs + m * p.K + k0 + 4);
                    const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[j] += dyv[m] * xv[j];
                }
                float4* dst = reinterpret_cast<float4*>(p.dW + (size_t)n * p.K + k0);
                dst[0] = make_float4(g[0], g[1], g[2], g[3]);
                dst[1] = make_float4(g[4], g[5], g[6], g[7]);
            }
        }
    }
    if (p.dx) {
        // the four waves add their registers into the LDS copy one after the other (fixed order)
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int m = 0; m < MR; ++m)
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) {
                        const int k0 = (lane + 64 * c) * 8;
                        if (k0 >= p.K) continue;
#pragma unroll
                        for (int j = 0; j < 8; ++j) dxs[m * p.K + k0 + j] += dxr[m][c][j];
                    }
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < p.M * p.K; i += 256) {
            float g = dxs[i];
            if (p.silu_in) g *= dsilu_f(p.x[i]);
            if (p.part) p.part[(size_t)blockIdx.x * p.M * p.K + i] = g;
            else p.dx[i] += g;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of gaussians_kernel (to_gs + pixel alignment).  One thread per Gaussian; writes the gradient of the decoder
// GEMM output as bf16 [B*lpad, ps*ps*C] (row-major; rows of learned tokens / padding stay zero) and of the upsampler
// output as f32 [B*ng, C].
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void gaussians_backward_kernel(GsBwdParams p) {
    const size_t HW = (size_t)p.H * p.W;
    const size_t P = (size_t)p.ng + (size_t)p.V * HW;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)p.B * P) return;
    const int b = (int)(gid / P);
    const size_t i = gid % P;
    float d[14];
    const float gx = p.dxyz[gid * 3], gy = p.dxyz[gid * 3 + 1], gz = p.dxyz[gid * 3 + 2];
    d[3] = p.dfeatures[gid * 3]; d[4] = p.dfeatures[gid * 3 + 1]; d[5] = p.dfeatures[gid * 3 + 2];
    d[9] = p.drotation[gid * 4]; d[10] = p.drotation[gid * 4 + 1]; d[11] = p.drotation[gid * 4 + 2]; d[12] = p.drotation[gid * 4 + 3];
    d[13] = p.dopacity[gid];
    if (i < (size_t)p.ng) {
        const float* src = p.up + ((size_t)b * p.ng + i) * p.C;
        d[0] = gx; d[1] = gy; d[2] = gz;
#pragma unroll
        for (int k = 0; k < 3; ++k) d[6 + k] = (src[6 + k] - 2.3f < -1.2f) ? p.dscaling[gid * 3 + k] : 0.f;   // clamp(max=-1.2)
        float* dst = p.dup + ((size_t)b * p.ng + i) * p.C;
#pragma unroll
        for (int k = 0; k < 14; ++k) dst[k] = d[k];
        return;
    }
    const size_t j = i - p.ng;
    const int pp = p.ps * p.ps;
    const size_t tok = j / pp;
    const int pi = (int)(j % pp);
    const size_t off = ((size_t)b * p.lpad + tok) * (size_t)(pp * p.C) + (size_t)pi * p.C;
    const float* src = p.dec + off;
    const int np_w = p.W / p.ps, np = (p.H / p.ps) * np_w;
    const int v = (int)(tok / np), hh = (int)((tok % np) / np_w), ww = (int)(tok % np_w);
    const int h = hh * p.ps + pi / p.ps, w = ww * p.ps + pi % p.ps;
    const size_t base = ((size_t)b * p.V + v) * 3 * HW + (size_t)h * p.W + w;
    const float dx = p.ray_d[base], dy = p.ray_d[base + HW], dz = p.ray_d[base + 2 * HW];
    const float mean = (src[0] + src[1] + src[2]) / 3.0f;
    const float sg = 1.0f / (1.0f + __expf(-mean));
    float ddepth = gx * dx + gy * dy + gz * dz;                     // xyz = ray_o + depth * ray_d
    float k = sg * (1.0f - sg);
    if (p.scene) k *= (p.range_far - p.range_near);
    else if (p.relative_plk) k *= 2.0f * 1.8f;
    ddepth = ddepth * k / 3.0f;
    d[0] = d[1] = d[2] = ddepth;
#pragma unroll
    for (int q = 0; q < 3; ++q) d[6 + q] = (src[6 + q] - 2.3f < -1.2f) ? p.dscaling[gid * 3 + q] : 0.f;
    bf16_t* dst = p.ddec + off;
#pragma unroll
    for (int q = 0; q < 14; ++q) dst[q] = (bf16_t)f2bf_fast(d[q]);
}

static int ok() { return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE; }

int launch_transpose(const bf16_t* in, int ld, bf16_t* out, int B, int rows, int F, hipStream_t st) {
    if (rows % 64 || F % 64) return DGS_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(transpose_kernel, dim3(F / 64, rows / 64, B), dim3(256), 0, st, in, ld, out, rows, F);
    return ok();
}

int launch_gate_mul(const float* dx, const bf16_t* y, const float* gate, int gate_stride, bf16_t* dy, bf16_t* dyT, float* part, int B,
                    int rows, int W, hipStream_t st) {
    if (rows % 64 || W % 64 || !part) return DGS_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gate_mul_kernel, dim3(W / 64, rows / 64, B), dim3(256), 0, st, dx, y, gate, gate_stride, dy, dyT, part, rows, W);
    return ok();
}

int launch_col_reduce(const ColReduceJob* jobs, int njobs, hipStream_t st) {
    if (njobs <= 0 || njobs > 8) return DGS_ERR_INVALID_ARGUMENT;
    ColReduceParams p{};
    int cols = 0, batches = 0;
    for (int i = 0; i < njobs; ++i) {
        p.job[i] = jobs[i];
        if (!jobs[i].part || !jobs[i].out || jobs[i].slots <= 0 || jobs[i].cols <= 0 || jobs[i].batches <= 0) return DGS_ERR_INVALID_ARGUMENT;
        cols = jobs[i].cols > cols ? jobs[i].cols : cols;
        batches = jobs[i].batches > batches ? jobs[i].batches : batches;
    }
    hipLaunchKernelGGL(col_reduce_kernel, dim3((cols + 63) / 64, batches, njobs), dim3(256), 0, st, p);
    return ok();
}

int ln_backward_compute_units() { return compute_unit_count_cached(); }

int launch_layernorm_backward(const LnBwdParams& p0, hipStream_t st) {
    LnBwdParams p = p0;
    if (p.rows <= 0 || p.width % 256 || p.width > 2048) return DGS_ERR_INVALID_ARGUMENT;
    if (p.rows_per_batch <= 0) p.rows_per_batch = p.rows;
    p.rows_per_block = ln_backward_rows_per_block(p.rows_per_batch, p.rows, ln_backward_compute_units());
    const dim3 grid((p.rows + p.rows_per_block - 1) / p.rows_per_block), block(512);
    if (!p.part && grid.x != 1 && (p.dshift || p.dscale || p.dweight)) return DGS_ERR_INVALID_ARGUMENT;   // column sums need the slab
    if (p.part && p.part_stride < 3 * p.width) return DGS_ERR_INVALID_ARGUMENT;
    switch (p.width / 256) {
        case 1: hipLaunchKernelGGL((layernorm_backward_kernel<1>), grid, block, 0, st, p); break;
        case 2: hipLaunchKernelGGL((layernorm_backward_kernel<2>), grid, block, 0, st, p); break;
        case 4: hipLaunchKernelGGL((layernorm_backward_kernel<4>), grid, block, 0, st, p); break;
        default: return DGS_ERR_INVALID_ARGUMENT;
    }
    return ok();
}

// part: [colsum_slots(M, N, ld)][N] slab rows, summed by launch_col_reduce
int colsum_slots(int M, int N, int ld) { return (N % 512 == 0 && ld % 8 == 0) ? (M + 127) / 128 : (M + 255) / 256; }

int launch_colsum(const bf16_t* dy, int ld, int M, int N, float* part, hipStream_t st) {
    if (N % 64 || !part) return DGS_ERR_INVALID_ARGUMENT;
    if (N % 512 == 0 && ld % 8 == 0) hipLaunchKernelGGL(colsum_wide_kernel, dim3(N / 512, (M + 127) / 128), dim3(256), 0, st, dy, ld, M, part, N);
    else hipLaunchKernelGGL(colsum_kernel, dim3(N / 64, (M + 255) / 256), dim3(256), 0, st, dy, ld, M, part, N);
    return ok();
}

int launch_rowlinear_backward(const RowLinBwdParams& p, hipStream_t st) {
    if (p.M <= 0 || p.M > 8 || p.N <= 0 || p.K <= 0 || p.K % 8 || p.K > 1024) return DGS_ERR_INVALID_ARGUMENT;
    const int rpb = p.rows_per_block > 0 ? p.rows_per_block : ROWLINEAR_BWD_ROWS;
    const dim3 grid((p.N + rpb - 1) / rpb), block(256);
    if (p.dx && !p.part && grid.x != 1) return DGS_ERR_INVALID_ARGUMENT;       // dx of several workgroups needs the slab
    const int mr = p.M <= 1 ? 1 : p.M <= 2 ? 2 : p.M <= 4 ? 4 : 8;
    const size_t lds = (size_t)2 * mr * p.K * sizeof(float);
    if (lds > 65536) return DGS_ERR_INVALID_ARGUMENT;
    switch (mr) {
        case 1: hipLaunchKernelGGL((rowlinear_backward_kernel<1>), grid, block, lds, st, p, rpb); break;
        case 2: hipLaunchKernelGGL((rowlinear_backward_kernel<2>), grid, block, lds, st, p, rpb); break;
        case 4: hipLaunchKernelGGL((rowlinear_backward_kernel<4>), grid, block, lds, st, p, rpb); break;
        default: hipLaunchKernelGGL((rowlinear_backward_kernel<8>), grid, block, lds, st, p, rpb); break;
    }
    return ok();
}

int launch_gaussians_backward(const GsBwdParams& p, hipStream_t st) {
    const size_t n = (size_t)p.B * ((size_t)p.ng + (size_t)p.V * p.H * p.W);
    hipLaunchKernelGGL(gaussians_backward_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    return ok();
}

}  // namespace dgs

using namespace dgs;

namespace {
int ln_blocks(int rows, int rows_per_batch) {
    const int rpb = ln_backward_rows_per_block(rows_per_batch > 0 ? rows_per_batch : rows, rows, ln_backward_compute_units());
    return (rows + rpb - 1) / rpb;
}
}  // namespace

extern "C" size_t dgs_dit_layernorm_backward_scratch_bytes(int32_t rows, int32_t width, int32_t rows_per_batch) {
    if (rows <= 0 || width <= 0) return 0;
    return (size_t)ln_blocks(rows, rows_per_batch) * 3 * width * sizeof(float);
}
extern "C" size_t dgs_dit_gate_mul_scratch_bytes(int32_t B, int32_t rows, int32_t width) {
    if (B <= 0 || rows <= 0 || width <= 0) return 0;
    return (size_t)B * (rows / 64) * 2 * width * sizeof(float);
}
extern "C" size_t dgs_dit_rowlinear_backward_scratch_bytes(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    return (size_t)((N + ROWLINEAR_BWD_ROWS - 1) / ROWLINEAR_BWD_ROWS) * M * K * sizeof(float);
}

// The three entry points below WRITE their column sums (dshift / dscale / dweight, dgate / dbias, dx): partial rows in the caller's
// scratch, then one col_reduce launch -- the same two steps dgs_dit_backward runs, with the reduce of a whole block batched there.
extern "C" int dgs_dit_layernorm_backward(const DgsDitLayerNormBackwardArgs* a, dgs_stream_t stream) {
    if (!a || !a->x || !a->dh || !a->dx_out) return DGS_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    LnBwdParams p{};
    p.rows = a->rows; p.width = a->width; p.mod_stride = a->mod_stride; p.rows_per_batch = a->rows_per_batch; p.dh_f32 = a->dh_f32;
    p.eps = a->eps; p.x = a->x; p.dh = a->dh; p.weight = a->weight; p.scale = a->scale; p.dx_in = a->dx_in; p.dx_out = a->dx_out;
    p.dshift = a->dshift; p.dscale = a->dscale; p.dweight = a->dweight;
    if (!(a->dshift || a->dscale || a->dweight)) return launch_layernorm_backward(p, st);
    if (!a->scratch || a->scratch_bytes < dgs_dit_layernorm_backward_scratch_bytes(a->rows, a->width, a->rows_per_batch)) return DGS_ERR_ALLOC;
    p.part = static_cast<float*>(a->scratch); p.part_stride = 3 * a->width;
    const int rc = launch_layernorm_backward(p, st);
    if (rc != DGS_OK) return rc;
    const int rpbatch = a->rows_per_batch > 0 ? a->rows_per_batch : a->rows;
    const int batches = a->rows / rpbatch, blocks = ln_blocks(a->rows, a->rows_per_batch);
    ColReduceJob jobs[3];
    int n = 0;
    if (a->dshift) jobs[n++] = ColReduceJob{p.part, a->dshift, blocks / batches, p.part_stride, a->width, batches, a->mod_stride};
    if (a->dscale) jobs[n++] = ColReduceJob{p.part + a->width, a->dscale, blocks / batches, p.part_stride, a->width, batches, a->mod_stride};
    if (a->dweight) jobs[n++] = ColReduceJob{p.part + 2 * a->width, a->dweight, blocks, p.part_stride, a->width, 1, 0};
    return launch_col_reduce(jobs, n, st);
}

extern "C" int dgs_dit_rowlinear_backward(const DgsDitRowLinearBackwardArgs* a, dgs_stream_t stream) {
    if (!a || !a->x || !a->W || !a->dy) return DGS_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    RowLinBwdParams p{};
    p.M = a->M; p.N = a->N; p.K = a->K; p.silu_in = a->silu_input; p.x = a->x; p.W = a->W; p.dy = a->dy;
    p.dW = a->dW; p.db = a->db; p.dx = a->dx;
    if (!a->dx) return launch_rowlinear_backward(p, st);
    if (!a->scratch || a->scratch_bytes < dgs_dit_rowlinear_backward_scratch_bytes(a->M, a->N, a->K)) return DGS_ERR_ALLOC;
    p.part = static_cast<float*>(a->scratch);
    const int rc = launch_rowlinear_backward(p, st);
    if (rc != DGS_OK) return rc;
    const ColReduceJob job{p.part, a->dx, (a->N + ROWLINEAR_BWD_ROWS - 1) / ROWLINEAR_BWD_ROWS, a->M * a->K, a->M * a->K, 1, 0};
    return launch_col_reduce(&job, 1, st);
}

extern "C" int dgs_dit_gate_mul(const DgsDitGateMulArgs* a, dgs_stream_t stream) {
    if (!a || !a->dx || !a->y || !a->gate || !a->dy || !a->dyT || !a->dgate) return DGS_ERR_INVALID_ARGUMENT;
    if (!a->scratch || a->scratch_bytes < dgs_dit_gate_mul_scratch_bytes(a->B, a->rows, a->width)) return DGS_ERR_ALLOC;
    hipStream_t st = static_cast<hipStream_t>(stream);
    float* part = static_cast<float*>(a->scratch);
    const int rc = launch_gate_mul(a->dx, a->y, a->gate, a->gate_stride, a->dy, a->dyT, part, a->B, a->rows, a->width, st);
    if (rc != DGS_OK) return rc;
    ColReduceJob jobs[2];
    int n = 0;
    jobs[n++] = ColReduceJob{part, a->dgate, a->rows / 64, 2 * a->width, a->width, a->B, a->gate_stride};
    if (a->dbias) jobs[n++] = ColReduceJob{part + a->width, a->dbias, a->B * (a->rows / 64), 2 * a->width, a->width, 1, 0};
    return launch_col_reduce(jobs, n, st);
}
