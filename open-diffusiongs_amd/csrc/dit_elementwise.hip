// dit_elementwise.hip -- the HBM-bound kernels around the DiT GEMMs (gfx950, wave64):
//   layernorm_kernel      nn.LayerNorm (+ optional weight) fused with modulate()   utils_transformer.py:26-27,271-290
//   rowlinear_kernel      Linear on <= 16 rows (adaLN for all layers at once, TimestepEmbedder, upsampler head)
//   timestep_freq_kernel  sinusoidal timestep embedding                            denoiser.py:46-67
//   embed_patchify_kernel ray / Plucker embedding + 8x8 patchify -> bf16 GEMM operand   denoiser.py:312-334,210-215
//   gaussians_kernel      GaussiansUpsampler.to_gs + hard pixel alignment           denoiser.py:103-120,370-413
// All are one pass over their input with 16-byte accesses; nothing here is MFMA work.
#include "dit_kernels.h"
#include "dit_gemm_epilogue.h"

namespace dgs {

// ------------------------------------------------------------------------------------------------
// LayerNorm (+weight) (+modulate): one wave per row, each lane holds width/64 elements in registers.
// Two-pass statistics in registers (mean, then centred variance) like torch's fp32 LayerNorm.
// ------------------------------------------------------------------------------------------------

// The per-column operands (LayerNorm weight, adaLN shift / scale) do not depend on the statistics: they are requested right behind the
// row itself, so the kernel is ONE memory round trip.  (As run-time tests of p.weight / p.shift inside the output loop the
// compiler emitted, per 1 KiB of the row, branch -> load -> s_waitcnt vmcnt(0) -> store: four more dependent round trips behind the
// reductions, ~2 of the kernel's 6.9 us at the DiT shape.)  Same arithmetic in the same order as before: outputs are bit-identical.
// One row by one wave: y = the row normalised (weighted, modulated), lane's float4 i = columns 4 (64 i + lane) .. + 3.
template <int VPL, bool WEIGHT, bool MOD>   // float4 vectors per lane: width = 256 * VPL
__device__ __forceinline__ void ln_row(const LnParams& p, int b, int row, int lane, float4 (&y)[VPL]) {
    const float4* xr = reinterpret_cast<const float4*>(p.x + (size_t)row * p.width);
    float4 v[VPL], w[WEIGHT ? VPL : 1], sh[MOD ? VPL : 1], sc[MOD ? VPL : 1];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) v[i] = xr[i * 64 + lane];
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c4 = i * 64 + lane;   // float4 index inside the row
        if constexpr (WEIGHT) w[i] = reinterpret_cast<const float4*>(p.weight)[c4];
        if constexpr (MOD) {
            sh[i] = reinterpret_cast<const float4*>(p.shift + (size_t)b * p.mod_stride)[c4];
            sc[i] = reinterpret_cast<const float4*>(p.scale + (size_t)b * p.mod_stride)[c4];
        }
    }
    sched_fence();                      // every load is issued before the first use of the row (hipcc otherwise sinks half of them)
#pragma unroll
    for (int i = 0; i < VPL; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(sum) / (float)p.width;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        sq += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)p.width + p.eps);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        y[i] = make_float4(v[i].x * rstd, v[i].y * rstd, v[i].z * rstd, v[i].w * rstd);
        if constexpr (WEIGHT) { y[i].x *= w[i].x; y[i].y *= w[i].y; y[i].z *= w[i].z; y[i].w *= w[i].w; }
        if constexpr (MOD) {
            y[i].x = y[i].x * (1.0f + sc[i].x) + sh[i].x; y[i].y = y[i].y * (1.0f + sc[i].y) + sh[i].y;
            y[i].z = y[i].z * (1.0f + sc[i].z) + sh[i].z; y[i].w = y[i].w * (1.0f + sc[i].w) + sh[i].w;
        }
    }
}

template <int VPL, bool WEIGHT, bool MOD, bool F32OUT>
__device__ __forceinline__ void ln_rows_role(const LnParams& p, int b, int local, int lane) {
    const int row = b * p.rows_per_batch + local;
    if (local >= p.rows_per_batch || row >= p.rows) return;
    float4 y[VPL];
    ln_row<VPL, WEIGHT, MOD>(p, b, row, lane, y);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c4 = i * 64 + lane;
        if constexpr (F32OUT) reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (size_t)row * p.width)[c4] = y[i];
        else reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(p.out) + (size_t)row * p.width)[c4] =
                 make_uint2(pack_bf2(y[i].x, y[i].y), pack_bf2(y[i].z, y[i].w));
    }
}

template <int VPL, bool WEIGHT, bool MOD, bool F32OUT>
__global__ __launch_bounds__(256) void layernorm_kernel(LnParams p) {
    // the sample: grid.y (no per-row division in front of the operand loads)
    ln_rows_role<VPL, WEIGHT, MOD, F32OUT>(p, blockIdx.y, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// The LayerNorm + modulate launch in front of a GEMM whose samples end one or two rows behind their last full 256-row tile (the DiT's
// learned tokens: L = 4096 v + 2).  Inside the GEMM those rows are two-row GEMV items of the first workgroups' prologues, and the
// launch pays for them: QKV +2.8 us, fc1 +2.4 us at one sample (tools/gemm_tail_cost.py) -- the item's round trip queues behind the
// ring's first slabs, and the workgroups that carry one enter their loop 3 us late and end the launch (profiles/r05_gemm_timeline.txt).
// Here the same items are the FIRST workgroups of the LayerNorm launch, which is one memory round trip long anyway: an item
// normalises the sample's live rows itself (the rows' LayerNorm output does not exist yet in this launch: same ln_row, same bits),
// and then is the GEMM's item with 4 waves instead of 8: K ranges of 512 per wave, the ranges summed in the same order -- the
// outputs are bit-identical to the GEMM's own items (tests/test_dit_kernels_emu.py, tests/test_dit_gpu.py).
struct TailEpi {               // what tail_prefetch / tail_store read of a GEMM's parameter block (dit_gemm_epilogue.h)
    const float *bias, *resid, *gate;
    void *out, *aux;
    bf16_t* vt;
    int N, ldo, gate_stride, rows_per_batch;
    float q_scale;
};

template <int VPL, int EPI>
__global__ __launch_bounds__(256) void layernorm_rows_gemv_kernel(LnParams p, LnRowsGemv j) {
    constexpr int K = 256 * VPL, KS = K / 512 < 4 ? K / 512 : 4, CS = 4 / KS, CPI = 8 * CS, KW = K / KS;
    static_assert(KW == 512, "one 16-byte load per lane and row");
    __shared__ uint2 s_rows[2][K / 4];                                   // the live rows' LayerNorm output, bf16
    __shared__ float s_part[KS * 2 * CPI];
    const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if ((int)blockIdx.x >= j.items) {
        ln_rows_role<VPL, false, true, false>(p, b, ((int)blockIdx.x - j.items) * 4 + wave, lane);
        return;
    }
    const int blk = blockIdx.x, kq = wave % KS, cq = wave / KS, k_lo = kq * KW;
    uint4 w[8];
    const bf16_t* w_col0 = j.W + (size_t)(blk * CPI + cq * 8) * j.ldw + k_lo + lane * 8;
#pragma unroll
    for (int c = 0; c < 8; ++c) w[c] = *reinterpret_cast<const uint4*>(w_col0 + (size_t)c * j.ldw);
    TailEpi te{j.bias, nullptr, nullptr, j.out, j.aux, j.vt, j.N, j.ldo, 0, p.rows_per_batch, j.q_scale};
    const int er = tid / CPI, ec = tid - er * CPI;                        // the element thread `tid` finishes (tid < 2 CPI)
    const int erow = b * p.rows_per_batch + j.row0 + er;
    const bool finisher = tid < 2 * CPI && er < j.nrows;
    TailOperands ops{0.f, 0.f, 0.f};
    if (finisher) ops = tail_prefetch<EPI>(te, erow, blk * CPI + ec);
    if (wave < 2) {
        if (wave < j.nrows) {
            float4 y[VPL];
            ln_row<VPL, false, true>(p, b, b * p.rows_per_batch + j.row0 + wave, lane, y);
#pragma unroll
            for (int i = 0; i < VPL; ++i) s_rows[wave][i * 64 + lane] = make_uint2(pack_bf2(y[i].x, y[i].y), pack_bf2(y[i].z, y[i].w));
        } else {
#pragma unroll
            for (int i = 0; i < VPL; ++i) s_rows[wave][i * 64 + lane] = make_uint2(0u, 0u);
        }
    }
    __syncthreads();
    const uint4 a0 = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(s_rows[0]) + k_lo + lane * 8);
    const uint4 a1 = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(s_rows[1]) + k_lo + lane * 8);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        float s0 = dot8_bf16(a0, w[c], 0.f), s1 = dot8_bf16(a1, w[c], 0.f);
        s0 = wave_sum_lane63(s0); s1 = wave_sum_lane63(s1);
        if (lane == 63) { s_part[(kq * 2 + 0) * CPI + cq * 8 + c] = s0; s_part[(kq * 2 + 1) * CPI + cq * 8 + c] = s1; }
    }
    __syncthreads();
    if (finisher) {
        float v = 0.f;
        for (int q = 0; q < KS; ++q) v += s_part[(q * 2 + er) * CPI + ec];
        tail_store<EPI>(te, erow, blk * CPI + ec, v, ops);
    }
}

// ------------------------------------------------------------------------------------------------
// Linear on M <= 16 rows: out[m, n] = act_out( sum_k act_in(x[m, k]) W[n, k] + bias[n] ).
// Weight-streaming GEMV: one wave per output feature n, 16-byte bf16 loads along K, x staged in LDS as f32,
// wave-reduce at the end.  HBM-bound on W (each weight byte is read exactly once).
// ------------------------------------------------------------------------------------------------

__device__ __forceinline__ float silu(float v) { return v / (1.0f + __expf(-v)); }

constexpr int kRowLinFeatures = 32;

template <int MR>   // rows handled per pass (compile-time for register accumulators)
__global__ __launch_bounds__(256) void rowlinear_kernel(RowLinParams p) {
    DGS_DYNAMIC_LDS(smem);
    float* xs = reinterpret_cast<float*>(smem);   // [MR][K]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < MR * p.K; i += 256) {
        const int m = i / p.K;
        float v = (m < p.M) ? p.x[i] : 0.0f;
        if (p.silu_in) v = silu(v);
        xs[i] = v;
    }
    __syncthreads();
    // kRowLinFeatures output features per workgroup, a wave takes every fourth: the staging of x (and its SiLU) above is paid once
    // per 32 features instead of once per 4.  The adaLN GEMV of all 24 blocks streams 310 MB of weights once per forward, so what
    // counts is bytes in flight: NF features' weight rows are requested per wave before the first is used (two: 3.6 TB/s; eight,
    // with one or two input rows: see DESIGN.md section 8).  Per feature the sum runs over k in the same order whatever NF is.
    constexpr int NF = MR <= 2 ? 8 : 2;
    const int n_end = min(p.N, ((int)blockIdx.x + 1) * kRowLinFeatures);
    for (int n = blockIdx.x * kRowLinFeatures + wave; n < n_end; n += 4 * NF) {
        float acc[NF][MR];
        const bf16_t* wr[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            wr[f] = p.W + (size_t)(n + 4 * f < n_end ? n + 4 * f : n) * p.K;      // a feature past the end re-reads the first (result dropped)
#pragma unroll
            for (int m = 0; m < MR; ++m) acc[f][m] = 0.f;
        }
        for (int k0 = lane * 8; k0 < p.K; k0 += 512) {
            uint4 w[NF];
#pragma unroll
            for (int f = 0; f < NF; ++f) w[f] = *reinterpret_cast<const uint4*>(wr[f] + k0);
            float4 xa[MR], xb[MR];
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                xa[m] = *reinterpret_cast<const float4*>(xs + m * p.K + k0);
                xb[m] = *reinterpret_cast<const float4*>(xs + m * p.K + k0 + 4);
            }
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const float wf[8] = {bf2f(w[f].x & 0xffffu), bf2f(w[f].x >> 16), bf2f(w[f].y & 0xffffu), bf2f(w[f].y >> 16),
                                     bf2f(w[f].z & 0xffffu), bf2f(w[f].z >> 16), bf2f(w[f].w & 0xffffu), bf2f(w[f].w >> 16)};
#pragma unroll
                for (int m = 0; m < MR; ++m)
                    acc[f][m] += wf[0] * xa[m].x + wf[1] * xa[m].y + wf[2] * xa[m].z + wf[3] * xa[m].w + wf[4] * xb[m].x + wf[5] * xb[m].y + wf[6] * xb[m].z + wf[7] * xb[m].w;
            }
        }
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int nf = n + 4 * f;
#pragma unroll
            for (int m = 0; m < MR; ++m) {
                float v = wave_sum(acc[f][m]);
                if (lane == 0 && m < p.M && nf < n_end) {
                    if (p.bias) v += p.bias[nf];
                    if (p.silu_out) v = silu(v);
                    p.out[(size_t)m * p.N + nf] = v;
                }
            }
        }
    }
}

// denoiser.py:46-67: emb[b] = [cos(t f_0..f_127), sin(t f_0..f_127)], f_i = exp(-ln(10000) i / 128)
__global__ void timestep_freq_kernel(const int64_t* t, float* emb, int B) {
    const int b = blockIdx.x, i = threadIdx.x;   // 128 threads
    if (b >= B) return;
    const float f = expf(-9.210340371976184f * (float)i / 128.0f);
    const float a = (float)t[b] * f;
    emb[b * 256 + i] = cosf(a);
    emb[b * 256 + 128 + i] = sinf(a);
}

// ------------------------------------------------------------------------------------------------
// Ray embedding + patchify.  One thread per pixel: 9 channels -> 18 contiguous bytes of the token's GEMM row
// (column order (ph pw c), denoiser.py:211-215).  Token row = b * lpad + v * (H/ps)(W/ps) + hh * (W/ps) + ww.
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void embed_patchify_kernel(EmbedParams p) {
    const size_t HW = (size_t)p.H * p.W;
    const size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= (size_t)p.B * p.V * HW) return;
    const int w = (int)(pix % p.W), h = (int)((pix / p.W) % p.H);
    const int bv = (int)(pix / HW), v = bv % p.V, b = bv / p.V;
    const size_t base = (size_t)bv * 3 * HW + (size_t)h * p.W + w;
    const float r = p.images[base], g = p.images[base + HW], bl = p.images[base + 2 * HW];
    const float ox = p.ray_o[base], oy = p.ray_o[base + HW], oz = p.ray_o[base + 2 * HW];
    const float dx = p.ray_d[base], dy = p.ray_d[base + HW], dz = p.ray_d[base + 2 * HW];
    float c[9];
    c[0] = r * 2.0f - 1.0f; c[1] = g * 2.0f - 1.0f; c[2] = bl * 2.0f - 1.0f;
    if (p.relative_plk) {   // denoiser.py:316-322: cat(rgb, ray_d, ray_o + sum(-o d) d)
        const float od = (-ox * dx) + (-oy * dy) + (-oz * dz);
        c[3] = dx; c[4] = dy; c[5] = dz;
        c[6] = ox + od * dx; c[7] = oy + od * dy; c[8] = oz + od * dz;
    } else {                // denoiser.py:323-327: cat(rgb, cross(o, d), ray_d)
        c[3] = oy * dz - oz * dy; c[4] = oz * dx - ox * dz; c[5] = ox * dy - oy * dx;
        c[6] = dx; c[7] = dy; c[8] = dz;
    }
    const int ps = p.ps, np_w = p.W / ps, np = (p.H / ps) * np_w;
    const size_t row = (size_t)b * p.lpad + (size_t)v * np + (size_t)(h / ps) * np_w + (w / ps);
    bf16_t* dst = p.out + row * (size_t)(9 * ps * ps) + (size_t)((h % ps) * ps + (w % ps)) * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) dst[k] = (bf16_t)f2bf(c[k]);
}

// rows [L-n_g, L) of every sample <- gaussians_pos_embedding (denoiser.py:341-344)
__global__ void pos_embed_kernel(const float* pe, float* x, int B, int lpad, int L, int ng, int width) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * ng * width) return;
    const int c = i % width, g = (i / width) % ng, b = i / (width * ng);
    x[((size_t)b * lpad + (L - ng) + g) * width + c] = pe[g * width + c];
}

// internal rows -> reference token order [B, L, width]: [gaussian tokens, image tokens]
__global__ void gather_tokens_kernel(const float* x, float* out, int B, int lpad, int L, int ng, int width) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * L * width) return;
    const int c = (int)(i % width);
    const int tk = (int)((i / width) % L), b = (int)(i / ((size_t)width * L));
    const int src = tk < ng ? (L - ng) + tk : tk - ng;
    out[i] = x[((size_t)b * lpad + src) * width + c];
}

// reference token order [B, L, width] -> internal rows (inverse of the above; padding rows are not touched)
__global__ void scatter_tokens_kernel(const float* in, float* x, int B, int lpad, int L, int ng, int width) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * L * width) return;
    const int c = (int)(i % width);
    const int tk = (int)((i / width) % L), b = (int)(i / ((size_t)width * L));
    const int dst = tk < ng ? (L - ng) + tk : tk - ng;
    x[((size_t)b * lpad + dst) * width + c] = in[i];
}

// ------------------------------------------------------------------------------------------------
// to_gs + hard pixel alignment.  One thread per Gaussian.
//   image Gaussians: raw[B*lpad rows][ps*ps*C] f32 (decoder GEMM output), Gaussian (v, hh, ww, ph, pw)
//   learned Gaussians: up[B*ng][C] f32
// ------------------------------------------------------------------------------------------------

// NF = 3 (gaussians_sh_degree + 1)^2 feature channels between xyz and scaling (to_gs, denoiser.py:103-120): C = 11 + NF
template <int NF>
__global__ __launch_bounds__(256) void gaussians_kernel(GsParams p) {
    constexpr int C = 11 + NF;
    const size_t HW = (size_t)p.H * p.W;
    const size_t P = (size_t)p.ng + (size_t)p.V * HW;
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (size_t)p.B * P) return;
    const int b = (int)(gid / P);
    const size_t i = gid % P;
    float c[C];
    float x, y, z;
    if (i < (size_t)p.ng) {
        const float* src = p.up + ((size_t)b * p.ng + i) * C;
#pragma unroll
        for (int k = 0; k < C; ++k) c[k] = src[k];
        x = c[0]; y = c[1]; z = c[2];
    } else {
        const size_t j = i - p.ng;                       // (v, hh, ww, ph, pw)
        const int pp = p.ps * p.ps;
        const size_t tok = j / pp;
        const int pi = (int)(j % pp);
        const float* src = p.dec + ((size_t)b * p.lpad + tok) * (size_t)(pp * C) + (size_t)pi * C;
#pragma unroll
        for (int k = 0; k < C; ++k) c[k] = src[k];
        const int np_w = p.W / p.ps, np = (p.H / p.ps) * np_w;
        const int v = (int)(tok / np), hh = (int)((tok % np) / np_w), ww = (int)(tok % np_w);
        const int h = hh * p.ps + pi / p.ps, w = ww * p.ps + pi % p.ps;
        const size_t base = ((size_t)b * p.V + v) * 3 * HW + (size_t)h * p.W + w;
        const float ox = p.ray_o[base], oy = p.ray_o[base + HW], oz = p.ray_o[base + 2 * HW];
        const float dx = p.ray_d[base], dy = p.ray_d[base + HW], dz = p.ray_d[base + 2 * HW];
        float depth = (c[0] + c[1] + c[2]) / 3.0f;       // .mean(dim=2)
        depth = 1.0f / (1.0f + __expf(-depth));
        if (p.scene) depth = depth * (p.range_far - p.range_near) + p.range_near;   // denoiser_scene.py:407-410
        else if (p.relative_plk) depth = (2.0f * depth - 1.0f) * 1.8f + ((-ox * dx) + (-oy * dy) + (-oz * dz));
        x = ox + depth * dx; y = oy + depth * dy; z = oz + depth * dz;
        if (p.aligned) { p.aligned[base] = x; p.aligned[base + HW] = y; p.aligned[base + 2 * HW] = z; }
    }
    p.xyz[gid * 3] = x; p.xyz[gid * 3 + 1] = y; p.xyz[gid * 3 + 2] = z;
#pragma unroll
    for (int k = 0; k < NF; ++k) p.features[gid * NF + k] = c[3 + k];          // [.., (deg + 1)^2, 3] row-major = the split's order
    p.scaling[gid * 3] = fminf(c[3 + NF] - 2.3f, -1.2f);
    p.scaling[gid * 3 + 1] = fminf(c[4 + NF] - 2.3f, -1.2f);
    p.scaling[gid * 3 + 2] = fminf(c[5 + NF] - 2.3f, -1.2f);
    p.rotation[gid * 4] = c[6 + NF]; p.rotation[gid * 4 + 1] = c[7 + NF]; p.rotation[gid * 4 + 2] = c[8 + NF]; p.rotation[gid * 4 + 3] = c[9 + NF];
    p.opacity[gid] = c[10 + NF] - 2.0f;
}

static int launch_ok() { return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE; }

int launch_layernorm(const DgsDitLayerNormArgs* a, hipStream_t st) {
    if (!a || a->rows <= 0 || !a->x || !a->out || a->width % 256 || a->width > 2048 || a->width <= 0) return DGS_ERR_INVALID_ARGUMENT;
    if ((a->shift == nullptr) != (a->scale == nullptr)) return DGS_ERR_INVALID_ARGUMENT;
    LnParams p;
    p.rows = a->rows; p.width = a->width; p.mod_stride = a->mod_stride;
    p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : a->rows;
    p.out_f32 = a->out_f32; p.eps = a->eps; p.x = a->x; p.weight = a->weight; p.shift = a->shift; p.scale = a->scale; p.out = a->out;
    const dim3 grid((p.rows_per_batch + 3) / 4, (a->rows + p.rows_per_batch - 1) / p.rows_per_batch), block(256);
    const bool w = a->weight != nullptr, m = a->shift != nullptr, f = a->out_f32 != 0;
#define DGS_LN_CASE(V)                                                                                                                   \
    case V:                                                                                                                              \
        if (w) { if (m) { if (f) hipLaunchKernelGGL((layernorm_kernel<V, true, true, true>), grid, block, 0, st, p);                     \
                          else hipLaunchKernelGGL((layernorm_kernel<V, true, true, false>), grid, block, 0, st, p); }                    \
                 else   { if (f) hipLaunchKernelGGL((layernorm_kernel<V, true, false, true>), grid, block, 0, st, p);                    \
                          else hipLaunchKernelGGL((layernorm_kernel<V, true, false, false>), grid, block, 0, st, p); } }                 \
        else   { if (m) { if (f) hipLaunchKernelGGL((layernorm_kernel<V, false, true, true>), grid, block, 0, st, p);                    \
                          else hipLaunchKernelGGL((layernorm_kernel<V, false, true, false>), grid, block, 0, st, p); }                   \
                 else   { if (f) hipLaunchKernelGGL((layernorm_kernel<V, false, false, true>), grid, block, 0, st, p);                   \
                          else hipLaunchKernelGGL((layernorm_kernel<V, false, false, false>), grid, block, 0, st, p); } }                \
        break;
    switch (a->width / 256) {
        DGS_LN_CASE(1) DGS_LN_CASE(2) DGS_LN_CASE(3) DGS_LN_CASE(4) DGS_LN_CASE(8)
        default: return DGS_ERR_INVALID_ARGUMENT;
    }
#undef DGS_LN_CASE
    return launch_ok();
}

bool layernorm_rows_gemv_ok(const DgsDitLayerNormArgs* a, const DgsDitGemmArgs* g) {
    if (!a || !g || a->weight || !a->shift || !a->scale || a->out_f32) return false;
    if (a->width != 512 && a->width != 1024 && a->width != 2048) return false;
    if (g->epilogue != DGS_EPI_QKV && g->epilogue != DGS_EPI_GELU_BF16) return false;
    const int rpb = a->rows_per_batch > 0 ? a->rows_per_batch : a->rows;
    if (g->K != a->width || g->A != a->out || g->lda != a->width || g->M != a->rows || g->rows_per_batch != rpb || g->N % 32) return false;
    if (g->valid_rows <= 0 || g->valid_rows >= rpb) return false;
    const int live = g->valid_rows - (g->valid_rows - 1) / 256 * 256;            // live rows of the last tile row that has any
    return live <= 2 && g->valid_rows > 256;
}

int launch_layernorm_rows_gemv(const DgsDitLayerNormArgs* a, const DgsDitGemmArgs* g, hipStream_t st) {
    if (!layernorm_rows_gemv_ok(a, g) || a->rows <= 0 || !a->x || !a->out) return DGS_ERR_INVALID_ARGUMENT;
    LnParams p;
    p.rows = a->rows; p.width = a->width; p.mod_stride = a->mod_stride;
    p.rows_per_batch = a->rows_per_batch > 0 ? a->rows_per_batch : a->rows;
    p.out_f32 = 0; p.eps = a->eps; p.x = a->x; p.weight = nullptr; p.shift = a->shift; p.scale = a->scale; p.out = a->out;
    LnRowsGemv j;
    j.W = g->W; j.bias = g->bias; j.out = g->out; j.aux = g->aux; j.vt = g->vt; j.N = g->N; j.ldw = g->ldw; j.ldo = g->ldo; j.epilogue = g->epilogue;
    j.nrows = g->valid_rows - (g->valid_rows - 1) / 256 * 256; j.row0 = g->valid_rows - j.nrows;
    j.q_scale = g->q_scale != 0.0f ? g->q_scale : 1.0f;
    const int ks = a->width / 512 < 4 ? a->width / 512 : 4, cpi = 8 * (4 / ks);
    j.items = g->N / cpi;
    const dim3 grid(j.items + (p.rows_per_batch + 3) / 4, (a->rows + p.rows_per_batch - 1) / p.rows_per_batch), block(256);
#define DGS_LNG_CASE(V)                                                                                                                  \
    case V:                                                                                                                              \
        if (g->epilogue == DGS_EPI_QKV) hipLaunchKernelGGL((layernorm_rows_gemv_kernel<V, DGS_EPI_QKV>), grid, block, 0, st, p, j);       \
        else hipLaunchKernelGGL((layernorm_rows_gemv_kernel<V, DGS_EPI_GELU_BF16>), grid, block, 0, st, p, j);                            \
        break;
    switch (a->width / 256) {
        DGS_LNG_CASE(2) DGS_LNG_CASE(4) DGS_LNG_CASE(8)
        default: return DGS_ERR_INVALID_ARGUMENT;
    }
#undef DGS_LNG_CASE
    return launch_ok();
}

int launch_rowlinear(const DgsDitRowLinearArgs* a, hipStream_t st) {
    if (!a || a->M <= 0 || a->M > 16 || a->N <= 0 || a->K <= 0 || a->K % 8 || !a->x || !a->W || !a->out) return DGS_ERR_INVALID_ARGUMENT;
    RowLinParams p;
    p.M = a->M; p.N = a->N; p.K = a->K; p.silu_in = a->silu_input; p.silu_out = a->silu_output;
    p.x = a->x; p.W = a->W; p.bias = a->bias; p.out = a->out;
    const dim3 grid((a->N + kRowLinFeatures - 1) / kRowLinFeatures), block(256);
    if (a->M <= 1) hipLaunchKernelGGL((rowlinear_kernel<1>), grid, block, (size_t)1 * a->K * 4, st, p);
    else if (a->M <= 2) hipLaunchKernelGGL((rowlinear_kernel<2>), grid, block, (size_t)2 * a->K * 4, st, p);
    else if (a->M <= 4) hipLaunchKernelGGL((rowlinear_kernel<4>), grid, block, (size_t)4 * a->K * 4, st, p);
    else if (a->M <= 8) hipLaunchKernelGGL((rowlinear_kernel<8>), grid, block, (size_t)8 * a->K * 4, st, p);
    else {
        if ((size_t)16 * a->K * 4 > 65536) return DGS_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL((rowlinear_kernel<16>), grid, block, (size_t)16 * a->K * 4, st, p);
    }
    return launch_ok();
}

__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* dst, size_t n16) {
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = z;
}

int launch_zero_fill(void* dst, size_t bytes, hipStream_t st) {
    if (bytes == 0) return DGS_OK;
    if ((bytes & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return DGS_ERR_INVALID_ARGUMENT;
    const size_t n16 = bytes / 16;
    const unsigned blocks = (unsigned)((n16 + 255) / 256 < 2048 ? (n16 + 255) / 256 : 2048);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, st, static_cast<uint4*>(dst), n16);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

int launch_timestep(const int64_t* t, float* emb, int B, hipStream_t st) {
    hipLaunchKernelGGL(timestep_freq_kernel, dim3(B), dim3(128), 0, st, t, emb, B);
    return launch_ok();
}

int launch_embed(const EmbedParams& p, hipStream_t st) {
    const size_t n = (size_t)p.B * p.V * p.H * p.W;
    hipLaunchKernelGGL(embed_patchify_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    return launch_ok();
}

int launch_pos_embed(const float* pe, float* x, int B, int lpad, int L, int ng, int width, hipStream_t st) {
    hipLaunchKernelGGL(pos_embed_kernel, dim3((B * ng * width + 255) / 256), dim3(256), 0, st, pe, x, B, lpad, L, ng, width);
    return launch_ok();
}

int launch_gather_tokens(const float* x, float* out, int B, int lpad, int L, int ng, int width, hipStream_t st) {
    const size_t n = (size_t)B * L * width;
    hipLaunchKernelGGL(gather_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, out, B, lpad, L, ng, width);
    return launch_ok();
}

int launch_scatter_tokens(const float* in, float* x, int B, int lpad, int L, int ng, int width, hipStream_t st) {
    const size_t n = (size_t)B * L * width;
    hipLaunchKernelGGL(scatter_tokens_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, x, B, lpad, L, ng, width);
    return launch_ok();
}

int launch_gaussians(const GsParams& p, hipStream_t st) {
    const size_t n = (size_t)p.B * ((size_t)p.ng + (size_t)p.V * p.H * p.W);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    switch (p.C) {                                             // 11 + 3 (deg + 1)^2, gaussians_sh_degree 0 .. 3
        case 14: hipLaunchKernelGGL(gaussians_kernel<3>, grid, block, 0, st, p); break;
        case 23: hipLaunchKernelGGL(gaussians_kernel<12>, grid, block, 0, st, p); break;
        case 38: hipLaunchKernelGGL(gaussians_kernel<27>, grid, block, 0, st, p); break;
        case 59: hipLaunchKernelGGL(gaussians_kernel<48>, grid, block, 0, st, p); break;
        default: return DGS_ERR_INVALID_ARGUMENT;
    }
    return launch_ok();
}

}  // namespace dgs

extern "C" int dgs_dit_layernorm(const DgsDitLayerNormArgs* a, dgs_stream_t stream) {
    return dgs::launch_layernorm(a, static_cast<hipStream_t>(stream));
}
extern "C" int dgs_dit_rowlinear(const DgsDitRowLinearArgs* a, dgs_stream_t stream) {
    return dgs::launch_rowlinear(a, static_cast<hipStream_t>(stream));
}
