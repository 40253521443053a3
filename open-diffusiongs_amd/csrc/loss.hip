// loss.hip -- per-sample MSE / PSNR (+ the MSE gradient) in one pass (include/dgs_loss.h).  HBM-bound: two streams in,
// optionally one out, 16 bytes per lane.  Reduction order is fixed: thread-strided partial sums, a block tree, then the 64
// chunk sums of a sample added in index order -- the result does not depend on scheduling.
#include <hip/hip_runtime.h>

#include "dgs_device.h"
#include "dgs_loss.h"

namespace dgs {

__global__ __launch_bounds__(256) void mse_partial_kernel(DgsMseArgs a) {
    __shared__ float red[256];
    const int b = blockIdx.y, chunk = blockIdx.x;
    const long long n4 = a.n / 4, per = (n4 + DGS_LOSS_CHUNKS - 1) / DGS_LOSS_CHUNKS;
    const long long lo = (long long)chunk * per, hi = lo + per < n4 ? lo + per : n4;
    const float4* r = reinterpret_cast<const float4*>(a.rendering + (size_t)b * a.n);
    const float4* t = reinterpret_cast<const float4*>(a.target + (size_t)b * a.n);
    float4* g = a.grad ? reinterpret_cast<float4*>(a.grad + (size_t)b * a.n) : nullptr;
    const float gs = a.grad_scale * 2.0f / (float)a.n;
    float acc = 0.0f;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        float4 x = r[i], y = t[i];
        if (a.clamp01) {
            x.x = fminf(fmaxf(x.x, 0.f), 1.f); x.y = fminf(fmaxf(x.y, 0.f), 1.f); x.z = fminf(fmaxf(x.z, 0.f), 1.f); x.w = fminf(fmaxf(x.w, 0.f), 1.f);
            y.x = fminf(fmaxf(y.x, 0.f), 1.f); y.y = fminf(fmaxf(y.y, 0.f), 1.f); y.z = fminf(fmaxf(y.z, 0.f), 1.f); y.w = fminf(fmaxf(y.w, 0.f), 1.f);
        }
        const float4 d = make_float4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
        acc += d.x * d.x; acc += d.y * d.y; acc += d.z * d.z; acc += d.w * d.w;
        if (g) g[i] = make_float4(gs * d.x, gs * d.y, gs * d.z, gs * d.w);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.partial[(size_t)b * DGS_LOSS_CHUNKS + chunk] = red[0];
}

__global__ void mse_final_kernel(DgsMseArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    float s = 0.0f;
    for (int c = 0; c < DGS_LOSS_CHUNKS; ++c) s += a.partial[(size_t)b * DGS_LOSS_CHUNKS + c];
    const float l2 = s / (float)a.n;
    a.l2[b] = l2;
    if (a.psnr) a.psnr[b] = -10.0f * log10f(l2);
}

// ---- F.interpolate(x, size, mode='bilinear', align_corners=False) * mul + add  (losses.py:304-309: the LPIPS input) ----
// PyTorch's area_pixel_compute_source_index: src = max(0, (dst + 0.5) * in / out - 0.5); i0 = floor(src), i1 = min(i0 + 1,
// in - 1), l1 = src - i0, l0 = 1 - l1;  value = h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11).
struct ResizeCoef { int i0, i1; float l0, l1; };
__device__ __forceinline__ ResizeCoef resize_coef(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    src = src < 0.f ? 0.f : src;
    ResizeCoef c;
    c.i0 = (int)src;
    c.i1 = c.i0 + (c.i0 < in - 1 ? 1 : 0);
    c.l1 = src - (float)c.i0;
    c.l0 = 1.0f - c.l1;
    return c;
}

__global__ __launch_bounds__(256) void resize_bilinear_kernel(DgsResizeArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;       // one output element per thread, x fastest: coalesced stores
    const long long total = (long long)a.planes * a.out_h * a.out_w;
    if (i >= total) return;
    const int ox = (int)(i % a.out_w), oy = (int)((i / a.out_w) % a.out_h);
    const long long pl = i / ((long long)a.out_w * a.out_h);
    const ResizeCoef cx = resize_coef(ox, a.in_w, a.out_w), cy = resize_coef(oy, a.in_h, a.out_h);
    const float* s = a.src + pl * a.in_h * a.in_w;
    const float v00 = s[(size_t)cy.i0 * a.in_w + cx.i0], v01 = s[(size_t)cy.i0 * a.in_w + cx.i1];
    const float v10 = s[(size_t)cy.i1 * a.in_w + cx.i0], v11 = s[(size_t)cy.i1 * a.in_w + cx.i1];
    const float v = cy.l0 * (cx.l0 * v00 + cx.l1 * v01) + cy.l1 * (cx.l0 * v10 + cx.l1 * v11);
    a.dst[i] = v * a.mul + a.add;
}

// d src += mul * weights * d dst.  Atomics: for the reference's two cases (same size; 512 -> 256, where every input pixel feeds
// exactly one output) each address receives one contribution at most per weight, so the result is order-independent.
__global__ __launch_bounds__(256) void resize_bilinear_backward_kernel(DgsResizeArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.planes * a.out_h * a.out_w;
    if (i >= total) return;
    const int ox = (int)(i % a.out_w), oy = (int)((i / a.out_w) % a.out_h);
    const long long pl = i / ((long long)a.out_w * a.out_h);
    const ResizeCoef cx = resize_coef(ox, a.in_w, a.out_w), cy = resize_coef(oy, a.in_h, a.out_h);
    float* s = a.dsrc + pl * a.in_h * a.in_w;
    const float g = a.ddst[i] * a.mul;
    atomicAdd(s + (size_t)cy.i0 * a.in_w + cx.i0, cy.l0 * cx.l0 * g);
    atomicAdd(s + (size_t)cy.i0 * a.in_w + cx.i1, cy.l0 * cx.l1 * g);
    atomicAdd(s + (size_t)cy.i1 * a.in_w + cx.i0, cy.l1 * cx.l0 * g);
    atomicAdd(s + (size_t)cy.i1 * a.in_w + cx.i1, cy.l1 * cx.l1 * g);
}

// ------------------------------------------------------------------------------------------------------------------
// Points-distribution + xyz loss (dgs_loss.h DgsPointsLossArgs).  A (sample, view) plane of H W pixels is cut into
// DGS_LOSS_CHUNKS chunks; a workgroup reduces one chunk; chunk sums are added in index order by the finishing kernels.
// workspace layout (floats): stat [B V][CH][4] (sum dist, sum dist^2, masked squared error, mask sum) | moments [B V][2] (mean,
// 1 / (std + 1e-8)) | xyz totals [2] | pd [B V][CH] (sum (dist - target)^2)
// ------------------------------------------------------------------------------------------------------------------
struct PointsWs {
    float *stat, *moments, *xyz_tot, *pd;
    __host__ __device__ PointsWs(float* w, int BV) : stat(w), moments(w + (size_t)BV * DGS_LOSS_CHUNKS * 4), xyz_tot(moments + (size_t)BV * 2),
                                                     pd(xyz_tot + 2) {}
};

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void points_stats_kernel(DgsPointsLossArgs a) {
    __shared__ float red[256];
    const int bv = blockIdx.y, chunk = blockIdx.x;
    const long long HW = (long long)a.H * a.W, per = (HW + DGS_LOSS_CHUNKS - 1) / DGS_LOSS_CHUNKS;
    const long long lo = (long long)chunk * per, hi = lo + per < HW ? lo + per : HW;
    const float* p = a.aligned + (size_t)bv * 3 * HW;
    const float* o = a.ray_o + (size_t)bv * 3 * HW;
    float s1 = 0.f, s2 = 0.f, se = 0.f, sm = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const float dx = p[i] - o[i], dy = p[HW + i] - o[HW + i], dz = p[2 * HW + i] - o[2 * HW + i];
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        s1 += d; s2 += d * d;
        if (a.gt) {
            const float m = a.masks[(size_t)bv * HW + i];
            const float* g = a.gt + (size_t)bv * 3 * HW;
            const float ex = (p[i] - g[i]) * m, ey = (p[HW + i] - g[HW + i]) * m, ez = (p[2 * HW + i] - g[2 * HW + i]) * m;
            se += ex * ex + ey * ey + ez * ez;
            sm += m;
        }
    }
    s1 = block_sum_256(s1, red); s2 = block_sum_256(s2, red); se = block_sum_256(se, red); sm = block_sum_256(sm, red);
    if (threadIdx.x == 0) {
        float* st = PointsWs(a.workspace, a.B * a.V).stat + ((size_t)bv * DGS_LOSS_CHUNKS + chunk) * 4;
        st[0] = s1; st[1] = s2; st[2] = se; st[3] = sm;
    }
}

// one thread per (sample, view): mean / unbiased std of dist; thread 0 also totals the xyz sums over the batch
__global__ void points_moments_kernel(DgsPointsLossArgs a) {
    const int bv = blockIdx.x * blockDim.x + threadIdx.x, BV = a.B * a.V;
    PointsWs w(a.workspace, BV);
    if (bv < BV) {
        double s1 = 0.0, s2 = 0.0;
        for (int c = 0; c < DGS_LOSS_CHUNKS; ++c) { s1 += w.stat[((size_t)bv * DGS_LOSS_CHUNKS + c) * 4]; s2 += w.stat[((size_t)bv * DGS_LOSS_CHUNKS + c) * 4 + 1]; }
        const double n = (double)a.H * a.W, mean = s1 / n;
        const double var = n > 1.0 ? fmax((s2 - n * mean * mean) / (n - 1.0), 0.0) : 0.0;        // torch.std: correction = 1
        w.moments[2 * bv] = (float)mean;
        w.moments[2 * bv + 1] = (float)(1.0 / (sqrt(var) + 1e-8));
    }
    if (bv == 0 && a.gt) {
        double se = 0.0, sm = 0.0;
        for (int i = 0; i < BV * DGS_LOSS_CHUNKS; ++i) { se += w.stat[(size_t)i * 4 + 2]; sm += w.stat[(size_t)i * 4 + 3]; }
        w.xyz_tot[0] = (float)se; w.xyz_tot[1] = (float)sm;
        a.xyz[0] = (float)(se / sm);
    }
}

__global__ __launch_bounds__(256) void points_loss_kernel(DgsPointsLossArgs a) {
    __shared__ float red[256];
    const int bv = blockIdx.y, chunk = blockIdx.x, b = bv / a.V;
    const long long HW = (long long)a.H * a.W, per = (HW + DGS_LOSS_CHUNKS - 1) / DGS_LOSS_CHUNKS;
    const long long lo = (long long)chunk * per, hi = lo + per < HW ? lo + per : HW;
    PointsWs w(a.workspace, a.B * a.V);
    const float mean = w.moments[2 * bv], istd = w.moments[2 * bv + 1];
    const float* p = a.aligned + (size_t)bv * 3 * HW;
    const float* o = a.ray_o + (size_t)bv * 3 * HW;
    float* g = a.grad ? a.grad + (size_t)bv * 3 * HW : nullptr;
    const float wpd = (a.w_pointsdist ? a.w_pointsdist[b] : 0.f) * 2.0f / (float)((long long)a.V * HW);
    const float wx = a.gt ? a.w_xyz * 2.0f / w.xyz_tot[1] : 0.f;
    float acc = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const float ox = o[i], oy = o[HW + i], oz = o[2 * HW + i];
        const float dx = p[i] - ox, dy = p[HW + i] - oy, dz = p[2 * HW + i] - oz;
        const float d = sqrtf(dx * dx + dy * dy + dz * dz);
        const float target = (d - mean) * istd * 0.5f + sqrtf(ox * ox + oy * oy + oz * oz);
        const float e = d - target;
        acc += e * e;
        if (g) {
            const float k = d > 0.f ? wpd * e / d : 0.f;                  // d |x| / dx = x / |x| (0 at the origin, like torch.norm)
            float gx = k * dx, gy = k * dy, gz = k * dz;
            if (a.gt) {
                const float m = a.masks[(size_t)bv * HW + i];
                const float* t = a.gt + (size_t)bv * 3 * HW;
                const float mm = wx * m * m;
                gx += mm * (p[i] - t[i]); gy += mm * (p[HW + i] - t[HW + i]); gz += mm * (p[2 * HW + i] - t[2 * HW + i]);
            }
            g[i] = gx; g[HW + i] = gy; g[2 * HW + i] = gz;
        }
    }
    acc = block_sum_256(acc, red);
    if (threadIdx.x == 0) w.pd[(size_t)bv * DGS_LOSS_CHUNKS + chunk] = acc;
}

__global__ void points_final_kernel(DgsPointsLossArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    PointsWs w(a.workspace, a.B * a.V);
    double s = 0.0;
    for (int i = 0; i < a.V * DGS_LOSS_CHUNKS; ++i) s += w.pd[(size_t)b * a.V * DGS_LOSS_CHUNKS + i];
    a.pointsdist[b] = (float)(s / ((double)a.V * a.H * a.W));
}

}  // namespace dgs

extern "C" int dgs_resize_bilinear(const DgsResizeArgs* a, dgs_stream_t stream) {
    if (!a || a->planes <= 0 || a->in_h <= 0 || a->in_w <= 0 || a->out_h <= 0 || a->out_w <= 0 || !a->src || !a->dst) return DGS_ERR_INVALID_ARGUMENT;
    const long long total = (long long)a->planes * a->out_h * a->out_w;
    hipLaunchKernelGGL(dgs::resize_bilinear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

extern "C" int dgs_resize_bilinear_backward(const DgsResizeArgs* a, dgs_stream_t stream) {
    if (!a || a->planes <= 0 || a->in_h <= 0 || a->in_w <= 0 || a->out_h <= 0 || a->out_w <= 0 || !a->ddst || !a->dsrc) return DGS_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(a->dsrc, 0, sizeof(float) * (size_t)a->planes * a->in_h * a->in_w, st) != hipSuccess) return DGS_ERR_DEVICE;
    const long long total = (long long)a->planes * a->out_h * a->out_w;
    hipLaunchKernelGGL(dgs::resize_bilinear_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, *a);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

extern "C" int dgs_mse_psnr(const DgsMseArgs* a, dgs_stream_t stream) {
    if (!a || a->B <= 0 || a->n <= 0 || a->n % 4 || !a->rendering || !a->target || !a->l2 || !a->partial) return DGS_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(dgs::mse_partial_kernel, dim3(DGS_LOSS_CHUNKS, a->B), dim3(256), 0, st, *a);
    hipLaunchKernelGGL(dgs::mse_final_kernel, dim3((a->B + 63) / 64), dim3(64), 0, st, *a);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

extern "C" int64_t dgs_points_loss_workspace_floats(int32_t B, int32_t V) {
    if (B <= 0 || V <= 0) return 0;
    return (int64_t)B * V * DGS_LOSS_CHUNKS * 4 + (int64_t)B * V * 2 + 2 + (int64_t)B * V * DGS_LOSS_CHUNKS;
}

extern "C" int dgs_points_loss(const DgsPointsLossArgs* a, dgs_stream_t stream) {
    if (!a || a->B <= 0 || a->V <= 0 || a->H <= 0 || a->W <= 0 || !a->aligned || !a->ray_o || !a->pointsdist || !a->workspace)
        return DGS_ERR_INVALID_ARGUMENT;
    if (a->gt && (!a->masks || !a->xyz)) return DGS_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int BV = a->B * a->V;
    hipLaunchKernelGGL(dgs::points_stats_kernel, dim3(DGS_LOSS_CHUNKS, BV), dim3(256), 0, st, *a);
    hipLaunchKernelGGL(dgs::points_moments_kernel, dim3((BV + 63) / 64), dim3(64), 0, st, *a);
    hipLaunchKernelGGL(dgs::points_loss_kernel, dim3(DGS_LOSS_CHUNKS, BV), dim3(256), 0, st, *a);
    hipLaunchKernelGGL(dgs::points_final_kernel, dim3((a->B + 63) / 64), dim3(64), 0, st, *a);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
