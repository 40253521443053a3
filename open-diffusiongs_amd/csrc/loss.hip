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
