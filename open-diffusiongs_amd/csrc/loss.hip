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

}  // namespace dgs

extern "C" int dgs_mse_psnr(const DgsMseArgs* a, dgs_stream_t stream) {
    if (!a || a->B <= 0 || a->n <= 0 || a->n % 4 || !a->rendering || !a->target || !a->l2 || !a->partial) return DGS_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(dgs::mse_partial_kernel, dim3(DGS_LOSS_CHUNKS, a->B), dim3(256), 0, st, *a);
    hipLaunchKernelGGL(dgs::mse_final_kernel, dim3((a->B + 63) / 64), dim3(64), 0, st, *a);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
