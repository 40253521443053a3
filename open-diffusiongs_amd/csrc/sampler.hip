// sampler.hip -- one fused elementwise kernel for the diffusion sampler step (include/dgs_sampler.h).
// HBM-bound by construction: 3 streams in (model output, x_t, noise), 1-2 out, 16 bytes per lane, no tables on the host.
// Arithmetic order is the reference's (gaussian_diffusion.py:301-304, 504-512): c1 * x0 + c2 * x_t, then + sigma * noise, in
// fp32 without contraction (this file is built with -ffp-contract=off).
#include <hip/hip_runtime.h>

#include "dgs_device.h"
#include "dgs_sampler.h"

namespace dgs {

__global__ __launch_bounds__(256) void sampler_step_kernel(DgsSamplerStepArgs a) {
    const int b = blockIdx.y;
    const long long t = a.t[b];
    if (t < 0 || t >= a.T) {
        if (a.bad_t && threadIdx.x == 0 && blockIdx.x == 0) *a.bad_t = 1;
        return;
    }
    const float c1 = a.coef1[t], c2 = a.coef2[t], sg = t != 0 ? a.sigma[t] : 0.0f;
    const float* mo = a.model_output + (size_t)b * a.model_stride + a.model_offset;
    const size_t base = (size_t)b * a.per_sample;
    const long long n4 = a.per_sample / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 m = reinterpret_cast<const float4*>(mo)[i];
        const float4 x = reinterpret_cast<const float4*>(a.x_t + base)[i];
        float4 x0 = m;
        if (a.clip_denoised) {
            x0.x = fminf(fmaxf(m.x, -1.0f), 1.0f); x0.y = fminf(fmaxf(m.y, -1.0f), 1.0f);
            x0.z = fminf(fmaxf(m.z, -1.0f), 1.0f); x0.w = fminf(fmaxf(m.w, -1.0f), 1.0f);
        }
        float4 o = make_float4(c1 * x0.x + c2 * x.x, c1 * x0.y + c2 * x.y, c1 * x0.z + c2 * x.z, c1 * x0.w + c2 * x.w);
        if (t != 0 && a.noise) {
            const float4 z = reinterpret_cast<const float4*>(a.noise + base)[i];
            o.x = o.x + sg * z.x; o.y = o.y + sg * z.y; o.z = o.z + sg * z.z; o.w = o.w + sg * z.w;
        }
        reinterpret_cast<float4*>(a.out + base)[i] = o;
        if (a.pred_xstart) reinterpret_cast<float4*>(a.pred_xstart + base)[i] = x0;
    }
}

}  // namespace dgs

extern "C" int dgs_sampler_step(const DgsSamplerStepArgs* a, dgs_stream_t stream) {
    if (!a || a->B <= 0 || a->per_sample <= 0 || a->per_sample % 4 || a->model_offset % 4 || a->model_stride % 4 || a->T <= 0 ||
        !a->model_output || !a->x_t || !a->t || !a->coef1 || !a->coef2 || !a->sigma || !a->out)
        return DGS_ERR_INVALID_ARGUMENT;
    const long long n4 = a->per_sample / 4;
    const int gx = (int)((n4 + 255) / 256 < 1024 ? (n4 + 255) / 256 : 1024);
    hipLaunchKernelGGL(dgs::sampler_step_kernel, dim3(gx, a->B), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
