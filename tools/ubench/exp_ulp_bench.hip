// exp_ulp_bench: what the blend loops' exponential (csrc/dgs_device.h blend_exp<true>) is worth on the device, in units in the last
// place against exp() evaluated in double: a dense sweep of power over (-8, 0] (every pair that passes the alpha cut-off has
// -5.6 < power <= 0), for (a) the plain v_exp_f32(power * log2e), (b) the compensated form the product ships, (c) det_expf_core.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I open-diffusiongs_amd/csrc tools/ubench/exp_ulp_bench.hip -o tools/ubench/exp_ulp_bench
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#include "dgs_device.h"

__global__ void sweep(const float* x, float* plain, float* comp, float* det, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;                                     // n is a multiple of the block: whole waves reach the ballot inside blend_exp
    plain[i] = dgs::hw_exp2(x[i] * 1.44269504088896341f);
    comp[i] = dgs::blend_exp<true>(x[i], 0.5f);             // opacity 0.5: alpha = 1/255 needs power = -4.85; the guard band is one point of the sweep at most
    det[i] = dgs::det_expf_core(x[i]);
}

int main() {
    const int n = 1 << 24;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = -8.0f * (float)i / (float)n;
    float *dx, *dp, *dc, *dd;
    hipMalloc(&dx, n * 4); hipMalloc(&dp, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sweep, dim3(n / 256), dim3(256), 0, 0, dx, dp, dc, dd, n);
    std::vector<float> p(n), c(n), d(n);
    hipMemcpy(p.data(), dp, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
    double worst[3] = {0, 0, 0}, sum[3] = {0, 0, 0}, worst_hi[3] = {0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const double r = exp((double)x[i]);
        const float rf = (float)r;
        const double ulp = (double)nextafterf(rf, 2.0f) - (double)rf;
        const float* got[3] = {&p[i], &c[i], &d[i]};
        for (int k = 0; k < 3; ++k) {
            const double e = fabs((double)*got[k] - r) / ulp;
            if (e > worst[k]) worst[k] = e;
            if (x[i] < -4.0f && e > worst_hi[k]) worst_hi[k] = e;
            sum[k] += e;
        }
    }
    const char* names[3] = {"v_exp_f32(power * log2e)          ", "compensated (product default)     ", "det_expf_core (oracle's sequence) "};
    printf("# exp(power), power in (-8, 0], %d points, error in ulp of the fp32 result against double exp()\n", n);
    for (int k = 0; k < 3; ++k) printf("%s max %.2f ulp   max for power < -4: %.2f   mean %.3f\n", names[k], worst[k], worst_hi[k], sum[k] / n);
    return 0;
}
