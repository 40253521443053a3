// dgs_device.h -- device-side helpers shared by the gfx950 kernels.
//
// Arithmetic contract (DESIGN.md "bit-exactness"): the rasterizer translation units are compiled with
// -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt; a*b+c is two roundings unless written as
// __builtin_fmaf.  Everything that feeds an integer decision (depth key, radius, tile rect, alpha
// thresholds) is therefore a fixed sequence of IEEE-754 operations, reproducible on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifdef HIPEMU
// CPU emulation build of the test suite (tests/hipemu): dynamic LDS comes from the emulator.
#define DGS_DYNAMIC_LDS(name) char* name = hipemu::dyn_lds()
#else
#define DGS_DYNAMIC_LDS(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

#define DGS_WAVE 64

namespace dgs {

__device__ __forceinline__ int f2i_sat(float v) {
    // v_cvt_i32_f32 saturates and maps NaN to 0; spelled out so host restatements can match it.
    if (!(v == v)) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (-2147483647 - 1);
    return (int)v;
}

// Deterministic expf: Cody-Waite reduction + degree-6 polynomial in an fma chain + exact 2^n scaling.
// Lowers to v_mul, v_rndne, 8x v_fma, v_add, v_cvt_i32, v_ldexp -- every one IEEE-exact, so the CPU
// oracle (oracle/raster_oracle.cpp det_expf) reproduces it bit for bit.  < 1.5 ulp.
__device__ __forceinline__ float det_expf(float x) {
    if (!(x == x)) return x;
    if (x < -87.0f) return 0.0f;
    if (x > 88.0f) return __builtin_inff();
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    return __builtin_ldexpf(y, (int)n);
}

// det_expf without its three range checks, for callers that know x is a number in [-87, 88] (the blend kernels: a pair that
// passes the alpha cut-off has -5.6 < power <= 0): the same instructions, the same bits.
__device__ __forceinline__ float det_expf_core(float x) {
    const float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    float y = __builtin_fmaf(p, r2, r);
    y = y + 1.0f;
    return __builtin_ldexpf(y, (int)n);
}

// One v_exp_f32 (2^x, ~1 ulp, no denormal fix-up): the hardware exponential.  The CPU emulation build uses libm.
__device__ __forceinline__ float hw_exp2(float x) {
#ifdef HIPEMU
    return exp2f(x);
#else
    return __builtin_amdgcn_exp2f(x);
#endif
}

// One v_rcp_f32 (1 ulp); the emulation build divides.
__device__ __forceinline__ float hw_rcp(float x) {
#ifdef HIPEMU
    return 1.0f / x;
#else
    return __builtin_amdgcn_rcpf(x);
#endif
}

// exp(power) of a (pixel, Gaussian) pair inside the blend loops (-5.6 < power <= 0 for every pair that passes the cut-off).
// FAST (the product default): v_mul + v_exp_f32 -- what the reference's CUDA `exp()` under fast math is on its hardware, two VALU
//   instead of fourteen in the innermost loop of both blend kernels; integer artefacts that do not depend on alpha (radii, tile
//   lists, ranges, sort order) are unaffected, colours move by a few fp32 ulp.
// exact: det_expf_core, the fixed IEEE sequence the oracle restates (exp_mode 1): every float bit-identical with the oracle.
template <bool FAST>
__device__ __forceinline__ float blend_exp(float power) {
    if constexpr (FAST) return hw_exp2(power * 1.44269504088896341f);
    else return det_expf_core(power);
}

// Exclusive scan of one value per thread over a block of `NT` threads (NT multiple of 64, <= 1024).
// `scratch` needs NT/64 + 1 uint32 of LDS.  Returns the exclusive prefix; *total receives the block sum.
template <int NT>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* scratch, uint32_t* total) {
    const int lane = threadIdx.x & (DGS_WAVE - 1), wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < DGS_WAVE; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == DGS_WAVE - 1) scratch[wave] = inc;
    __syncthreads();
    uint32_t base = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < NT / DGS_WAVE; ++w) {
        const uint32_t s = scratch[w];
        if (w < wave) base += s;
        sum += s;
    }
    __syncthreads();
    *total = sum;
    return base + inc - v;
}

}  // namespace dgs
