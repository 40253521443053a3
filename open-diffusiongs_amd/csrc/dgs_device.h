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

// min(0.99, x) of the blend loops (forward.cu:340 / backward.cu:476: CUDA's min returns the number when one operand is NaN, and so
// does v_min_f32): ONE v_min_f32.  fminf() compiles to v_max_f32 x, x (quieting a signalling NaN) + v_min_f32 -- a VALU per step.
__device__ __forceinline__ float alpha_clamp(float x) {
#ifdef HIPEMU
    return fminf(0.99f, x);
#else
    float r;
    asm("v_min_f32 %0, 0x3f7d70a4, %1" : "=v"(r) : "v"(x));
    return r;
#endif
}

// Mask of the wave's lanes whose predicate holds, straight from the compare (HIP's __ballot(int) takes the predicate as an integer:
// v_cndmask + v_cmp_ne per call in the blend loops).
__device__ __forceinline__ unsigned long long wave_ballot(bool pred) {
#ifdef HIPEMU
    return __ballot(pred ? 1 : 0);
#else
    return __builtin_amdgcn_ballot_w64(pred);
#endif
}

// exp(power) of a (pixel, Gaussian) pair inside the blend loops (-5.6 < power <= 0 for every pair that passes the cut-off); `opacity`
// is the factor the caller multiplies it with to get alpha.  The reference calls `exp(power)` of a plain nvcc build (its setup.py
// passes no --use_fast_math): CUDA's <= 2 ulp expf (forward.cu:332-358, backward.cu:463-532).
// FAST (the product default): the hardware's 2^x with the rounding error of its ARGUMENT compensated.  t = fl(power * log2e) is off by
//   up to |t| * 2^-24 (3-6 ulp of the result near power = -5.5, which is where the 1/255 cut-off sits); e = fma(power, L2E_HI, -t) +
//   power * L2E_LO is that error to first order, and exp = 2^t * (1 + e ln 2): mul, 2 fma, v_exp_f32, mul, fma -- six VALU instead of the
//   oracle sequence's fourteen; measured on the device over 2^24 points of (-8, 0]: max 1.26 ulp against 7.63 for the bare
//   v_exp_f32(power * log2e) and 1.01 for the oracle's sequence (tools/ubench/exp_ulp_bench.hip, profiles/r06_exp_ulp.txt).
//   A pair whose alpha lands within 1e-6 (relative) of the 1/255 cut-off -- closer than the two exponentials may differ -- is evaluated
//   with det_expf_core, so the product and the oracle (hence the strict build of the reference) put every pair on the SAME side of the
//   cut-off; a pair on the other side would move its Gaussian's gradient by the pair's whole contribution (round 5 measured 1.8e-3 of
//   a gradient tensor's max from such pairs with the bare v_exp_f32; 5e-6 with this form, profiles/r06_raster_grad_error.txt).  The
//   branch is wave-uniform, out of line, and taken about once per 10^5 wave steps.  Forward and backward call this one function with the same operands, so they agree on
//   which pairs were blended.
// exact: det_expf_core, the fixed IEEE sequence the oracle restates (exp_mode 1): every float bit-identical with the oracle.
template <bool FAST>
__device__ __forceinline__ float blend_exp(float power, float opacity) {
    if constexpr (FAST) {
        const float kL2eHi = 1.44269504088896341f;                                      // fl(log2 e)
        const float kL2eLo = (float)(1.44269504088896341 - (double)1.44269504088896341f);   // log2 e - fl(log2 e) = 1.9259630e-8
        const float t = power * kL2eHi;
        float e = __builtin_fmaf(power, kL2eHi, -t);
        e = __builtin_fmaf(power, kL2eLo, e);
        const float ex = hw_exp2(t);
        float G = __builtin_fmaf(ex, e * 0.693147180559945309f, ex);
        const float kBand = 1e-6f;
        const float kCut = 1.0f / 255.0f;
        const bool edge = __builtin_fabsf(opacity * G - kCut) < kCut * kBand;
        if (__builtin_expect(wave_ballot(edge) != 0ull, 0)) G = edge ? det_expf_core(power) : G;
        return G;
    } else {
        return det_expf_core(power);
    }
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
