// dit_common.h -- shared device helpers of the DiT (denoiser) kernels: bf16 packing, MFMA fragment types,
// LDS-DMA staging, wave reductions.  gfx950 / wave64 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dgs_device.h"

namespace dgs {

typedef short bf16x8 __attribute__((ext_vector_type(8)));     // MFMA A/B fragment: 8 bf16 (4 VGPRs)
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));    // 32x32 accumulator fragment
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;

// round-to-nearest-even f32 -> bf16 (NaN kept quiet)
__device__ __forceinline__ uint32_t f2bf(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) { return f2bf(lo) | (f2bf(hi) << 16); }

// 16-byte global -> LDS DMA.  `lds_wave_base` must be wave-uniform: lane i's 16 bytes land at base + 16*i.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
#ifdef HIPEMU
    __builtin_amdgcn_global_load_lds(gsrc, lds_wave_base, 16, 0, 0);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// XCD-aware remap of a linear workgroup id (cdna_hip_programming.md T1, bijective form): hardware places block b on
// XCD b % 8; give every XCD a contiguous range of logical ids so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

}  // namespace dgs
