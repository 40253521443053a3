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
#ifdef HIPEMU
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) { return f2bf(lo) | (f2bf(hi) << 16); }
__device__ __forceinline__ float fast_exp2(float x) { return exp2f(x); }
__device__ __forceinline__ uint32_t f2bf_fast(float v) { return f2bf(v); }
#else
// one v_cvt_pk_bf16_f32 (round-to-nearest-even, same result as f2bf) instead of ~10 integer VALU ops per value
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint32_t f2bf_fast(float v) { return (uint32_t)__builtin_bit_cast(uint16_t, (__bf16)v); }
// bare v_exp_f32 (no denormal-range fix-up): callers only pass x <= ~8, tiny results may flush to 0
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
#endif

// two packed bf16 values times c, rounded back to bf16 (pre-scaled attention queries: c = +-scale * log2(e))
__device__ __forceinline__ uint32_t scale_bf2(uint32_t two, float c) {
    return pack_bf2(bf2f(two & 0xffffu) * c, bf2f(two >> 16) * c);
}

// 16-byte global -> LDS DMA.  `lds_wave_base` must be wave-uniform: lane i's 16 bytes land at base + 16*i.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
#ifdef HIPEMU
    __builtin_amdgcn_global_load_lds(gsrc, lds_wave_base, 16, 0, 0);
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

// Sum over the 64 lanes, valid in lane 63 only: six v_add_f32_dpp (row shifts inside the 16-lane rows, then row broadcasts) instead
// of six dependent ds_bpermute round trips (wave_sum below: ~400 cycles per sum; the GEMV side jobs do 16 of them).
__device__ __forceinline__ float wave_sum_lane63(float v) {
#ifdef HIPEMU
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
#else
#define DGS_DPP_ADD(ctrl) v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, true))
    DGS_DPP_ADD(0x111);   // row_shr:1
    DGS_DPP_ADD(0x112);   // row_shr:2
    DGS_DPP_ADD(0x114);   // row_shr:4
    DGS_DPP_ADD(0x118);   // row_shr:8    -> lane 15 of every 16-lane row holds the row sum
    DGS_DPP_ADD(0x142);   // row_bcast:15 -> lanes 31 / 63 hold the sum of their row pair
    DGS_DPP_ADD(0x143);   // row_bcast:31 -> lane 63 holds the wave sum
#undef DGS_DPP_ADD
    return v;
#endif
}

// Sum over the 64 lanes, in every lane: the DPP sum of lane 63 read back as a scalar (six ds_bpermute round trips -- ~400 cycles of
// latency per sum -- before: LayerNorm does two per row, the adaLN GEMV one per output feature).
__device__ __forceinline__ float wave_sum(float v) {
#ifdef HIPEMU
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
#else
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_lane63(v)), 63));
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Combine a value with the one held by lane ^ 32.  v_permlane32_swap exchanges the upper half of its first operand
// with the lower half of its second (VALU, no LDS round trip like ds_bpermute / __shfl_xor).
#ifdef HIPEMU
__device__ __forceinline__ float xor32_max(float v) { return fmaxf(v, __shfl_xor(v, 32)); }
__device__ __forceinline__ float xor32_sum(float v) { return v + __shfl_xor(v, 32); }
#else
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
#endif

// Exchange between the two halves of a wave (v_permlane32_swap): afterwards lanes 0-31 hold {their own a, the upper half's a} in
// (a, b) and lanes 32-63 hold {the lower half's b, their own b} -- two 8-byte row pieces of a lane pair become one 16-byte piece
// per lane (cdna_hip_programming.md T21: row-per-lane epilogue stores as dwordx4 instead of 2 x dwordx2).
__device__ __forceinline__ void half_swap(uint32_t& a, uint32_t& b) {
#ifdef HIPEMU
    const uint32_t oa = (uint32_t)__shfl_xor((int)a, 32), ob = (uint32_t)__shfl_xor((int)b, 32);
    if (threadIdx.x & 32) a = ob; else b = oa;
#else
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0]; b = r[1];
#endif
}

// XCD-aware remap of a linear workgroup id (cdna_hip_programming.md T1, bijective form): hardware places block b on
// XCD b % 8; give every XCD a contiguous range of logical ids so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

}  // namespace dgs
