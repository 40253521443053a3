// dit_common.h -- shared device helpers of the DiT (denoiser) kernels: bf16 packing, MFMA fragment types,
// LDS-DMA staging, wave reductions.  gfx950 / wave64 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

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
#endif
// bare v_exp_f32 (no denormal-range fix-up): callers only pass x <= ~8, tiny results may flush to 0
__device__ __forceinline__ float fast_exp2(float x) { return hw_exp2(x); }

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

// sum over each aligned group of 8 lanes, in all 8 (three DPP adds: the quad's two pair swaps, then the mirrored other quad)
__device__ __forceinline__ float oct_sum(float v) {
#ifdef HIPEMU
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    return v;
#else
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, true));    // quad_perm:[1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, true));    // quad_perm:[2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    return v;
#endif
}
// c + a.lo b.lo + a.hi b.hi of two packed bf16 pairs, fp32 accumulation (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
#ifdef HIPEMU
    return c + bf2f(a & 0xffffu) * bf2f(b & 0xffffu) + bf2f(a >> 16) * bf2f(b >> 16);
#else
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, a), __builtin_bit_cast(bf2_t, b), c, false);
#endif
}
__device__ __forceinline__ float dot8_bf16(uint4 a, uint4 b) {
    return dot2_bf16(a.w, b.w, dot2_bf16(a.z, b.z, dot2_bf16(a.y, b.y, dot2_bf16(a.x, b.x, 0.f))));
}

// compile-time loops: static_for<0, N>([&](auto ic) { constexpr int I = decltype(ic)::value; ... })
template <int I> struct IC { static constexpr int value = I; };
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IC<I>{}); static_for<I + 1, N>(f); }
}


// ---- execution primitives: the hardware form, and what tests/hipemu (cooperative fibers, synchronous memory) runs in its place.
//      The kernels themselves carry no #ifdef: every fork between the gfx950 build and the CPU emulation build lives in this file,
//      dgs_device.h and raster_common.h. ----
// counted wait on the vector-memory queue (LDS-DMAs complete in issue order: "at most N still in flight")
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#ifndef HIPEMU
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
__device__ __forceinline__ void wait_lgkmcnt0() {
#ifndef HIPEMU
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}
// workgroup barrier WITHOUT the memory-queue drain __syncthreads() carries (LDS-DMAs stay in flight across it)
__device__ __forceinline__ void raw_barrier() {
#ifndef HIPEMU
    __builtin_amdgcn_s_barrier();
#else
    __syncthreads();
#endif
}
// workgroup barrier that orders LDS accesses only: the wave's LDS operations have completed, its vector-memory queue is left alone
__device__ __forceinline__ void lds_barrier() {
#ifndef HIPEMU
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#else
    __syncthreads();
#endif
}
// fence for hipcc's instruction scheduler: what is written before it issues before it
__device__ __forceinline__ void sched_fence() {
#ifndef HIPEMU
    __builtin_amdgcn_sched_barrier(0);
#endif
}
// Instrumentation (cycle stamps, measurement variants: DGS_GEMM_DBG / DGS_ATTN_DBG / DGS_GEMM_EXP) exists only in a library built
// with -DDGS_INSTRUMENT (`DGS_INSTRUMENT=1 python -m dgs_amd.build` -> lib/libdgs_hip_instr.so, for tools/): in the product library
// every `if constexpr (kInstrumented)` branch is discarded, the run-time debug word is the literal 0.
#if defined(DGS_INSTRUMENT) && !defined(HIPEMU)
constexpr bool kInstrumented = true;
#else
constexpr bool kInstrumented = false;
#endif
// shader-clock / constant 100 MHz stamps of the debug modes (0 on the emulator)
__device__ __forceinline__ long long cycle_stamp() {
#ifndef HIPEMU
    return clock64();
#else
    return 0;
#endif
}
__device__ __forceinline__ long long wall_stamp() {
#ifndef HIPEMU
    return wall_clock64();
#else
    return 0;
#endif
}
// 32-bit LDS address of a pointer into the workgroup's LDS (what M0 takes); the emulator addresses LDS through the pointer itself
__device__ __forceinline__ uint32_t lds_address(const char* p) {
#ifndef HIPEMU
    return (uint32_t)(size_t)(__attribute__((address_space(3))) const char*)p;
#else
    (void)p;
    return 0u;
#endif
}
// One LDS-DMA wave-instruction (1 KiB) with all-scalar addressing: lane's 16 bytes at sbase + voff -> LDS lds_wave + LIT + 16 lane.
// Global address = SGPR pair + one loop-invariant 32-bit VGPR offset, LDS address = M0 = SGPR + literal: tools/ubench/dma_piece_bench
// prices this form at 1 cycle of the MFMA stream per piece, against 29 for what hipcc makes of the builtin with per-slab pointer
// arithmetic (64-bit VGPR address by v_lshl_add_u64, M0 restored from a spilled SGPR).  `lds_wave_ptr` = the same LDS place as a
// pointer (used by the emulator only).
template <int LIT>
__device__ __forceinline__ void lds_dma_scalar(const char* sbase, uint32_t voff, uint32_t lds_wave, char* lds_wave_ptr) {
#ifndef HIPEMU
    (void)lds_wave_ptr;
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_wave), "n"(LIT) : "memory", "m0", "scc");
#else
    (void)lds_wave;
    glds16(sbase + voff, lds_wave_ptr + LIT);
#endif
}
// agent-scope (sc1) relaxed store / 8-byte load: performed at the memory side, coherent across XCDs without a fence
__device__ __forceinline__ void st_agent(float* ptr, float v) {
#ifdef HIPEMU
    *ptr = v;
#else
    __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// 8- and 16-byte forms of the store (one instruction per 2 / 4 floats; 16-byte aligned for the latter).  The compiler does not
// count inline-asm memory operations in its own s_waitcnt bookkeeping: an uncounted OLDER or YOUNGER entry in the in-order queue
// only makes its waits stricter, and the code that needs these stores performed waits with an explicit wait_vmcnt.
__device__ __forceinline__ void st_agent2(float* ptr, float a, float b) {
#ifdef HIPEMU
    ptr[0] = a; ptr[1] = b;
#else
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {a, b};
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ void st_agent4(float* ptr, float a, float b, float c, float d) {
#ifdef HIPEMU
    ptr[0] = a; ptr[1] = b; ptr[2] = c; ptr[3] = d;
#else
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    const f32x4_t v = {a, b, c, d};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
#endif
}
__device__ __forceinline__ float2 ld_agent2(const float2* ptr) {
#ifdef HIPEMU
    return *ptr;
#else
    const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
#endif
}

// compute units of the current device (host side; the emulator's "chip" has DGS_EMU_CUS of them)
inline int compute_unit_count() {
#ifdef HIPEMU
    return getenv("DGS_EMU_CUS") ? atoi(getenv("DGS_EMU_CUS")) : 0;
#else
    int n = 0, d = 0;
    return hipGetDevice(&d) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) == hipSuccess ? n : 0;
#endif
}
// the same, asked once per process on the device (the emulator re-reads its environment variable: tests change it)
inline int compute_unit_count_cached() {
#ifdef HIPEMU
    return compute_unit_count();
#else
    static const int n = compute_unit_count();
    return n;
#endif
}

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
