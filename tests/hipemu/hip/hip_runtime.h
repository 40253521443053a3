// =============================================================================
// hipemu -- a tiny single-threaded CPU emulation of the HIP execution model.
//
// TEST INFRASTRUCTURE ONLY.  It exists because the build container has no GPU and
// GPU minutes are scarce: compiling csrc/*.hip against this header (instead of
// <hip/hip_runtime.h>) with a host clang gives a CPU library exporting the same C
// ABI, so kernel LOGIC (indexing, barriers, wave collectives, MFMA fragment maps,
// atomics) is checked against the oracle in the `-m "not gpu"` suite.  It says
// nothing about performance and is never loaded by the product path.
//
// Model: blocks run one after another; the threads of a block are fibers switched
// cooperatively (custom x86-64 context switch).  __syncthreads and wave64
// collectives (shfl / ballot / any / all / readfirstlane / mfma / ds_bpermute)
// are rendezvous points; lanes that have exited count as inactive.  A collective
// that can never complete (divergent lanes) aborts with a diagnostic.
// =============================================================================
#pragma once
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __constant__ static

// ----------------------------------------------------------------------------- basic types
struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct ushort4 { unsigned short x, y, z, w; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float3 make_float3(float x, float y, float z) { return {x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return {x, y}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return {x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return {x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return {x, y, z, w}; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace hipemu {

constexpr int WAVE = 64;
constexpr int WAVE_PAY = 160;  // bytes of payload per lane in a wave collective

struct Sync {
    int total = 0, finished = 0, arrived = 0;
    uint64_t gen = 0;
    int pay = 0;
    std::vector<unsigned char> in, snap[2];
    std::vector<unsigned char> present, snap_present[2];
    void init(int n, int pay_bytes) {
        total = n; finished = 0; arrived = 0; pay = pay_bytes;
        in.assign((size_t)n * pay, 0); snap[0] = in; snap[1] = in;
        present.assign(n, 0); snap_present[0] = present; snap_present[1] = present;
    }
    void complete() {
        const int b = (int)(gen & 1);
        snap[b] = in; snap_present[b] = present;
        std::fill(present.begin(), present.end(), 0);
        arrived = 0; ++gen;
    }
};

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    bool finished = false;
    uint3 tid{0, 0, 0};
    int flat = 0, lane = 0, wave = 0;
    Sync* wait = nullptr;
    uint64_t wait_gen = 0;
};

struct Ctx {
    std::vector<Fiber> fibers;
    std::vector<Sync> waves;
    Sync block;
    Fiber* cur = nullptr;
    void* sched_sp = nullptr;
    uint3 bid{0, 0, 0};
    dim3 bdim, gdim;
    std::function<void()> body;
    std::vector<char> dyn_lds;
    uint64_t progress = 0;
    int last_error = 0;
};
Ctx& ctx();
void run_grid(dim3 grid, dim3 block, size_t shmem, std::function<void()> body);
void yield_wait(Sync* s, uint64_t g);
inline char* dyn_lds() { return ctx().dyn_lds.data(); }

// Arrive at a rendezvous with `n` payload bytes; returns the snapshot buffer index to read from.
inline int arrive(Sync& s, int idx, const void* payload, int n) {
    Ctx& c = ctx();
    assert(n <= s.pay);
    if (n) std::memcpy(&s.in[(size_t)idx * s.pay], payload, n);
    s.present[idx] = 1;
    ++s.arrived; ++c.progress;
    const uint64_t g = s.gen;
    if (s.arrived + s.finished == s.total) s.complete();
    else yield_wait(&s, g);
    return (int)(g & 1);
}
inline Sync& my_wave() { Ctx& c = ctx(); return c.waves[c.cur->wave]; }
template <class T> inline T snap_get(const Sync& s, int b, int lane, int off = 0) {
    T v; std::memcpy(&v, &s.snap[b][(size_t)lane * s.pay + off], sizeof(T)); return v;
}

template <class T> inline T shfl(T v, int src) {
    Sync& s = my_wave(); const int me = ctx().cur->lane;
    const int b = arrive(s, me, &v, sizeof(T));
    src &= (WAVE - 1);
    if (src < s.total && s.snap_present[b][src]) return snap_get<T>(s, b, src);
    return v;
}
inline unsigned long long ballot(int pred) {
    Sync& s = my_wave(); const int me = ctx().cur->lane;
    const int b = arrive(s, me, &pred, sizeof(int));
    unsigned long long m = 0;
    for (int l = 0; l < s.total; ++l)
        if (s.snap_present[b][l] && snap_get<int>(s, b, l)) m |= 1ull << l;
    return m;
}
template <class T> inline T readfirstlane(T v) {
    Sync& s = my_wave(); const int me = ctx().cur->lane;
    const int b = arrive(s, me, &v, sizeof(T));
    for (int l = 0; l < s.total; ++l)
        if (s.snap_present[b][l]) return snap_get<T>(s, b, l);
    return v;
}
inline int block_reduce(int v, int op) {  // 0 barrier, 1 count, 2 and, 3 or
    Ctx& c = ctx(); Sync& s = c.block;
    const int b = arrive(s, c.cur->flat, &v, sizeof(int));
    if (op == 0) return 0;
    int cnt = 0, all = 1, any = 0;
    for (int l = 0; l < s.total; ++l)
        if (s.snap_present[b][l]) { const int x = snap_get<int>(s, b, l) != 0; cnt += x; all &= x; any |= x; }
    return op == 1 ? cnt : (op == 2 ? all : any);
}

// ---- MFMA emulation (fragment maps: cdna_hip_programming.md section 3) ----
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
inline float bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
inline float f16_to_f32(unsigned short h) { _Float16 x; std::memcpy(&x, &h, 2); return (float)x; }

template <bool F16>
inline f32x4_t mfma_16x16x32(bf16x8_t a, bf16x8_t b, f32x4_t c) {
    struct P { bf16x8_t a, b; f32x4_t c; } p{a, b, c};
    Sync& s = my_wave(); const int me = ctx().cur->lane;
    const int buf = arrive(s, me, &p, sizeof(P));
    // lane l: A[i=l&15][k=8*(l>>4)+j], B[k=8*(l>>4)+j][n=l&15]; D: col=l&15, row=4*(l>>4)+r
    f32x4_t d = c;
    const int col = me & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (me >> 4) + r;
        float acc = c[r];
        for (int kb = 0; kb < 4; ++kb) {
            const P pa = snap_get<P>(s, buf, row + 16 * kb);
            const P pb = snap_get<P>(s, buf, col + 16 * kb);
            for (int j = 0; j < 8; ++j) {
                const float x = F16 ? f16_to_f32((unsigned short)pa.a[j]) : bf16_to_f32((unsigned short)pa.a[j]);
                const float y = F16 ? f16_to_f32((unsigned short)pb.b[j]) : bf16_to_f32((unsigned short)pb.b[j]);
                acc += x * y;
            }
        }
        d[r] = acc;
    }
    return d;
}
template <bool F16>
inline f32x16_t mfma_32x32x16(bf16x8_t a, bf16x8_t b, f32x16_t c) {
    struct P { bf16x8_t a, b; } p{a, b};
    Sync& s = my_wave(); const int me = ctx().cur->lane;
    const int buf = arrive(s, me, &p, sizeof(P));
    // lane l: A[i=l&31][k=8*(l>>5)+j], B[k=8*(l>>5)+j][n=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
    f32x16_t d = c;
    const int col = me & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (me >> 5);
        float acc = c[r];
        for (int kb = 0; kb < 2; ++kb) {
            const P pa = snap_get<P>(s, buf, row + 32 * kb);
            const P pb = snap_get<P>(s, buf, col + 32 * kb);
            for (int j = 0; j < 8; ++j) {
                const float x = F16 ? f16_to_f32((unsigned short)pa.a[j]) : bf16_to_f32((unsigned short)pa.a[j]);
                const float y = F16 ? f16_to_f32((unsigned short)pb.b[j]) : bf16_to_f32((unsigned short)pb.b[j]);
                acc += x * y;
            }
        }
        d[r] = acc;
    }
    return d;
}
inline int ds_bpermute(int addr, int v) {
    Sync& s = my_wave(); const int me = ctx().cur->lane;
    const int b = arrive(s, me, &v, sizeof(int));
    const int src = (addr >> 2) & (WAVE - 1);
    return (src < s.total && s.snap_present[b][src]) ? snap_get<int>(s, b, src) : 0;
}
// global_load_lds: LDS destination = first active lane's lds pointer + lane * size (wave-uniform base).
inline void global_load_lds(const void* g, void* lds, int size, int offset) {
    Sync& s = my_wave(); const int me = ctx().cur->lane;
    struct P { const void* g; void* l; } p{g, lds};
    const int b = arrive(s, me, &p, sizeof(P));
    void* base = nullptr;
    for (int l = 0; l < s.total; ++l) if (s.snap_present[b][l]) { base = snap_get<P>(s, b, l).l; break; }
    std::memcpy((char*)base + offset + (size_t)me * size, (const char*)g + offset, size);
}

}  // namespace hipemu

// ----------------------------------------------------------------------------- device builtins
#define threadIdx (hipemu::ctx().cur->tid)
#define blockIdx (hipemu::ctx().bid)
#define blockDim (hipemu::ctx().bdim)
#define gridDim (hipemu::ctx().gdim)
#define warpSize 64

static inline void __syncthreads() { hipemu::block_reduce(0, 0); }
static inline int __syncthreads_count(int p) { return hipemu::block_reduce(p, 1); }
static inline int __syncthreads_and(int p) { return hipemu::block_reduce(p, 2); }
static inline int __syncthreads_or(int p) { return hipemu::block_reduce(p, 3); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}
static inline int __lane_id() { return hipemu::ctx().cur->lane; }

template <class T> static inline T __shfl(T v, int src, int = 64) { return hipemu::shfl(v, src); }
template <class T> static inline T __shfl_xor(T v, int m, int = 64) { return hipemu::shfl(v, __lane_id() ^ m); }
template <class T> static inline T __shfl_down(T v, unsigned d, int = 64) {
    const int src = __lane_id() + (int)d; return hipemu::shfl(v, src < 64 ? src : __lane_id()); }
template <class T> static inline T __shfl_up(T v, unsigned d, int = 64) {
    const int src = __lane_id() - (int)d; return hipemu::shfl(v, src >= 0 ? src : __lane_id()); }
static inline unsigned long long __ballot(int p) { return hipemu::ballot(p); }
static inline int __any(int p) { return hipemu::ballot(p) != 0; }
static inline int __all(int p) { return hipemu::ballot(!p) == 0; }
#define __builtin_amdgcn_readfirstlane(v) hipemu::readfirstlane(v)
#define __builtin_amdgcn_ds_bpermute(a, v) hipemu::ds_bpermute(a, v)
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z) hipemu::mfma_16x16x32<false>(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu::mfma_32x32x16<false>(a, b, c)
#define __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z) hipemu::mfma_16x16x32<true>(a, b, c)
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) hipemu::mfma_32x32x16<true>(a, b, c)
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) hipemu::global_load_lds(g, l, size, off)
#define __builtin_amdgcn_s_barrier() __syncthreads()
#define __builtin_amdgcn_wave_barrier() ((void)hipemu::ballot(1))   /* lanes of a wave run as separate fibers here */
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))

template <class T> static inline T min(T a, T b) { return b < a ? b : a; }
template <class T> static inline T max(T a, T b) { return a < b ? b : a; }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
static inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float __fmaf_rn(float a, float b, float c) { return fmaf(a, b, c); }
static inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
static inline float __saturatef(float a) { return a < 0 ? 0 : (a > 1 ? 1 : a); }

#define HIPEMU_ATOMIC(T)                                                                    \
    static inline T atomicAdd(T* p, T v) { T o = *p; *p = (T)(o + v); return o; }           \
    static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
HIPEMU_ATOMIC(int) HIPEMU_ATOMIC(unsigned) HIPEMU_ATOMIC(unsigned long long) HIPEMU_ATOMIC(float)
#define HIPEMU_ATOMIC_INT(T)                                                                \
    static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }                 \
    static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }                \
    static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }         \
    static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }         \
    static inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }
HIPEMU_ATOMIC_INT(int) HIPEMU_ATOMIC_INT(unsigned) HIPEMU_ATOMIC_INT(unsigned long long)

// ----------------------------------------------------------------------------- host runtime
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::run_grid((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })
static inline hipError_t hipGetLastError() { int e = hipemu::ctx().last_error; hipemu::ctx().last_error = 0; return e; }
static inline hipError_t hipPeekAtLastError() { return hipemu::ctx().last_error; }
static inline const char* hipGetErrorString(hipError_t e) { return e ? "hipemu error" : "hipSuccess"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t = nullptr) { std::memset(p, v, n); return 0; }
static inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy(d, s, n); return 0; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return 0; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
static inline hipError_t hipDeviceSynchronize() { return 0; }
template <class T> static inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)std::calloc(1, n ? n : 1); return 0; }
template <class T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { *p = (T*)std::calloc(1, n ? n : 1); return 0; }
static inline hipError_t hipFree(void* p) { std::free(p); return 0; }
static inline hipError_t hipHostFree(void* p) { std::free(p); return 0; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return 0; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return 0; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return 0; }
#define hipEventDisableTiming 0x2
#define hipStreamNonBlocking 0x1
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return 0; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0; return 0; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
