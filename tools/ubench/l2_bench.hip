// l2_bench.hip -- how fast can a CU pull L2-resident data, and does the route matter?   (development microbenchmark)
//   mode 0: global -> LDS DMA, 16 B / lane (what the GEMM / attention rings use)
//   mode 1: global_load_dwordx4 into VGPRs (consumed by an xor)
//   mode 2: global_load_dwordx4 into VGPRs, then ds_write_b128 into LDS
// Every XCD (workgroup id % 8) streams its own `ws` bytes (<= its 4 MiB L2) over and over; workgroups of an XCD start at
// different offsets.  Build: hipcc --offload-arch=gfx950 -O3 -o l2_bench l2_bench.hip ; run: ./l2_bench [ws_kib] [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void pull(const uint4* __restrict__ src, unsigned* sink, int ws_vec, int iters, int inflight) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int xcd = blockIdx.x & 7, wgx = blockIdx.x >> 3;
    const uint4* base = src + (size_t)xcd * ws_vec;
    const int nthr = blockDim.x, tid = threadIdx.x, wave = tid >> 6;
    unsigned acc = 0;
    int off = (wgx * 8192 + tid) % ws_vec;                       // 128 KiB apart
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint4* g = base + off;
            if (MODE == 0) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                                 (__attribute__((address_space(3))) void*)(lds + (wave * 8 + u) * 1024), 16, 0, 0);
            } else {
                const uint4 v = *g;
                if (MODE == 2) *reinterpret_cast<uint4*>(lds + ((wave * 8 + u) * 64 + (tid & 63)) * 16) = v;
                else acc ^= v.x ^ v.y ^ v.z ^ v.w;
            }
            off += nthr;
            if (off >= ws_vec) off -= ws_vec;
        }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (MODE != 1) acc ^= *reinterpret_cast<unsigned*>(lds + tid * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int ws_kib = argc > 1 ? atoi(argv[1]) : 2048, threads = argc > 2 ? atoi(argv[2]) : 256, wgs = argc > 3 ? atoi(argv[3]) : 512;
    const int ws_vec = ws_kib * 64, iters = 400;
    uint4* src; unsigned* sink;
    CHECK(hipMalloc(&src, (size_t)8 * ws_vec * 16));
    CHECK(hipMemset(src, 1, (size_t)8 * ws_vec * 16));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(pull<0>, dim3(wgs), dim3(threads), 65536, 0, src, sink, ws_vec, iters, 8);
            if (mode == 1) hipLaunchKernelGGL(pull<1>, dim3(wgs), dim3(threads), 65536, 0, src, sink, ws_vec, iters, 8);
            if (mode == 2) hipLaunchKernelGGL(pull<2>, dim3(wgs), dim3(threads), 65536, 0, src, sink, ws_vec, iters, 8);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        const double bytes = (double)wgs * threads * 8 * 16 * iters;
        printf("ws %d KiB/XCD, %d WGs x %d threads, mode %d (%s): %.3f ms  %.2f TB/s  (%.1f B/clk/CU at 2.0 GHz, 256 CUs)\n", ws_kib, wgs, threads, mode,
               mode == 0 ? "LDS DMA b128" : mode == 1 ? "dwordx4 -> VGPR" : "dwordx4 -> VGPR -> ds_write_b128", best, bytes / best / 1e9,
               bytes / (best * 1e-3) / 256 / 2.0e9);
    }
    return 0;
}
