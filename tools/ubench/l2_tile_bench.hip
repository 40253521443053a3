// l2_tile_bench.hip -- L2 -> LDS DMA bandwidth for the access pattern of a GEMM operand tile, as a function of the row stride.
// A workgroup (4 waves) stages [128 rows][128 B] slabs: lane l of a wave-instruction takes 16 B chunk (l & 7) of row (l >> 3),
// i.e. 8 rows x one 128-byte line; it walks the K extent of its 128 rows slab by slab (row_bytes / 128 slabs), then moves on to
// the next 128 rows.  All rows of a slab share the same offset within their row: with a power-of-two row stride they also share
// the low address bits that select the L2 channel.  Every XCD (workgroup id % 8) works on its own `rows` x stride region.
// Build: hipcc --offload-arch=gfx950 -O3 -o l2_tile_bench l2_tile_bench.hip ; run: ./l2_tile_bench row_bytes stride_bytes [rows] [wgs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void pull_tiles(const char* __restrict__ src, unsigned* sink, int row_bytes, int stride, int rows, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int xcd = blockIdx.x & 7, wgx = blockIdx.x >> 3;
    const char* base = src + (size_t)xcd * rows * stride;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nslab = row_bytes / 128, ntile = rows / 128;
    for (int t = 0; t < tiles_per_wg; ++t) {
        const int tile = (wgx * 5 + t) % ntile;
        const char* g = base + (size_t)(tile * 128 + wave * 32 + (lane >> 3)) * stride + (lane & 7) * 16;
        for (int s = 0; s < nslab; ++s) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (size_t)q * 8 * stride + s * 128),
                                                 (__attribute__((address_space(3))) void*)(lds + ((s & 3) * 16 + wave * 4 + q) * 1024), 16, 0, 0);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (*reinterpret_cast<unsigned*>(lds + tid * 4) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
    const int row_bytes = argc > 1 ? atoi(argv[1]) : 2048, stride = argc > 2 ? atoi(argv[2]) : 2048;
    const int rows = argc > 3 ? atoi(argv[3]) : 1024, wgs = argc > 4 ? atoi(argv[4]) : 512;
    const int tiles_per_wg = 64 * 16 / (row_bytes / 128);            // the same bytes per workgroup for every row length
    char* src; unsigned* sink;
    CHECK(hipMalloc(&src, (size_t)8 * rows * stride + 4096));
    CHECK(hipMemset(src, 1, (size_t)8 * rows * stride));
    CHECK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(pull_tiles, dim3(wgs), dim3(256), 65536, 0, src, sink, row_bytes, stride, rows, tiles_per_wg);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double bytes = (double)wgs * tiles_per_wg * 128.0 * row_bytes;
    printf("rows of %5d B, stride %5d B, %4d rows/XCD (%.2f MiB), %d WGs: %.3f ms  %6.2f TB/s  (%.1f B/clk/CU at 2.0 GHz)\n", row_bytes, stride, rows,
           rows * (double)stride / 1048576.0, wgs, best, bytes / best / 1e9, bytes / (best * 1e-3) / 256 / 2.0e9);
    return 0;
}
