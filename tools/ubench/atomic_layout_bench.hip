// Microbenchmark: what does a (tile, Gaussian) instance's gradient flush cost as
//   A  nine atomic instructions into four arrays, one lane per instance       (blend_backward_kernel's flush, round 4)
//   B  ONE atomic instruction per 16-lane group into a 64-byte record, lane z adds value z
//   C  nine atomic instructions into the 64-byte record, one lane per instance (layout alone)
// Instances pick Gaussians at random (a tile's list is depth ordered: unrelated indices).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void flush_soa(const uint32_t* ids, size_t n, float* mean2d, float* conic, float* col, float* op) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = ids[i];
    const float v = 1.0f + (float)(i & 7);
    atomicAdd(col + 3 * (size_t)g, v); atomicAdd(col + 3 * (size_t)g + 1, v); atomicAdd(col + 3 * (size_t)g + 2, v);
    atomicAdd(mean2d + 3 * (size_t)g, v); atomicAdd(mean2d + 3 * (size_t)g + 1, v);
    atomicAdd(conic + 4 * (size_t)g, v); atomicAdd(conic + 4 * (size_t)g + 1, v); atomicAdd(conic + 4 * (size_t)g + 3, v);
    atomicAdd(op + g, v);
}

__global__ void flush_record_lanes(const uint32_t* ids, size_t n, float* rec) {
    // 256 threads flush 256 instances in 16 passes: lane = (instance in pass, value slot)
    const int tid = threadIdx.x, z = tid & 15;
    const size_t base = (size_t)blockIdx.x * 256;
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
        const size_t i = base + (size_t)pass * 16 + (tid >> 4);
        if (i < n && z < 9) {
            const uint32_t g = ids[i];
            atomicAdd(rec + 16 * (size_t)g + z, 1.0f + (float)(i & 7));
        }
    }
}

__global__ void flush_record_one_lane(const uint32_t* ids, size_t n, float* rec) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t g = ids[i];
    const float v = 1.0f + (float)(i & 7);
#pragma unroll
    for (int z = 0; z < 9; ++z) atomicAdd(rec + 16 * (size_t)g + z, v);
}

int main(int argc, char** argv) {
    const size_t G = argc > 1 ? strtoull(argv[1], 0, 10) : (size_t)4 * 262146;      // (view, Gaussian) pairs
    const size_t n = argc > 2 ? strtoull(argv[2], 0, 10) : (size_t)4 * 677775;      // instances flushed (trained-like regime, 4 views at 256^2)
    std::vector<uint32_t> h(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (size_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % G); }
    uint32_t* ids; float *m, *c, *col, *op, *rec;
    hipMalloc(&ids, n * 4); hipMemcpy(ids, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMalloc(&m, G * 12); hipMalloc(&c, G * 16); hipMalloc(&col, G * 12); hipMalloc(&op, G * 4); hipMalloc(&rec, G * 64);
    hipMemset(m, 0, G * 12); hipMemset(c, 0, G * 16); hipMemset(col, 0, G * 12); hipMemset(op, 0, G * 4); hipMemset(rec, 0, G * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned blocks = (unsigned)((n + 255) / 256);
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %8.1f us per flush of %zu instances (%.1f G value-adds/s)\n", name, ms * 50.0f, n, 9.0 * n / (ms / 20 * 1e-3) * 1e-9);
    };
    time("A nine instructions, four arrays, lane = instance", [&] { hipLaunchKernelGGL(flush_soa, dim3(blocks), dim3(256), 0, 0, ids, n, m, c, col, op); });
    time("B one instruction, 64-byte record, lane = (instance, value)", [&] { hipLaunchKernelGGL(flush_record_lanes, dim3(blocks), dim3(256), 0, 0, ids, n, rec); });
    time("C nine instructions, 64-byte record, lane = instance", [&] { hipLaunchKernelGGL(flush_record_one_lane, dim3(blocks), dim3(256), 0, 0, ids, n, rec); });
    return 0;
}
