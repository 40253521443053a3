// sumsq_bench: the gradient-norm partial pass (csrc/optim.hip sumsq_partials_kernel) at chunk sizes of 65,536 .. 4,096 elements
// per workgroup, on a bucket of the trainer's size (one DiT block's gradients: 18.9 M floats) and on the whole flat buffer.
// The same loop as the product's full-chunk branch (eight 16-byte loads in flight per lane, adds in index order).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/sumsq_bench.hip -o tools/ubench/sumsq_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int CHUNK, int UNR>
__global__ __launch_bounds__(256) void sumsq(const float* __restrict__ x, float* __restrict__ partials) {
    __shared__ float s_w[4];
    const int tid = threadIdx.x;
    const float4* src = reinterpret_cast<const float4*>(x + (long long)blockIdx.x * CHUNK) + tid;
    float acc = 0.0f;
#pragma unroll 1
    for (int j0 = 0; j0 < CHUNK / 1024; j0 += UNR) {
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = src[(j0 + u) * 256];
#pragma unroll
        for (int u = 0; u < UNR; ++u) { acc += v[u].x * v[u].x; acc += v[u].y * v[u].y; acc += v[u].z * v[u].z; acc += v[u].w * v[u].w; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) s_w[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) partials[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

template <int CHUNK, int UNR>
static void run(const float* x, float* part, long long n, const char* what) {
    const int grid = (int)(n / CHUNK);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((sumsq<CHUNK, UNR>), dim3(grid), dim3(256), 0, 0, x, part);
    const int reps = 20;
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((sumsq<CHUNK, UNR>), dim3(grid), dim3(256), 0, 0, x, part);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1000.0 / reps;
    printf("%-12s chunk %6d x %d loads: %6d workgroups  %8.2f us  %6.2f TB/s\n", what, CHUNK, UNR, grid, us, (double)grid * CHUNK * 4 / us * 1e-6);
}

int main() {
    const long long whole = 460LL << 20, bucket = 18874368;      // floats (bucket: 288 chunks of 65,536)
    float *x, *part;
    hipMalloc(&x, whole * 4); hipMalloc(&part, (whole / 4096) * 4);
    hipMemset(x, 0, whole * 4);
    for (int pass = 0; pass < 2; ++pass) {
        const long long n = pass ? whole : bucket;
        const char* what = pass ? "whole buffer" : "one bucket";
        run<65536, 8>(x, part, n, what);
        run<32768, 8>(x, part, n, what);
        run<16384, 8>(x, part, n, what);
        run<16384, 16>(x, part, n, what);
        run<8192, 8>(x, part, n, what);
        run<4096, 4>(x, part, n, what);
    }
    return 0;
}
