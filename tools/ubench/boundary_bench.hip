// boundary_bench.hip -- what does the boundary between two DEPENDENT kernels on one stream cost at batch-1 kernel sizes?
// N launches of a kernel whose every workgroup spins for a fixed time; each workgroup stamps the constant 100 MHz clock
// (wall_clock64) when it starts and when it ends.  Per boundary i -> i+1:
//     gap    = first start of launch i+1 - last end of launch i      (no workgroup of either launch is running)
//     ramp   = last start - first start of launch i+1                (the dispatcher filling the chip)
//     spread = last end - first end of launch i                      (the tail, here only what the ramp leaves behind)
// and, from HIP events around the whole sequence, the time per launch beyond the spin itself; last, the same sequence captured
// into one hipGraph.  Grids / workgroup shapes are
// those of the DiT step's kernels: 256 x 512 threads (attention, one round), 1088 x 256 (LayerNorm), 512 x 256 (128-wide GEMM).
//   hipcc --offload-arch=gfx950 -O2 -o boundary_bench boundary_bench.hip && ./boundary_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__global__ void spin_kernel(unsigned long long* stamps, long long spin_ticks, int launch, int lds_words) {
    extern __shared__ float pad[];
    if (lds_words) pad[threadIdx.x % lds_words] = (float)threadIdx.x;
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(2);
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned long long* s = stamps + ((size_t)launch * gridDim.x + blockIdx.x) * 2;
        s[0] = t0;
        s[1] = t1 + (lds_words && pad[0] == 12345.f ? 1 : 0);
    }
}

static void run(int G, int threads, int lds_bytes, double spin_us, int N, bool graph = false) {
    unsigned long long* d;
    hipMalloc(&d, (size_t)N * G * 16);
    const long long ticks = (long long)(spin_us * 100.0);      // 100 MHz
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipStream_t st;
    hipStreamCreate(&st);
    hipGraphExec_t exec = nullptr;
    if (graph) {                                               // the same N launches captured once, replayed as one hipGraph
        hipGraph_t g;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(G), dim3(threads), lds_bytes, st, d, ticks, i, lds_bytes / 4);
        hipStreamEndCapture(st, &g);
        hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    }
    for (int rep = 0; rep < 2; ++rep) {                        // the second pass is the measured one
        hipEventRecord(e0, st);
        if (graph) hipGraphLaunch(exec, st);
        else for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin_kernel, dim3(G), dim3(threads), lds_bytes, st, d, ticks, i, lds_bytes / 4);
        hipEventRecord(e1, st);
        hipDeviceSynchronize();
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)N * G * 2);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    double gap = 0, ramp = 0, spread = 0, busy = 0;
    std::vector<double> gaps;
    for (int i = 0; i < N; ++i) {
        unsigned long long s_min = ~0ull, s_max = 0, e_min = ~0ull, e_max = 0;
        for (int b = 0; b < G; ++b) {
            const unsigned long long s = h[((size_t)i * G + b) * 2], e = h[((size_t)i * G + b) * 2 + 1];
            s_min = std::min(s_min, s); s_max = std::max(s_max, s); e_min = std::min(e_min, e); e_max = std::max(e_max, e);
        }
        ramp += (double)(s_max - s_min) * 0.01;
        spread += (double)(e_max - e_min) * 0.01;
        busy += (double)(e_max - s_min) * 0.01;
        if (i + 1 < N) {
            unsigned long long n_min = ~0ull;
            for (int b = 0; b < G; ++b) n_min = std::min(n_min, h[((size_t)(i + 1) * G + b) * 2]);
            const double g = ((double)n_min - (double)e_max) * 0.01;
            gap += g;
            gaps.push_back(g);
        }
    }
    std::sort(gaps.begin(), gaps.end());
    printf("%s grid %4d x %3d thr, LDS %6d B, spin %6.1f us: per launch %7.2f us (events) = spin + %5.2f | first start -> last end %7.2f, "
           "ramp %5.2f, end spread %5.2f, gap to next launch mean %5.2f median %5.2f us\n",
           graph ? "graph " : "stream", G, threads, lds_bytes, spin_us, ms * 1e3 / N, ms * 1e3 / N - spin_us, busy / N, ramp / N, spread / N, gap / (N - 1),
           gaps[gaps.size() / 2]);
    hipFree(d);
}

int main() {
    const int N = 200;
    for (double us : {5.0, 20.0, 90.0}) {
        run(256, 512, 65536, us, N);       // attention: one workgroup per CU
        run(512, 256, 49152, us, N);       // 128-wide GEMM: two per CU
        run(1088, 256, 0, us, N);          // LayerNorm: four rows per workgroup
    }
    for (double us : {5.0, 20.0}) {        // the same sequences as ONE hipGraph launch
        run(256, 512, 65536, us, N, true);
        run(1088, 256, 0, us, N, true);
    }
    return 0;
}
