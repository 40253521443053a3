// place_bench.hip -- where does the dispatcher put workgroup b?  One launch of G workgroups x 256 threads (10 KB LDS each,
// like the rasterizer's blend kernels); every workgroup spins ~SPIN cycles so that the whole grid is co-resident, and records
// XCC_ID and HW_ID (cu / sh / se).  Prints block -> (xcc, se, sh, cu) and a summary: workgroups per CU, and whether
// block b and block b + 256 share a CU.
//   hipcc --offload-arch=gfx950 -O2 -o place_bench place_bench.hip && ./place_bench [G] [SPIN]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <vector>

__global__ __launch_bounds__(256) void where_kernel(unsigned* out, long long spin) {
    __shared__ float pad[2560];
    pad[threadIdx.x] = (float)threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc + (unsigned)(pad[5] == 77.f);
    }
}

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 1024;
    const long long spin = argc > 2 ? atoll(argv[2]) : 200000;
    unsigned* d;
    hipMalloc(&d, G * 8);
    std::vector<unsigned> h(2 * G);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(where_kernel, dim3(G), dim3(256), 0, 0, d, spin);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> per_cu;
    for (int b = 0; b < G; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15u;
        const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        per_cu[key].push_back(b);
        if (b < 48 || (b >= 256 && b < 272)) printf("block %4d -> xcc %u se %u sh %u cu %2u (hw %08x)\n", b, xcc, se, sh, cu, hw);
    }
    std::map<int, int> hist;
    int same = 0;
    for (auto& kv : per_cu) hist[(int)kv.second.size()]++;
    for (auto& kv : per_cu) {
        bool all = true;
        for (size_t i = 1; i < kv.second.size(); ++i) all = all && ((kv.second[i] - kv.second[0]) % 256 == 0);
        same += all;
    }
    printf("distinct CUs %zu; workgroups per CU histogram:", per_cu.size());
    for (auto& kv : hist) printf("  %d x%d", kv.first, kv.second);
    printf("\nCUs whose workgroups are all congruent mod 256: %d\n", same);
    int shown = 0;
    for (auto& kv : per_cu) {
        if (shown++ >= 12) break;
        printf("cu %05x:", kv.first);
        for (int b : kv.second) printf(" %d", b);
        printf("\n");
    }
    return 0;
}
