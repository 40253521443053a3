// fetch_calib_bench.hip -- what rocprofv3's FETCH_SIZE reports for access patterns of KNOWN byte counts (measurement tool).
//
// MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B and reports exactly 1/2 of a wide coalesced streaming read
// (128-byte requests tallied at 64 B); "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern".  The rasterizer's blend kernels do not stream: they GATHER 8 + 16 + 16-byte records of three SoA arrays at list
// indices.  One kernel per pattern, every array far larger than L2 + Infinity Cache (no line is touched twice):
//   stream    16 B per lane, contiguous                      (the calibrated case: expect reported = 1/2 of the bytes)
//   line128   16 B per lane, one lane per 128-byte line      (sparse: a line is requested for 16 useful bytes)
//   line64    16 B per lane, one lane per 64-byte half line
//   records   the blend's staging: index list (random permutation) -> float2 + float4 + float4 of three arrays, 40 B used per lane
// Prints per pattern: lanes, useful bytes, milliseconds, lanes/s -- tools/pmc_traffic.py runs it under `--pmc FETCH_SIZE` and sets
// the reported KiB against these numbers (requests per lane = reported bytes / 64; a request that really moved 128 B would put
// `line128` above the HBM peak at the measured lanes/s -- that is how the size of a sparse request is decided).
//   hipcc --offload-arch=gfx950 -O2 -o fetch_calib_bench fetch_calib_bench.hip && ./fetch_calib_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

__global__ __launch_bounds__(256) void stream_kernel(const uint4* src, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= src[i].x;
    if (acc == 0x12345678u) *sink = acc;
}

template <int STRIDE>   // bytes between the 16-byte pieces of consecutive lanes
__global__ __launch_bounds__(256) void line_kernel(const char* src, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        acc ^= reinterpret_cast<const uint4*>(src + i * STRIDE)->x;
    if (acc == 0x12345678u) *sink = acc;
}

__global__ __launch_bounds__(256) void records_kernel(const unsigned* list, const float2* xy, const float4* co, const float4* rc, size_t n,
                                                      unsigned* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned id = list[i];
        acc += xy[id].x + co[id].w + rc[id].y;
    }
    if (acc == 1.2345e30f) *sink = 1u;
}

static float timed(void (*launch)(void*), void* ctx) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(ctx);                                   // warm-up (also the launch the PMC pass averages with)
    hipEventRecord(e0);
    launch(ctx);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

struct Ctx { const void *a, *b, *c, *d; size_t n; unsigned* sink; };

int main() {
    const size_t GiB = (size_t)1 << 30;
    char* buf = nullptr;
    if (hipMalloc(&buf, 3 * GiB) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, 3 * GiB);
    unsigned* sink = nullptr;
    hipMalloc(&sink, 4);
    // records: 48 M records (xy 384 MB, co 768 MB, rc 768 MB), every one read exactly once through a random permutation
    const size_t NR = (size_t)48 << 20;
    unsigned* list = nullptr;
    hipMalloc(&list, NR * 4);
    {
        std::vector<unsigned> h(NR);
        for (size_t i = 0; i < NR; ++i) h[i] = (unsigned)i;
        unsigned long long s = 88172645463325252ull;
        for (size_t i = NR - 1; i > 0; --i) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            const size_t j = (size_t)(s % (i + 1));
            const unsigned t = h[i]; h[i] = h[j]; h[j] = t;
        }
        hipMemcpy(list, h.data(), NR * 4, hipMemcpyHostToDevice);
    }
    Ctx c{};
    c.sink = sink;
    printf("pattern lanes useful_bytes ms lanes_per_us\n");
    c.a = buf; c.n = GiB / 16;
    float ms = timed([](void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, (const uint4*)c->a, c->n, c->sink); }, &c);
    printf("stream %zu %zu %.4f %.1f\n", c.n, c.n * 16, ms, c.n / (ms * 1e3));
    c.n = 2 * GiB / 128;
    ms = timed([](void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(line_kernel<128>, dim3(4096), dim3(256), 0, 0, (const char*)c->a, c->n, c->sink); }, &c);
    printf("line128 %zu %zu %.4f %.1f\n", c.n, c.n * 16, ms, c.n / (ms * 1e3));
    c.n = 2 * GiB / 64;
    ms = timed([](void* p) { Ctx* c = (Ctx*)p; hipLaunchKernelGGL(line_kernel<64>, dim3(4096), dim3(256), 0, 0, (const char*)c->a, c->n, c->sink); }, &c);
    printf("line64 %zu %zu %.4f %.1f\n", c.n, c.n * 16, ms, c.n / (ms * 1e3));
    c.a = list; c.b = buf; c.c = buf + NR * 8; c.d = buf + NR * 24; c.n = NR;
    ms = timed([](void* p) {
        Ctx* c = (Ctx*)p;
        hipLaunchKernelGGL(records_kernel, dim3(4096), dim3(256), 0, 0, (const unsigned*)c->a, (const float2*)c->b, (const float4*)c->c, (const float4*)c->d, c->n, c->sink);
    }, &c);
    printf("records %zu %zu %.4f %.1f\n", c.n, c.n * 44, ms, c.n / (ms * 1e3));
    hipDeviceSynchronize();
    return 0;
}
