// Can a small-footprint kernel on a second stream run BESIDE the one-round MFMA kernels of the DiT step without holding their
// workgroups back?  The sliced GEMMs take 2 waves x 240 registers per SIMD (32 left) and 128 of 160 KiB of LDS; this background
// kernel is built to fit what is left: <= 32 VGPRs, no LDS, 256-thread workgroups.  Each wave streams `chunks` KiB (16 bytes per
// lane per step, one step in flight: the register budget) and dots it with a second stream -- a GEMV's shape.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libcoreside.so coreside_bench.hip        (tools/coreside_run.py drives it)
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256, 8) void bg_gemv_kernel(const uint4* __restrict__ w, const uint4* __restrict__ a, float* __restrict__ out, int chunks,
                                                        size_t wave_stride) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint4* wp = w + wave * wave_stride + lane;
    const uint4* ap = a + lane;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const uint4 x = wp[(size_t)c * 64], y = ap[(size_t)(c & 7) * 64];
        s += __uint_as_float(x.x & 0xffff0000u) * __uint_as_float(y.x & 0xffff0000u) + __uint_as_float(x.y << 16) * __uint_as_float(y.y << 16) +
             __uint_as_float(x.z & 0xffff0000u) * __uint_as_float(y.z & 0xffff0000u) + __uint_as_float(x.w << 16) * __uint_as_float(y.w << 16);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[wave] = s;
}

// An RCCL-shaped neighbour: `wgs` workgroups of 512 threads copy `n16` 16-byte words in a grid-stride loop (a ring all-reduce step on one
// rank is a few dozen such workgroups streaming its buckets between HBM and the xGMI links for the whole backward).
__global__ __launch_bounds__(512) void bg_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) dst[i] = src[i];
}

extern "C" int coreside_copy(const void* src, void* dst, size_t bytes, int wgs, void* stream) {
    hipLaunchKernelGGL(bg_copy_kernel, dim3(wgs), dim3(512), 0, static_cast<hipStream_t>(stream), static_cast<const uint4*>(src), static_cast<uint4*>(dst), bytes / 16);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int coreside_bg(const void* w, const void* a, float* out, int wgs, int chunks, size_t wave_stride_uint4, void* stream) {
    hipLaunchKernelGGL(bg_gemv_kernel, dim3(wgs), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const uint4*>(w), static_cast<const uint4*>(a), out, chunks,
                       wave_stride_uint4);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
