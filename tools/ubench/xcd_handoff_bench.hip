// xcd_handoff_bench.hip -- what does a device-side hand-off between two workgroups of ONE launch cost on MI355X?
//
// The structural next step of the DiT step (one persistent kernel for proj -> LN2 -> fc1 -> fc2 -> LN1' -> QKV'
// behind per-row-block counters) and the attention kernel's tail merge both hand data from one workgroup to another without a kernel
// boundary.  L2 is per XCD and not coherent across XCDs, so there are two ways to make the payload visible:
//     mode 0  "fence":  plain stores, agent-scope RELEASE fence (writes back the XCD's dirty L2 lines), flag store;
//                       consumer: spin on the flag, agent-scope ACQUIRE fence (invalidates), plain loads
//     mode 1  "sc1":    payload stored with sc1 (performed at the memory side), s_waitcnt vmcnt(0), flag store;
//                       consumer: spin on the flag, payload loaded with sc1 (agent-scope relaxed atomic loads: bypass the L2s)
//     mode 2  "sc1dma": producer as mode 1; the consumer pulls the payload into LDS with sc1 LDS-DMA (global_load_lds_dwordx4 sc1:
//                       1 KiB per wave-instruction, no registers) in 64 KiB chunks and checks it there -- how a GEMM stage of a
//                       persistent pipeline would read the previous stage's tiles
// 256 workgroups of 256 threads, one per CU (LDS pad); workgroup p < 128 produces for consumer p + 128 (same XCD: the dispatcher
// places block b on XCD b % 8) or p + 129 (the next XCD); XCC_ID is read back to check that.  Every workgroup stamps the constant
// 100 MHz clock: producer start / flag stored, consumer flag seen / payload read.  `rounds` hand-offs per pair through the SAME
// buffer with new contents each round (a stale line in the consumer's L2 shows up as an error).  `dirty_mb`: other workgroups'
// plain stores in flight per round (what a release fence has to write back besides the payload).
//   hipcc --offload-arch=gfx950 -O2 -o xcd_handoff_bench xcd_handoff_bench.hip && ./xcd_handoff_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

struct Stamp { unsigned long long t0, t1, t2, t3; unsigned xcc, errors; };

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_sc1_16(void* p, uint4 v) {
    const u32x4_t r = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(r) : "memory");
}
__device__ __forceinline__ unsigned long long ld_sc1_8(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bounded spin (0.2 s of the 100 MHz clock): a partner that never became resident must not hang the GPU -- counted as 1e6 errors
__device__ __forceinline__ void spin_until(const unsigned* flag, unsigned want, unsigned* errors) {
    const unsigned long long t = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t > 20000000ull) { *errors += 1000000u; break; }
    }
}

__global__ __launch_bounds__(256) void handoff_kernel(char* payload, unsigned* flags, Stamp* stamps, char* scratch, int bytes, int mode, int partner_shift,
                                                      int rounds, int dirty_bytes_per_wg) {
    extern __shared__ float pad[];
    const int tid = threadIdx.x, b = blockIdx.x, half = gridDim.x / 2;
    if (tid == 0) pad[0] = 0.f;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const bool producer = b < half;
    const int pair = producer ? b : (b - half - partner_shift + half) % half;          // consumer c = half + (p + shift) % half serves producer p
    char* buf = payload + (size_t)pair * bytes;
    unsigned* flag = flags + pair * 32;                                                // one 128-byte line per pair
    unsigned long long t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    unsigned errors = 0;
    __shared__ unsigned bail;
    if (tid == 0) bail = 0;
    __syncthreads();
    // background: every workgroup dirties its own scratch with plain stores (what a release fence also has to write back)
    char* mine = scratch + (size_t)b * dirty_bytes_per_wg;
    for (int r = 1; r <= rounds; ++r) {
        for (int i = tid * 16; i < dirty_bytes_per_wg; i += 256 * 16) *reinterpret_cast<uint4*>(mine + i) = make_uint4(r, i, b, 7);
        if (producer) {
            // wait until the consumer has finished the previous round (its ack is the same flag word + 1 line further)
            if (tid == 0) { spin_until(flag + 16, (unsigned)(r - 1), &errors); if (errors >= 1000000u) bail = 1; }
            __syncthreads();
            if (bail) break;
            const unsigned long long s0 = wall_clock64();
            for (int i = tid * 16; i < bytes; i += 256 * 16) {
                const uint4 v = make_uint4((unsigned)r, (unsigned)i, (unsigned)pair, (unsigned)(r * 2654435761u + i));
                if (mode == 0) *reinterpret_cast<uint4*>(buf + i) = v;
                else st_sc1_16(buf + i, v);
            }
            if (mode == 0) __atomic_thread_fence(__ATOMIC_RELEASE);                    // agent scope in HIP: L2 write-back
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(flag, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long s1 = wall_clock64();
            t0 += s0; t1 += s1;
        } else {
            if (tid == 0) { spin_until(flag, (unsigned)r, &errors); if (errors >= 1000000u) bail = 1; }
            __syncthreads();
            if (bail) break;
            const unsigned long long s2 = wall_clock64();
            if (mode == 0) __atomic_thread_fence(__ATOMIC_ACQUIRE);
            if (mode == 2) {
                char* lbuf = reinterpret_cast<char*>(pad) + 1024;                     // 64 KiB chunk behind the pad word
                const int wave = tid >> 6, lane = tid & 63;
                for (int c0 = 0; c0 < bytes; c0 += 65536) {
                    const int n = bytes - c0 < 65536 ? bytes - c0 : 65536;
                    for (int piece = wave; piece * 1024 < n; piece += 4)              // lane i's 16 bytes land at the piece's base + 16 i
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(buf + c0 + piece * 1024 + lane * 16),
                                                         (__attribute__((address_space(3))) void*)(lbuf + piece * 1024), 16, 0, 16 /* sc1 */);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    for (int i = tid * 16; i < n; i += 256 * 16) {
                        const uint4 v = *reinterpret_cast<const uint4*>(lbuf + i);
                        const int gi = c0 + i;
                        if (v.x != (unsigned)r || v.y != (unsigned)gi || v.w != (unsigned)(r * 2654435761u + gi)) ++errors;
                    }
                    __syncthreads();
                }
            } else
            for (int i = tid * 16; i < bytes; i += 256 * 16) {
                uint4 v;
                if (mode == 0) v = *reinterpret_cast<const uint4*>(buf + i);
                else {
                    const unsigned long long lo = ld_sc1_8(buf + i), hi = ld_sc1_8(buf + i + 8);
                    v = make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
                }
                if (v.x != (unsigned)r || v.y != (unsigned)i || v.w != (unsigned)(r * 2654435761u + i)) ++errors;
            }
            __syncthreads();
            const unsigned long long s3 = wall_clock64();
            if (tid == 0) __hip_atomic_store(flag + 16, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t2 += s2; t3 += s3;
        }
    }
    __shared__ unsigned etot;
    if (tid == 0) etot = 0;
    __syncthreads();
    if (errors) atomicAdd(&etot, errors);
    __syncthreads();
    if (tid == 0) stamps[b] = Stamp{t0, t1, t2, t3, xcc, etot + (pad[0] == 1.f ? 1u : 0u)};
}

int main(int argc, char** argv) {
    const int G = 256, rounds = 50;
    const int dirty_kb = argc > 1 ? atoi(argv[1]) : 0;                                 // per workgroup and round
    char *payload, *scratch;
    unsigned* flags;
    Stamp* stamps;
    const int max_bytes = 1 << 20;
    hipMalloc(&payload, (size_t)(G / 2) * max_bytes);
    hipMalloc(&scratch, (size_t)G * (dirty_kb ? dirty_kb * 1024 : 16));
    hipMalloc(&flags, (G / 2) * 128);
    hipMalloc(&stamps, G * sizeof(Stamp));
    hipFuncSetAttribute(reinterpret_cast<const void*>(handoff_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    printf("# hand-off between two workgroups of one launch, %d rounds per pair, 128 pairs at once; background dirty stores %d KiB per workgroup and round\n", rounds, dirty_kb);
    printf("# mode   placement  payload    producer write+release   flag -> seen   consumer acquire+read   errors   (us, mean over pairs and rounds)\n");
    for (int mode = 0; mode < 3; ++mode)
        for (int shift = 0; shift < 2; ++shift)
            for (int bytes : {4096, 65536, 1 << 20}) {
                hipMemset(flags, 0, (G / 2) * 128);
                hipLaunchKernelGGL(handoff_kernel, dim3(G), dim3(256), 100 * 1024, 0, payload, flags, stamps, scratch, bytes, mode, shift, rounds, dirty_kb * 1024);
                if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
                std::vector<Stamp> h(G);
                hipMemcpy(h.data(), stamps, G * sizeof(Stamp), hipMemcpyDeviceToHost);
                double wr = 0, hand = 0, rd = 0;
                unsigned long long err = 0;
                int same = 0;
                for (int p = 0; p < G / 2; ++p) {
                    const int c = G / 2 + (p + shift) % (G / 2);
                    wr += (double)(h[p].t1 - h[p].t0);
                    hand += (double)((long long)(h[c].t2 - h[p].t1));
                    rd += (double)(h[c].t3 - h[c].t2);
                    err += h[c].errors;
                    same += h[p].xcc == h[c].xcc;
                }
                const double k = 1.0 / (G / 2) / rounds / 100.0;                       // 100 MHz ticks -> us
                printf("%-6s %-10s %7d B   %10.2f               %8.2f       %10.2f            %llu   (%d of 128 pairs on one XCD)\n", mode == 0 ? "fence" : mode == 1 ? "sc1" : "sc1dma",
                       shift ? "next XCD" : "same XCD", bytes, wr * k, hand * k, rd * k, err, same);
            }
    return 0;
}
