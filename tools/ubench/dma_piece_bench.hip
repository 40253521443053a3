// dma_piece_bench.hip -- issue cost of one LDS-DMA wave-instruction (global_load_lds_dwordx4, 1 KiB) beside MFMAs, by the
// shape of the 1 KiB in memory:  P8  = 8 rows x 128 B (a [rows][64] bf16 slab piece: 8 full cache lines)
//                                P16 = 16 rows x 64 B (a [rows][32] bf16 slab piece: 16 half lines)
//                                P1  = 1 KiB contiguous
// Row stride 8 KiB (K = 4096 bf16).  Every group = { v_mfma_f32_32x32x16_bf16, ds_read_b128 }; a DMA (+ 64-bit address add +
// M0 update, as a GEMM loop has them) follows every N-th MFMA.  One wave per SIMD (256 threads), cycles per group from
// workgroup 0 while G workgroups run (1 = alone, 256 = every CU streaming).
//   hipcc --offload-arch=gfx950 -O2 -o dma_piece_bench dma_piece_bench.hip && ./dma_piece_bench
#include <hip/hip_runtime.h>
#include <stdio.h>

#define MFMA(acc) "v_mfma_f32_32x32x16_bf16 " acc ", v[180:183], v[184:187], " acc "\n\t"
#define DSRD "ds_read_b128 v[188:191], v194\n\t"
#define DMA "global_load_lds_dwordx4 v[196:197], off\n\tv_add_co_u32 v196, vcc, v198, v196\n\tv_addc_co_u32 v197, vcc, 0, v197, vcc\n\ts_add_u32 m0, m0, 0x400\n\ts_and_b32 m0, m0, 0x3fff\n\t"
// addressing forms of the same DMA (shape P8):
//  F0 fixed address, fixed M0            F1 fixed address, M0 stepped (2 SALU)
//  F2 SGPR base + 32-bit VGPR offset, base stepped (2 SALU), M0 stepped (2 SALU): no VALU at all
//  F4 fresh 64-bit address = VGPR pair + SGPR pair (v_lshl_add_u64), SGPR stepped, M0 stepped (what hipcc emits for the GEMM)
//  F5 = F4 with M0 taken from a VGPR lane (v_readlane_b32 + s_mov_b32 m0: an SGPR spill, as in the sliced GEMM's loop)
#define DMA0 "global_load_lds_dwordx4 v[196:197], off\n\t"
#define M0STEP "s_add_u32 m0, m0, 0x400\n\ts_and_b32 m0, m0, 0x3fff\n\t"
#define DMA1 DMA0 M0STEP
#define DMA2 "global_load_lds_dwordx4 v199, s[20:21]\n\ts_add_u32 s20, s20, 128\n\ts_addc_u32 s21, s21, 0\n\t" M0STEP
#define DMA4 "v_lshl_add_u64 v[200:201], v[196:197], 0, s[22:23]\n\tglobal_load_lds_dwordx4 v[200:201], off\n\ts_add_u32 s22, s22, 128\n\ts_addc_u32 s23, s23, 0\n\t" M0STEP
#define DMA5 "v_readlane_b32 s24, v202, 3\n\ts_mov_b32 m0, s24\n\tv_lshl_add_u64 v[200:201], v[196:197], 0, s[22:23]\n\tglobal_load_lds_dwordx4 v[200:201], off\n\ts_add_u32 s22, s22, 128\n\ts_addc_u32 s23, s23, 0\n\t"
#define G1 MFMA("v[100:115]") DSRD
#define G2 MFMA("v[116:131]") DSRD
#define G3 MFMA("v[132:147]") DSRD
#define G4 MFMA("v[148:163]") DSRD

template <int EVERY, int SHAPE>
__global__ __launch_bounds__(256) void k(long long* out, int reps, const char* gbuf, long long span) {
    __shared__ char lds[32768];
    for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t off;
    unsigned step;
    if (SHAPE == 8) { off = (size_t)(lane >> 3) * 8192 + (lane & 7) * 16; step = 128; }
    else if (SHAPE == 16) { off = (size_t)(lane >> 2) * 8192 + (lane & 3) * 16; step = 64; }
    else { off = (size_t)lane * 16; step = 1024; }
    const char* a = gbuf + ((size_t)blockIdx.x * 4 + wave) * 16 * 8192 % span + off;
    asm volatile("v_mov_b32 v196, %0\n\tv_mov_b32 v197, %1\n\tv_mov_b32 v198, %2\n\tv_lshlrev_b32 v194, 4, %3\n\ts_mov_b32 m0, 0\n\t"
                 ::"v"((unsigned)(size_t)a), "v"((unsigned)((size_t)a >> 32)), "v"(step), "v"(lane)
                 : "v194", "v196", "v197", "v198", "m0");
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (EVERY == 0)
            asm volatile(G1 G2 G3 G4 G1 G2 G3 G4 ::: "memory", "vcc", "m0", "v188", "v189", "v190", "v191");
        else if (EVERY == 4)
            asm volatile(G1 G2 G3 G4 DMA G1 G2 G3 G4 DMA ::: "memory", "vcc", "m0", "v188", "v189", "v190", "v191", "v196", "v197");
        else if (EVERY == 2)
            asm volatile(G1 G2 DMA G3 G4 DMA G1 G2 DMA G3 G4 DMA ::: "memory", "vcc", "m0", "v188", "v189", "v190", "v191", "v196", "v197");
        else
            asm volatile(G1 DMA G2 DMA G3 DMA G4 DMA G1 DMA G2 DMA G3 DMA G4 DMA ::: "memory", "vcc", "m0", "v188", "v189", "v190", "v191", "v196", "v197");
        if ((r & 15) == 15) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int FORM>
__global__ __launch_bounds__(256) void kf(long long* out, int reps, const char* gbuf, long long span) {
    __shared__ char lds[32768];
    for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned voff = (unsigned)((lane >> 3) * 8192 + (lane & 7) * 16);
    const char* base = gbuf + ((size_t)blockIdx.x * 4 + wave) * 16 * 8192 % span;
    const char* a = base + voff;
    asm volatile("v_mov_b32 v196, %0\n\tv_mov_b32 v197, %1\n\tv_mov_b32 v199, %2\n\tv_lshlrev_b32 v194, 4, %3\n\ts_mov_b32 m0, 0\n\t"
                 "s_mov_b32 s20, %4\n\ts_mov_b32 s21, %5\n\ts_mov_b64 s[22:23], 0\n\tv_mov_b32 v202, 0\n\t"
                 ::"v"((unsigned)(size_t)a), "v"((unsigned)((size_t)a >> 32)), "v"(voff), "v"(lane), "s"(__builtin_amdgcn_readfirstlane((unsigned)(size_t)base)), "s"(__builtin_amdgcn_readfirstlane((unsigned)((size_t)base >> 32)))
                 : "v194", "v196", "v197", "v199", "v202", "m0", "s20", "s21", "s22", "s23");
#define CLOB "memory", "vcc", "m0", "v188", "v189", "v190", "v191", "v200", "v201", "s20", "s21", "s22", "s23", "s24"
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
        if (FORM == 0) asm volatile(G1 G2 DMA0 G3 G4 DMA0 G1 G2 DMA0 G3 G4 DMA0 ::: CLOB);
        else if (FORM == 1) asm volatile(G1 G2 DMA1 G3 G4 DMA1 G1 G2 DMA1 G3 G4 DMA1 ::: CLOB);
        else if (FORM == 2) asm volatile(G1 G2 DMA2 G3 G4 DMA2 G1 G2 DMA2 G3 G4 DMA2 ::: CLOB);
        else if (FORM == 4) asm volatile(G1 G2 DMA4 G3 G4 DMA4 G1 G2 DMA4 G3 G4 DMA4 ::: CLOB);
        else asm volatile(G1 G2 DMA5 G3 G4 DMA5 G1 G2 DMA5 G3 G4 DMA5 ::: CLOB);
        if ((r & 15) == 15) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int FORM>
void runf(const char* name, long long* dout, const char* gbuf, long long span) {
    const int reps = 512;
    for (int G : {1, 256}) {
        hipLaunchKernelGGL((kf<FORM>), dim3(G), dim3(256), 0, 0, dout, reps, gbuf, span);
        hipLaunchKernelGGL((kf<FORM>), dim3(G), dim3(256), 0, 0, dout, reps, gbuf, span);
        long long cyc = 0;
        (void)hipMemcpy(&cyc, dout, sizeof(cyc), hipMemcpyDeviceToHost);
        printf("%-44s workgroups %3d   cycles per {MFMA + ds_read} group %6.1f\n", name, G, (double)cyc / (reps * 8.0));
    }
}

template <int EVERY, int SHAPE>
void run(const char* name, long long* dout, const char* gbuf, long long span) {
    const int reps = 512;    // x 8 groups; addresses advance <= 512 * 8 * 1 KiB
    for (int G : {1, 256}) {
        hipLaunchKernelGGL((k<EVERY, SHAPE>), dim3(G), dim3(256), 0, 0, dout, reps, gbuf, span);
        hipLaunchKernelGGL((k<EVERY, SHAPE>), dim3(G), dim3(256), 0, 0, dout, reps, gbuf, span);
        long long cyc = 0;
        (void)hipMemcpy(&cyc, dout, sizeof(cyc), hipMemcpyDeviceToHost);
        printf("%-44s workgroups %3d   cycles per {MFMA + ds_read} group %6.1f\n", name, G, (double)cyc / (reps * 8.0));
    }
}

int main() {
    long long* dout;
    (void)hipMalloc(&dout, 64);
    const long long span = 24ll << 20;                 // 24 MiB window: L2 + infinity cache resident
    char* gbuf;
    (void)hipMalloc(&gbuf, span + (16ll << 20));
    (void)hipMemset(gbuf, 0, span + (16ll << 20));
    run<0, 8>("no DMA", dout, gbuf, span);
    run<4, 1>("1 KiB contiguous every 4th MFMA", dout, gbuf, span);
    run<4, 8>("8 rows x 128 B every 4th MFMA", dout, gbuf, span);
    run<4, 16>("16 rows x 64 B every 4th MFMA", dout, gbuf, span);
    run<2, 1>("1 KiB contiguous every 2nd MFMA", dout, gbuf, span);
    run<2, 8>("8 rows x 128 B every 2nd MFMA", dout, gbuf, span);
    run<2, 16>("16 rows x 64 B every 2nd MFMA", dout, gbuf, span);
    run<1, 8>("8 rows x 128 B every MFMA", dout, gbuf, span);
    run<1, 16>("16 rows x 64 B every MFMA", dout, gbuf, span);
    printf("addressing forms, 8 rows x 128 B every 2nd MFMA:\n");
    runf<0>("F0 fixed address, fixed M0", dout, gbuf, span);
    runf<1>("F1 fixed address, M0 stepped", dout, gbuf, span);
    runf<2>("F2 SGPR base + VGPR offset, all scalar", dout, gbuf, span);
    runf<4>("F4 v_lshl_add_u64 fresh address", dout, gbuf, span);
    runf<5>("F5 F4 + M0 from a VGPR lane", dout, gbuf, span);
    return 0;
}
