// raster_common.h -- helpers shared by the forward and backward rasterizer translation units.
#pragma once
#include "dgs_device.h"
#include "raster_state.h"
#include "dgs_raster.h"

namespace dgs {

// Per-tile measurement counters of the two blend kernels (ImageState::tile_stats: cell-list entries walked, wave loop trips): compiled
// into the tools' library (-DDGS_INSTRUMENT, lib/libdgs_hip_instr.so) and the emulator build only.  In the product library every
// `if constexpr (kRasterStats)` branch is discarded -- no counters, no cross-lane reduce, no store in the hottest loops of the
// rasterizer; tile_stats then stays zero-filled.  bench.py counts a call's pair evaluations with the tools' library and times the
// product library.
#if defined(DGS_INSTRUMENT) || defined(HIPEMU)
constexpr bool kRasterStats = true;
#else
constexpr bool kRasterStats = false;
#endif

// Ablation switches of the backward blend kernels, tools' library only (DGS_RASTER_BWD_ABLATE, a bit set; results are WRONG by design):
// 1 no global atomics / slot stores, 2 no LDS adds, 4 no walk at all, 16 no cross-lane reduction.  What a part costs = time with - time without.
#if defined(DGS_INSTRUMENT)
constexpr bool kRasterAblate = true;
#else
constexpr bool kRasterAblate = false;
#endif

// auxiliary.h:46-56 (getRect): the tile rectangle [x0, x1) x [y0, y1) of a Gaussian from its pixel position and radius
__device__ __forceinline__ void tile_rect(float px, float py, int radius, int gx, int gy, int* x0, int* y0, int* x1, int* y1) {
    const float r = (float)radius;
    *x0 = min(gx, max(0, f2i_sat((px - r) / (float)kTile)));
    *y0 = min(gy, max(0, f2i_sat((py - r) / (float)kTile)));
    *x1 = min(gx, max(0, f2i_sat((px + r + (float)(kTile - 1)) / (float)kTile)));
    *y1 = min(gy, max(0, f2i_sat((py + r + (float)(kTile - 1)) / (float)kTile)));
}

// ---- small column-major 3x3 helper with glm's product order (type_mat3x3.inl:486-519) ----
struct M3 { float c[3][3]; };
__device__ __forceinline__ M3 m3_cols(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7, float a8) {
    M3 m;
    m.c[0][0] = a0; m.c[0][1] = a1; m.c[0][2] = a2;
    m.c[1][0] = a3; m.c[1][1] = a4; m.c[1][2] = a5;
    m.c[2][0] = a6; m.c[2][1] = a7; m.c[2][2] = a8;
    return m;
}
__device__ __forceinline__ M3 m3_mul(const M3& A, const M3& B) {
    M3 R;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i)
            R.c[j][i] = A.c[0][i] * B.c[j][0] + A.c[1][i] * B.c[j][1] + A.c[2][i] * B.c[j][2];
    return R;
}
__device__ __forceinline__ M3 m3_t(const M3& A) {
    M3 R;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) R.c[j][i] = A.c[i][j];
    return R;
}


// computeCov3D, forward.cu:118-152, with the activations of gs_core.py:330-334 fused when `raw_act` (exp of the scales, F.normalize of
// the quaternion).  Called by the forward's preprocess and -- once per (set, Gaussian), the covariance does not depend on the view -- by
// preprocess_backward_kernel: the same instruction sequence (the files are built without contraction), so the backward sees the
// forward's bits without a 24-byte copy per (view, Gaussian) going through memory.
__device__ __forceinline__ void cov3d_from_scale_rot(const float* scales3, const float* rot4, bool raw_act, float scale_mod, float* c6) {
    float sx = scales3[0], sy = scales3[1], sz = scales3[2];
    float qr = rot4[0], qx = rot4[1], qy = rot4[2], qz = rot4[3];
    if (raw_act) {
        sx = det_expf(sx); sy = det_expf(sy); sz = det_expf(sz);
        const float nrm = fmaxf(sqrtf(qr * qr + qx * qx + qy * qy + qz * qz), 1e-12f);
        qr = qr / nrm; qx = qx / nrm; qy = qy / nrm; qz = qz / nrm;
    }
    M3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
    S.c[0][0] = scale_mod * sx; S.c[1][1] = scale_mod * sy; S.c[2][2] = scale_mod * sz;
    const float r = qr, x = qx, y = qy, z = qz;
    const M3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                         2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                         2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
    const M3 Mm = m3_mul(S, R);
    const M3 Sig = m3_mul(m3_t(Mm), Mm);
    c6[0] = Sig.c[0][0]; c6[1] = Sig.c[0][1]; c6[2] = Sig.c[0][2];
    c6[3] = Sig.c[1][1]; c6[4] = Sig.c[1][2]; c6[5] = Sig.c[2][2];
}

// Which of a tile's sixteen 4 x 4 pixel cells can a Gaussian touch at all?
// A (pixel, Gaussian) pair contributes only if cut <= power <= 0 with power = -0.5 (A dx^2 + C dy^2) - B dx dy, i.e. inside
// the ellipse A dx^2 + 2 B dx dy + C dy^2 <= -2 cut, whose bounding box has half extents sqrt(q C / det), sqrt(q A / det).
// The box is inflated (1 % of q, 0.1 % + 0.01 px of the extents) against the rounding of the fp32 `power`; conics that are
// not comfortably positive definite (condition number above 1e4, non-finite values) get all sixteen bits.  CONSERVATIVE by
// construction: a cell without its bit holds no pixel that passes the exact per-pixel test, so skipping it changes no
// result bit.  (The per-tile LIST still is the reference's -- radius = ceil(3 sigma_max) rectangles; a small Gaussian is in
// the lists of tiles its ellipse never reaches.)
// Bit 4 g + c = cell column c (pixels 4c .. 4c+3) of strip g (rows 4g .. 4g+3) of the tile at (tx0, ty0).
__device__ __forceinline__ unsigned cell_mask(float2 xy, float4 co, float cut, float tx0, float ty0) {
    const float A = co.x, B = co.y, C = co.z;
    if (cut > 0.0f) return 0u;
    const float det = A * C - B * B, tr = A + C;
    const float q = -2.0f * cut * 1.01f + 0.01f;
    if (!(det > 0.0f) || !(A > 0.0f) || !(C > 0.0f) || !(tr * tr < 1.0e4f * det) || !(q < 1.0e30f)) return 0xFFFFu;
    const float hx = sqrtf(q * C / det) * 1.001f + 0.01f, hy = sqrtf(q * A / det) * 1.001f + 0.01f;
    if (!(hx < 1.0e30f) || !(hy < 1.0e30f) || !(xy.x == xy.x) || !(xy.y == xy.y)) return 0xFFFFu;
    unsigned xm = 0u, m = 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float xs = tx0 + 4.0f * (float)c;
        if (!(xy.x + hx < xs) && !(xy.x - hx > xs + 3.0f)) xm |= 1u << c;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float ys = ty0 + 4.0f * (float)g;
        if (!(xy.y + hy < ys) && !(xy.y - hy > ys + 3.0f)) m |= xm << (4 * g);
    }
    return m;
}

#ifndef HIPEMU
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
#endif

// Sum over each 16-lane row of a wave; the total of a row is valid in its lane 15 only.
__device__ __forceinline__ float row_sum_to_lane15(float v) {
#ifdef HIPEMU
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
#else
    v += dpp_mov<0x111>(v);   // row_shr:1
    v += dpp_mov<0x112>(v);   // row_shr:2
    v += dpp_mov<0x114>(v);   // row_shr:4
    v += dpp_mov<0x118>(v);   // row_shr:8
    return v;
#endif
}

// Sum over each 8-lane group of a wave; the total of a group is valid in its lane 7 only.
__device__ __forceinline__ float oct_sum_to_lane7(float v) {
#ifdef HIPEMU
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
#else
    v += dpp_mov<0x111>(v);   // row_shr:1
    v += dpp_mov<0x112>(v);   // row_shr:2
    v += dpp_mov<0x114>(v);   // row_shr:4 (lanes 8..11 of a row pick up the other group's lanes 4..7: never lane 7 or 15)
    return v;
#endif
}

// two fp32 values per lane: arithmetic on them is ONE packed instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32, gfx90a+)
typedef float v2f __attribute__((ext_vector_type(2)));

// fp32 add into LDS, no return value (ds_add_f32).
__device__ __forceinline__ void lds_add(float* addr, float v) {
#ifdef HIPEMU
    atomicAdd(addr, v);
#else
    __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP, false);
#endif
}

// Launch order of the per-tile kernels, by one workgroup of 1,024 threads: order[b] = the (view, tile) workgroup b takes.
// The blend / sort kernels of a call are all resident at once at 4 views of 256^2 (1,024 workgroups, 4 per CU) and workgroup
// b lands on CU b mod 256: in tile order every CU would get the SAME tile of all four views -- the image centre four times on
// one CU, a corner four times on another (longest CU 1.6x the mean in the trained-like regime).  Tiles are ranked by `work`
// instead (counting sort on 1,024 classes, most work first) and dealt out boustrophedon, 256 at a time: under the same
// placement the longest CU is 1.02x the mean; with more workgroups than slots most-work-first is the usual greedy order.
// Only the schedule depends on it, never a result.  s_class: 1,024 words of LDS, scratch: 17.
__device__ __forceinline__ void deal_tiles(const uint32_t* work, int n, uint32_t most, uint32_t* order, uint32_t* s_class, uint32_t* scratch) {
    const unsigned long long top = most ? most : 1u;
    s_class[threadIdx.x] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) atomicAdd(&s_class[1023u - (uint32_t)(min((unsigned long long)work[i], top) * 1023ull / top)], 1u);
    __syncthreads();
    uint32_t tot;
    const uint32_t first = block_exclusive_scan<1024>(s_class[threadIdx.x], scratch, &tot);
    __syncthreads();
    s_class[threadIdx.x] = first;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) {
        const uint32_t r = atomicAdd(&s_class[1023u - (uint32_t)(min((unsigned long long)work[i], top) * 1023ull / top)], 1u);
        const uint32_t q = r >> 8, j = r & 255u, m = min(256u, (uint32_t)n - (q << 8));
        order[(q << 8) + ((q & 1u) ? m - 1u - j : j)] = (uint32_t)i;
    }
}

}  // namespace dgs
