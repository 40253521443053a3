// raster_forward.hip -- forward pass of the MI355X-native 3D-Gaussian rasterizer (gfx950, wave64).
//
// What it computes is fixed by the reference (cuda_rasterizer/forward.cu, rasterizer_impl.cu:198-336);
// HOW is redesigned for CDNA4:
//
//   reference (CUDA)                                   this file
//   -------------------------------------------------  ----------------------------------------------------
//   preprocessCUDA            forward.cu:155-256       preprocess_kernel (+ per-tile instance histogram in LDS)
//   cub InclusiveSum over P + D2H sync  impl.cu:277-281 scan_tiles_kernel over V*T tile counts (ranges fall out of it;
//                                                       identifyTileRanges / its memset disappear)
//   duplicateWithKeys: 64-bit (tile|depth) keys :70-111 emit_instances_kernel: 32-bit depth RANK per instance, grouped by tile
//   cub RadixSortPairs over N 64-bit keys  impl.cu:303  (a) 4-pass LSD radix sort of the P depth keys (P << N), then
//                                                       (b) tile_sort_kernel: order-independent LDS bitmap of ranks +
//                                                           popcount prefix = sorted tile list, O(len + P/32), no log factor
//   renderCUDA                forward.cu:261-374       blend_forward_kernel (LDS-staged 48-byte records incl. colour,
//                                                       per-Gaussian alpha cut-off so dead wave-iterations skip the exp)
//
// The final per-tile order is the reference's: (tile, depth bits, Gaussian index) -- a stable sort of the depth
// keys breaks ties by index exactly like the reference's stable radix over emission order.
// All V views of a call are processed by the same launches (grid.y / grid.z = view).
#include <string.h>

#include "raster_common.h"

namespace dgs {

struct FwdParams {
    int P, D, M, W, H, V, vps, gx, gy, T;
    const float *bg, *means3D, *shs, *colors_pre, *opac, *scales, *rots, *cov_pre, *viewm, *projm, *campos, *tanfov;
    float tanfovx, tanfovy, scale_mod;
    int prefiltered, raw_act;
    int bin_mode;            // binning form: 0 chosen from the instance statistics (binning_form), else forced: 1 instance list + rank
                             // bitmap sort, 2 per-tile scan, 3 instance list + per-tile bitonic sort in LDS
    int bitonic_cap;         // longest tile list the bitonic form is launched for (LDS entries), 0: form not available
    int bitonic_any;         // 1: the bitonic form was launched ALONE (async call with a forced form): it takes lists of any length --
                             // those beyond its LDS in sorted chunks merged by rank (tile_bitonic_kernel), never another form
    int exact_exp;           // blend exponential: 0 hardware v_exp_f32 (default), 1 det_expf (bit-identical floats with the oracle)
    int debug;               // DgsRasterForwardArgs.debug: also keeps the per-view cov3D copy in the state (inspection)
    int* radii;
    float* out_color;
    GeomState g;
    ImageState im;
    BinningState bn;
};

__device__ __forceinline__ void view_tanfov(const FwdParams& p, int v, float* tx, float* ty) {
    if (p.tanfov) { *tx = p.tanfov[2 * v]; *ty = p.tanfov[2 * v + 1]; }
    else { *tx = p.tanfovx; *ty = p.tanfovy; }
}

// forward.cu:20-71, degree 0..3.  sh points at this Gaussian's M coefficients (3 floats each).
__device__ __forceinline__ void sh_to_rgb(int deg, const float* sh, float dx, float dy, float dz, float* out, unsigned* clamp_bits) {
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    const float x = dx / len, y = dy / len, z = dz / len;
    unsigned bits = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#define SH(k) sh[3 * (k) + c]
        float r = 0.28209479177387814f * SH(0);
        if (deg > 0) {
            r = r - 0.4886025119029199f * y * SH(1) + 0.4886025119029199f * z * SH(2) - 0.4886025119029199f * x * SH(3);
            if (deg > 1) {
                const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + 1.0925484305920792f * xy * SH(4) + -1.0925484305920792f * yz * SH(5) +
                    0.31539156525252005f * (2.0f * zz - xx - yy) * SH(6) + -1.0925484305920792f * xz * SH(7) +
                    0.5462742152960396f * (xx - yy) * SH(8);
                if (deg > 2) {
                    r = r + -0.5900435899266435f * y * (3.0f * xx - yy) * SH(9) + 2.890611442640554f * xy * z * SH(10) +
                        -0.4570457994644658f * y * (4.0f * zz - xx - yy) * SH(11) +
                        0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                        -0.4570457994644658f * x * (4.0f * zz - xx - yy) * SH(13) + 1.445305721320277f * z * (xx - yy) * SH(14) +
                        -0.5900435899266435f * x * (xx - 3.0f * yy) * SH(15);
                }
            }
        }
#undef SH
        r += 0.5f;
        if (r < 0) bits |= 1u << c;
        out[c] = fmaxf(r, 0.0f);
    }
    *clamp_bits = bits;
}

// One Gaussian of one view: everything preprocessCUDA does (forward.cu:155-256).  Returns visibility.
__device__ __forceinline__ bool preprocess_one(const FwdParams& p, int v, int idx, int* rx0, int* ry0, int* rx1, int* ry1, uint32_t* depth_key) {
    const int s = v / p.vps;
    const size_t gi = (size_t)v * p.P + idx;       // per-view state slot
    const size_t si = (size_t)s * p.P + idx;       // per-set input slot
    p.radii[gi] = 0;
    p.g.tiles_touched[gi] = 0;
    p.g.keys[0][gi] = 0xFFFFFFFFu;
    const float* vm = p.viewm + 16 * v;
    const float* pm = p.projm + 16 * v;
    const float mx = p.means3D[3 * si], my = p.means3D[3 * si + 1], mz = p.means3D[3 * si + 2];
    // in_frustum, auxiliary.h:139-164
    const float hx = pm[0] * mx + pm[4] * my + pm[8] * mz + pm[12];
    const float hy = pm[1] * mx + pm[5] * my + pm[9] * mz + pm[13];
    const float hw = pm[3] * mx + pm[7] * my + pm[11] * mz + pm[15];
    const float pw = 1.0f / (hw + 0.0000001f);
    const float projx = hx * pw, projy = hy * pw;
    float tx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
    float ty = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
    const float tz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
    if (tz <= 0.2f) {
        if (p.prefiltered) p.im.totals[1] = DGS_ERR_PREFILTERED_CULLED;
        return false;
    }
    float tanx, tany;
    view_tanfov(p, v, &tanx, &tany);
    const float focal_y = p.H / (2.0f * tany), focal_x = p.W / (2.0f * tanx);   // rasterizer_impl.cu:222-223

    // ---- 3D covariance: precomputed or computeCov3D (forward.cu:118-152) ----
    float c6[6];
    if (p.cov_pre) {
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = p.cov_pre[6 * si + k];
    } else {
        cov3d_from_scale_rot(p.scales + 3 * si, p.rots + 4 * si, p.raw_act != 0, p.scale_mod, c6);
        // the state copy is for inspection only (dgs_raster_state_read "cov3D": the parity tests, which run with `debug`); the backward
        // recomputes the covariance once per Gaussian
        if (p.debug) {
#pragma unroll
            for (int k = 0; k < 6; ++k) p.g.cov3D[6 * gi + k] = c6[k];
        }
    }
    // ---- computeCov2D (forward.cu:74-113) ----
    const float limx = 1.3f * tanx, limy = 1.3f * tany;
    const float txtz = tx / tz, tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    const M3 J = m3_cols(focal_x / tz, 0.0f, -(focal_x * tx) / (tz * tz), 0.0f, focal_y / tz, -(focal_y * ty) / (tz * tz), 0, 0, 0);
    const M3 Wm = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
    const M3 Tm = m3_mul(Wm, J);
    const M3 Vrk = m3_cols(c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]);
    const M3 cov = m3_mul(m3_mul(m3_t(Tm), m3_t(Vrk)), Tm);
    const float ca = cov.c[0][0] + 0.3f, cb = cov.c[0][1], cc = cov.c[1][1] + 0.3f;
    // ---- conic, radius, tile rect (forward.cu:215-236) ----
    const float det = ca * cc - cb * cb;
    if (det == 0.0f) return false;
    const float det_inv = 1.f / det;
    const float conx = cc * det_inv, cony = -cb * det_inv, conz = ca * det_inv;
    const float mid = 0.5f * (ca + cc);
    const float l1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
    const float l2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
    const float my_radius = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
    const float px = (float)(((projx + 1.0) * p.W - 1.0) * 0.5);   // ndc2Pix, auxiliary.h:41-44 (double)
    const float py = (float)(((projy + 1.0) * p.H - 1.0) * 0.5);
    const int irad = f2i_sat(my_radius);
    int x0, y0, x1, y1;
    tile_rect(px, py, irad, p.gx, p.gy, &x0, &y0, &x1, &y1);
    if ((x1 - x0) * (y1 - y0) == 0) return false;
    // ---- colour (forward.cu:238-246) ----
    float rgb[3];
    unsigned cbits = 0;
    if (p.colors_pre) {
        rgb[0] = p.colors_pre[3 * si]; rgb[1] = p.colors_pre[3 * si + 1]; rgb[2] = p.colors_pre[3 * si + 2];
    } else {
        const float* cam = p.campos + 3 * v;
        sh_to_rgb(p.D, p.shs + 3 * (size_t)p.M * si, mx - cam[0], my - cam[1], mz - cam[2], rgb, &cbits);
    }
    float op = p.opac[si];
    if (p.raw_act) op = 1.0f / (1.0f + det_expf(-op));   // torch.sigmoid, gs_core.py:334
    // Alpha cut-off: alpha = min(.99, op*exp(power)) < 1/255 whenever power < log(1/(255 op)) - margin; lets the blend
    // skip the exponential for wave-iterations no lane needs.  Conservative by construction => results unchanged.
    const float cut = (op > 0.0f) ? (__logf(1.0f / (255.0f * op)) - 0.004f) : __builtin_inff();
    // ---- stores (forward.cu:248-255) ----
    p.g.depths[gi] = tz;
    p.radii[gi] = irad;
    p.g.means2D[gi] = make_float2(px, py);
    BlendRecord* rec = p.g.blend + gi;
    rec->co = make_float4(conx, cony, conz, op);
    rec->rc = make_float4(rgb[0], rgb[1], rgb[2], cut);
    rec->xy = make_float2(px, py);
    p.g.clamped[gi] = (uint8_t)cbits;
    p.g.tiles_touched[gi] = (uint32_t)((y1 - y0) * (x1 - x0));
    p.g.keys[0][gi] = __float_as_uint(tz);
    *depth_key = __float_as_uint(tz);
    *rx0 = x0; *ry0 = y0; *rx1 = x1; *ry1 = y1;
    return true;
}

// grid (ceil(P/256), V), 256 threads.  LDS_HIST: per-block difference grid of the tile counts in LDS (T*4 bytes), flushed with
// one global atomic per non-zero entry; otherwise straight global atomics.
template <bool LDS_HIST>
__global__ __launch_bounds__(256) void preprocess_kernel(FwdParams p) {
    DGS_DYNAMIC_LDS(smem);
    uint32_t* lhist = reinterpret_cast<uint32_t*>(smem);
    const int v = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (LDS_HIST) {
        for (int i = threadIdx.x; i < p.T; i += 256) lhist[i] = 0;
        __syncthreads();
    }
    if (blockIdx.x == 0) {                                       // the depth range sort's bucket counts of this view (range_count_kernel adds to them)
        uint32_t* w = p.g.range_ws + (size_t)v * range_ws_stride(p.P);
        for (int i = threadIdx.x; i <= kRangeBuckets; i += 256) w[i] = 0u;
    }
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    bool vis = false;
    uint32_t depth_key = 0xFFFFFFFFu;
    if (idx < p.P) vis = preprocess_one(p, v, idx, &x0, &y0, &x1, &y1, &depth_key);
    {   // range of the view's depth keys that are not the culled key (range_*_kernel map it onto their coarse buckets): a max of the key
        // and of its complement over the workgroup, then ONE integer atomic each, without a return value, into copy blockIdx.x %
        // kRangeSlots of the view's two words.  (One copy: 4,096 atomics per word serialise at the memory side; an atomic per wave: 16 k,
        // 0.3 ms; a read first and the atomic only if it raises the word: the read's round trip at the end of every workgroup, +14 us on
        // this 60 us kernel.)
        __shared__ uint32_t s_rng[2][4];
        const uint32_t key = vis ? depth_key : 0xFFFFFFFFu;
        uint32_t hi = key != 0xFFFFFFFFu ? key : 0u, lo_c = key != 0xFFFFFFFFu ? ~key : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { hi = max(hi, (uint32_t)__shfl_xor((int)hi, o)); lo_c = max(lo_c, (uint32_t)__shfl_xor((int)lo_c, o)); }
        if ((threadIdx.x & 63) == 0) { s_rng[0][threadIdx.x >> 6] = hi; s_rng[1][threadIdx.x >> 6] = lo_c; }
        __syncthreads();
        if (threadIdx.x < 2) {
            const uint32_t m = max(max(s_rng[threadIdx.x][0], s_rng[threadIdx.x][1]), max(s_rng[threadIdx.x][2], s_rng[threadIdx.x][3]));
            if (m) atomicMax(p.im.depth_range + ((size_t)threadIdx.x * p.V + v) * kRangeSlots + (blockIdx.x % kRangeSlots), m);
        }
    }
    // The number of rectangles that cover a tile is the 2-D prefix sum of a difference grid with +1 at a rectangle's (y0, x0) and
    // (y1, x1) and -1 at (y0, x1) and (y1, x0) (corners on the grid's far edges have nothing behind them and are dropped): FOUR
    // atomics per Gaussian instead of one per covered tile.  With random-init weights a Gaussian covers ~50 of the 256 tiles: 13 M
    // LDS atomics per view in a divergent loop were half of this kernel.  uint32 arithmetic is modular, the counts come out exact;
    // scan_tiles_kernel turns the grid into counts (in place) before anything reads them.
    uint32_t* gcount = p.im.tile_count + (size_t)v * p.T;
    if (vis) {
        uint32_t* const grid = LDS_HIST ? lhist : gcount;
        atomicAdd(&grid[y0 * p.gx + x0], 1u);
        if (x1 < p.gx) atomicAdd(&grid[y0 * p.gx + x1], 0xffffffffu);
        if (y1 < p.gy) {
            atomicAdd(&grid[y1 * p.gx + x0], 0xffffffffu);
            if (x1 < p.gx) atomicAdd(&grid[y1 * p.gx + x1], 1u);
        }
    }
    if (LDS_HIST) {
        __syncthreads();
        for (int i = threadIdx.x; i < p.T; i += 256) {
            const uint32_t c = lhist[i];
            if (c) atomicAdd(&gcount[i], c);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Depth radix sort of the P Gaussians of every view: 4 LSD passes of 8 bits, stable.
// Pass = histogram (grid NB x V) -> column scan (grid 1 x V) -> scatter (grid NB x V).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void radix_hist_kernel(const uint32_t* keys, uint32_t* hist, int P, int NB, int shift) {
    __shared__ uint32_t h[256];
    const int v = blockIdx.y, b = blockIdx.x;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t* k = keys + (size_t)v * P;
    const int base = b * kSortTile;
#pragma unroll 4
    for (int r = 0; r < kSortItems; ++r) {
        const int i = base + r * 256 + threadIdx.x;
        if (i < P) atomicAdd(&h[(k[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[((size_t)v * NB + b) * 256 + threadIdx.x] = h[threadIdx.x];
}

// thread d owns digit d: turns hist[v][b][d] into the exclusive prefix over blocks and writes the
// exclusive scan of the digit totals to base[v][d].
__global__ __launch_bounds__(256) void radix_colscan_kernel(uint32_t* hist, uint32_t* base, int NB) {
    __shared__ uint32_t scratch[8];
    const int v = blockIdx.y, d = threadIdx.x;
    uint32_t* h = hist + (size_t)v * NB * 256;
    uint32_t run = 0;
    for (int b0 = 0; b0 < NB; b0 += 16) {                      // sixteen loads per round trip: the walk is a latency chain
        uint32_t c[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u] = (b0 + u < NB) ? h[(size_t)(b0 + u) * 256 + d] : 0u;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (b0 + u < NB) h[(size_t)(b0 + u) * 256 + d] = run;
            run += c[u];
        }
    }
    uint32_t total;
    const uint32_t ex = block_exclusive_scan<256>(run, scratch, &total);
    base[v * 256 + d] = ex;
}

// Stable scatter.  Round r handles keys base + r*256 + tid (coalesced); inside a round the order is thread order:
// same-digit lanes of a wave are ranked with ballots, waves are chained through LDS counters.
__global__ __launch_bounds__(256) void radix_scatter_kernel(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                                                           uint32_t* vals_out, uint32_t* rank_of, const uint32_t* hist,
                                                           const uint32_t* base, int P, int NB, int shift) {
    __shared__ uint32_t wcount[4][256];
    __shared__ uint32_t running[256];
    const int v = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const size_t vo = (size_t)v * P;
    running[tid] = base[v * 256 + tid] + hist[((size_t)v * NB + b) * 256 + tid];
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < kSortItems; ++r) {
#pragma unroll
        for (int w = 0; w < 4; ++w) wcount[w][tid] = 0;
        __syncthreads();
        const int i = b * kSortTile + r * 256 + tid;
        const bool valid = i < P;
        const uint32_t key = valid ? keys_in[vo + i] : 0xFFFFFFFFu;
        const uint32_t val = valid ? (vals_in ? vals_in[vo + i] : (uint32_t)i) : 0u;
        const uint32_t dig = (key >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long m = __ballot((dig >> bit) & 1u);
            peers &= ((dig >> bit) & 1u) ? m : ~m;
        }
        const uint32_t rank_in_wave = (uint32_t)__popcll(peers & lt_mask);
        if (valid && rank_in_wave == 0) wcount[wave][dig] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t pos = running[dig] + rank_in_wave;
            for (int w = 0; w < wave; ++w) pos += wcount[w][dig];
            keys_out[vo + pos] = key;
            vals_out[vo + pos] = val;
            if (rank_of) rank_of[vo + val] = pos;
        }
        __syncthreads();
        running[tid] += wcount[0][tid] + wcount[1][tid] + wcount[2][tid] + wcount[3][tid];
    }
}

// ------------------------------------------------------------------------------------------------
// Depth RANGE sort (round 6): the same order as the radix sort above -- (depth bits, Gaussian index), the culled key 0xFFFFFFFF last --
// from four plain kernels, none of which waits for another workgroup:
//   range_count_kernel    (grid blocks x V)  every key's coarse bucket under the monotone map of [min, max] of the view's keys onto
//                                            kRangeBuckets buckets; LDS histogram, one integer atomic per touched bucket and block
//   range_plan_kernel     (grid V)           exclusive scan of the bucket counts -> bucket starts; the culled keys' first ranks per block
//   range_scatter_kernel  (grid blocks x V)  (key, index) pairs into their bucket's span (arrival order inside a bucket is arbitrary);
//                                            culled keys straight to their final ranks, in index order
//   range_sort_kernel     (grid kRangeBuckets / kRangeGroup x V)  a workgroup sorts kRangeGroup consecutive buckets' pairs completely in
//                                            LDS as 64-bit (key << 32 | index) keys -- the per-tile sorter's algorithm (tile_bitonic_kernel):
//                                            a local bucket map, every key counts the smaller keys of its bucket, a bitonic network when
//                                            the depths pile up, chunks merged by rank when a group outgrows the LDS -- and writes
//                                            order[rank] and rank_of[index]
// The map is monotone and the final sort is total, so the result does not depend on the order the atomics arrived in: the same bits as
// the four-pass radix (every binning form's tests run on it).  1 M keys (4 views of P = 262,146): 50 us against 113 us for four one-kernel
// radix passes whose workgroups waited for each other inside the launch (round 5's form, removed with this: profiles/r06_depth_sort_ab.txt);
// the three-kernel radix pass above stays as DGS_RASTER_SORT=radix (cross-check, and P > 2 M).
// ------------------------------------------------------------------------------------------------
struct RangeMap {
    uint32_t lo; float scale; bool any;
    __device__ __forceinline__ uint32_t bucket(uint32_t key) const { return min((uint32_t)((float)(key - lo) * scale), (uint32_t)kRangeBuckets - 1u); }
};
// (every thread of a 256-thread workgroup calls it: the 2 x kRangeSlots partial copies are combined through `s_map`)
__device__ __forceinline__ RangeMap range_map(const uint32_t* depth_range, int V, int v, uint32_t* s_map /* LDS, 2 words */) {
    if (threadIdx.x < 64) {
        const int which = threadIdx.x >> 5, slot = threadIdx.x & 31;
        uint32_t x = slot < kRangeSlots ? depth_range[((size_t)which * V + v) * kRangeSlots + slot] : 0u;
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) x = max(x, (uint32_t)__shfl_xor((int)x, o));
        if (slot == 0) s_map[which] = x;
    }
    __syncthreads();
    const uint32_t hi_w = s_map[0], lo_w = s_map[1];
    const uint32_t hi = hi_w, lo = ~lo_w;
    RangeMap m;
    m.any = (hi_w | lo_w) != 0u;
    m.lo = m.any ? lo : 0u;
    // float conversion, multiplication by a positive constant and truncation are monotone
    m.scale = m.any ? (float)kRangeBuckets / ((float)(hi - lo) + 1.0f) : 0.f;
    return m;
}

__global__ __launch_bounds__(256) void range_count_kernel(const uint32_t* keys, const uint32_t* depth_range, uint32_t* ws, int P, int V, int stride) {
    __shared__ uint32_t h[kRangeBuckets];
    __shared__ uint32_t s_c[4];
    __shared__ uint32_t s_map[2];
    const int v = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
    const RangeMap m = range_map(depth_range, V, v, s_map);
    for (int i = tid; i < kRangeBuckets; i += 256) h[i] = 0;
    __syncthreads();
    const uint32_t* k = keys + (size_t)v * P;
    uint32_t culled = 0;
#pragma unroll 4
    for (int r = 0; r < kRangeItems; ++r) {
        const int i = b * kRangeTile + r * 256 + tid;
        if (i < P) {
            const uint32_t key = k[i];
            if (key != 0xFFFFFFFFu) atomicAdd(&h[m.bucket(key)], 1u); else ++culled;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) culled += (uint32_t)__shfl_xor((int)culled, o);
    if ((tid & 63) == 0) s_c[tid >> 6] = culled;
    __syncthreads();
    uint32_t* w = ws + (size_t)v * stride;
    for (int i = tid; i < kRangeBuckets; i += 256) { const uint32_t c = h[i]; if (c) atomicAdd(&w[i], c); }
    if (tid == 0) w[2 * kRangeBuckets + 1 + b] = (s_c[0] + s_c[1]) + (s_c[2] + s_c[3]);
}

__global__ __launch_bounds__(1024) void range_plan_kernel(uint32_t* ws, int NB, int NWIN, int stride) {
    __shared__ uint32_t scratch[20];
    __shared__ uint32_t s_start[kRangeBuckets + 1];
    uint32_t* w = ws + (size_t)blockIdx.x * stride;
    const int tid = threadIdx.x;
    const uint32_t c = tid < kRangeBuckets ? w[tid] : 0u;
    uint32_t nvis;
    const uint32_t ex = block_exclusive_scan<1024>(c, scratch, &nvis);
    __syncthreads();
    if (tid < kRangeBuckets) { w[tid] = ex; w[kRangeBuckets + 1 + tid] = 0u; s_start[tid] = ex; }
    if (tid == 0) { w[kRangeBuckets] = nvis; s_start[kRangeBuckets] = nvis; }
    __syncthreads();
    // segments of the final sort: window j = the buckets whose first rank lies in [j, j + 1) * kRangeWindow, i.e. ranks from the first
    // bucket start >= j * kRangeWindow (binary search over the non-decreasing starts) up to the next window's; a bucket larger than a
    // window leaves the windows it spans empty
    for (int j = tid; j <= NWIN; j += 1024) {
        const uint32_t want = (uint32_t)j * (uint32_t)kRangeWindow;
        int lo = 0, hi = kRangeBuckets;                           // first index with s_start[index] >= want (s_start[kRangeBuckets] = nvis ends the search)
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_start[mid] < want) lo = mid + 1; else hi = mid; }
        w[2 * kRangeBuckets + 1 + NB + j] = j == NWIN ? nvis : s_start[lo];
    }
    // culled keys: block j's first rank = keys that are not culled + culled keys of the blocks in front (NB <= 1024: P <= 4 M)
    const uint32_t cc = tid < NB ? w[2 * kRangeBuckets + 1 + tid] : 0u;
    uint32_t all;
    const uint32_t cex = block_exclusive_scan<1024>(cc, scratch, &all);
    if (tid < NB) w[2 * kRangeBuckets + 1 + tid] = nvis + cex;
}

__global__ __launch_bounds__(256) void range_scatter_kernel(const uint32_t* keys, const uint32_t* depth_range, uint32_t* ws, uint32_t* keys_out,
                                                           uint32_t* vals_out, uint32_t* order, uint32_t* rank_of, int P, int V, int stride) {
    __shared__ uint32_t h[kRangeBuckets];
    __shared__ uint32_t s_cc[kRangeItems][4];
    __shared__ uint32_t s_map[2];
    const int v = blockIdx.y, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const RangeMap m = range_map(depth_range, V, v, s_map);
    const size_t vo = (size_t)v * P;
    uint32_t* w = ws + (size_t)v * stride;
    for (int i = tid; i < kRangeBuckets; i += 256) h[i] = 0;
    __syncthreads();
    uint32_t key[kRangeItems], slot[kRangeItems];
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < kRangeItems; ++r) {
        const int i = b * kRangeTile + r * 256 + tid;
        key[r] = i < P ? keys[vo + i] : 0u;
        const bool in = i < P, cul = in && key[r] == 0xFFFFFFFFu;
        slot[r] = (in && !cul) ? atomicAdd(&h[m.bucket(key[r])], 1u) : 0u;
        // culled keys keep their index order: rank inside the wave now, the (round, wave) prefix below
        const unsigned long long cm = __ballot(cul);
        if (cul) slot[r] = (uint32_t)__popcll(cm & lt_mask);
        if (lane == 0) s_cc[r][wave] = (uint32_t)__popcll(cm);
    }
    __syncthreads();
    // a span of its bucket for this block's keys: one returning atomic per touched bucket
    {
        constexpr int Q = kRangeBuckets / 256;                  // buckets per thread: their starts and their atomics all in flight together
        uint32_t c[Q], st0[Q], got[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) { c[q] = h[tid + 256 * q]; st0[q] = w[tid + 256 * q]; }
#pragma unroll
        for (int q = 0; q < Q; ++q) got[q] = c[q] ? atomicAdd(&w[kRangeBuckets + 1 + tid + 256 * q], c[q]) : 0u;
#pragma unroll
        for (int q = 0; q < Q; ++q) h[tid + 256 * q] = st0[q] + got[q];
    }
    __syncthreads();
    const uint32_t cbase = w[2 * kRangeBuckets + 1 + b];
#pragma unroll
    for (int r = 0; r < kRangeItems; ++r) {
        const int i = b * kRangeTile + r * 256 + tid;
        if (i >= P) continue;
        if (key[r] != 0xFFFFFFFFu) {
            const uint32_t pos = h[m.bucket(key[r])] + slot[r];
            keys_out[vo + pos] = key[r];
            vals_out[vo + pos] = (uint32_t)i;
        } else {
            uint32_t off = 0;
            for (int q = 0; q < r; ++q) off += (s_cc[q][0] + s_cc[q][1]) + (s_cc[q][2] + s_cc[q][3]);
            for (int q = 0; q < wave; ++q) off += s_cc[r][q];
            const uint32_t pos = cbase + off + slot[r];
            order[vo + pos] = (uint32_t)i;
            rank_of[vo + i] = pos;
        }
    }
}

// Exclusive scan of the V*T tile counts -> [start,end) ranges into the packed instance list (one workgroup), and the launch
// order of the per-tile kernels (deal_tiles, by list length).
__global__ __launch_bounds__(1024) void scan_tiles_kernel(uint32_t* count, uint2* ranges, uint32_t* cursor, int n,
                                                         int32_t* totals, long long capacity, uint32_t* order, int gx, int gy,
                                                         int32_t* stats_dev, int32_t* stats_host) {
    __shared__ uint32_t scratch[20];
    __shared__ uint32_t smax;
    __shared__ uint32_t s_class[1024];
    // ---- difference grids (preprocess_kernel) -> instance counts, in place: per view a prefix sum along x, then along y.  A thread
    //      owns a row (then a column) of one view; the walks are short (gx, gy <= a few dozen at the shipped resolutions) and all of
    //      a pass's rows are independent ----
    {
        const int T = gx * gy, V = n / T;
        for (int r = threadIdx.x; r < V * gy; r += 1024) {
            uint32_t* row = count + (size_t)(r / gy) * T + (size_t)(r % gy) * gx;
            uint32_t run = 0;
            for (int x0 = 0; x0 < gx; x0 += 16) {                  // sixteen loads per round trip
                uint32_t c[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) c[u] = x0 + u < gx ? row[x0 + u] : 0u;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (x0 + u < gx) { run += c[u]; row[x0 + u] = run; }
            }
        }
        __syncthreads();                                           // one workgroup: its own global writes are visible behind the barrier
        for (int cidx = threadIdx.x; cidx < V * gx; cidx += 1024) {
            uint32_t* col = count + (size_t)(cidx / gx) * T + (cidx % gx);
            uint32_t run = 0;
            for (int y0 = 0; y0 < gy; y0 += 16) {
                uint32_t c[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) c[u] = y0 + u < gy ? col[(size_t)(y0 + u) * gx] : 0u;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (y0 + u < gy) { run += c[u]; col[(size_t)(y0 + u) * gx] = run; }
            }
        }
        __syncthreads();
    }
    uint32_t carry = 0, mx = 0;
    if (threadIdx.x == 0) smax = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const uint32_t c = (i < n) ? count[i] : 0u;
        mx = max(mx, c);
        uint32_t tot;
        const uint32_t ex = block_exclusive_scan<1024>(c, scratch, &tot);
        if (i < n) { ranges[i] = make_uint2(carry + ex, carry + ex + c); cursor[i] = 0; }
        carry += tot;
    }
    __syncthreads();
    atomicMax(&smax, mx);
    __syncthreads();
    if (threadIdx.x == 0) {
        totals[0] = (int32_t)carry;
        totals[2] = (int32_t)smax;
        if (capacity >= 0 && (long long)carry > capacity) totals[1] = DGS_ERR_BINNING_OVERFLOW;
        // the caller's copies of the four words (DgsRasterForwardArgs.num_rendered_dev / num_rendered_host), written by this kernel:
        // the host copy is a store into pinned, device-visible host memory -- no memcpy node in a captured call
        const int32_t st1 = totals[1];
        // word [3] = 1 is written LAST, behind a system-scope fence: a host that polls it (dgs_amd/raster.py: a plan at risk) reads the
        // other three as soon as it turns non-zero -- long before the call's blend kernels have run
        if (stats_dev) { stats_dev[0] = (int32_t)carry; stats_dev[1] = st1; stats_dev[2] = (int32_t)smax; stats_dev[3] = 1; }
        if (stats_host) {
            stats_host[0] = (int32_t)carry; stats_host[1] = st1; stats_host[2] = (int32_t)smax;
            __threadfence_system();
            *reinterpret_cast<volatile int32_t*>(stats_host + 3) = 1;
        }
    }
    deal_tiles(count, n, smax, order, s_class, scratch);
}

// Which binning form runs.  In the sync mode the host has read {num_rendered, longest tile list} back and launches one form
// only; in the async mode every form is launched and the ones whose turn it is not return at once -- both sides evaluate
// THIS function.  scan: at least a tenth of all (tile, Gaussian) pairs are instances (measured crossover at 256^2, 4 views:
// the scan is 0.17 ms faster at a density of 0.2 -- random-init weights -- and slower at 0.012 -- trained-like scenes);
// bitonic: sparse scenes whose longest tile list fits the LDS the kernel was launched with; rank sort: the rest.
enum { kFormRankSort = 1, kFormScan = 2, kFormBitonic = 3 };
__host__ __device__ __forceinline__ int binning_form_of(int bin_mode, int bitonic_cap, long long num_rendered, long long longest,
                                                        long long T, long long P, long long V, bool any_len = false) {
    const bool fits = bitonic_cap > 0 && (any_len || longest <= bitonic_cap);
    if (bin_mode == kFormScan || bin_mode == kFormRankSort) return bin_mode;
    if (bin_mode == kFormBitonic) return fits ? kFormBitonic : kFormRankSort;
    if (num_rendered * 10 >= T * P * V) return kFormScan;
    return fits ? kFormBitonic : kFormRankSort;
}
// the scan is on demand: T x P is its worst case, not its cost; tile coordinates are packed in 8 bits each (rank_rects_kernel)
__host__ __device__ __forceinline__ bool scan_form_possible(int gx, int gy, long long T, long long P) {
    return gx <= 255 && gy <= 255 && T * P <= (1ll << 31);
}
__device__ __forceinline__ int binning_form(const FwdParams& p) {
    return binning_form_of(p.bin_mode, p.bitonic_cap, (long long)(uint32_t)p.im.totals[0], (long long)(uint32_t)p.im.totals[2], p.T, p.P, p.V,
                           p.bitonic_any != 0);
}
__device__ __forceinline__ bool binning_is_scan(const FwdParams& p) { return binning_form(p) == kFormScan; }

// grid (ceil(P/256), V).  Writes, for every (Gaussian, touched tile), the Gaussian's sort key into that tile's segment: its depth
// rank (rank-sort form) or (depth bits << 32 | index) (bitonic form).  Slot order inside a segment is arbitrary (both per-tile
// sorts are order-independent: the keys of a tile are distinct).
template <bool LDS_AGG>
__global__ __launch_bounds__(256) void emit_instances_kernel(FwdParams p) {
    DGS_DYNAMIC_LDS(smem);
    uint32_t* lcnt = reinterpret_cast<uint32_t*>(smem);
    uint32_t* lbase = lcnt + p.T;
    const int form = binning_form(p);
    if (p.im.totals[1] != 0 || form == kFormScan) return;
    const bool wide = form == kFormBitonic;
    const int v = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const size_t gi = (size_t)v * p.P + idx;
    int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
    bool vis = false;
    if (idx < p.P && p.radii[gi] > 0) {
        const float2 m = p.g.means2D[gi];
        tile_rect(m.x, m.y, p.radii[gi], p.gx, p.gy, &x0, &y0, &x1, &y1);
        vis = true;
    }
    const uint32_t rank = (vis && !wide) ? p.g.rank_of[gi] : 0u;
    // the reference's sort key without the tile id (rasterizer_impl.cu:98-109): depth bits, ties by Gaussian index
    const uint64_t key = (vis && wide) ? (((uint64_t)__float_as_uint(p.g.depths[gi]) << 32) | (uint32_t)idx) : 0ull;
    const uint2* ranges = p.im.ranges + (size_t)v * p.T;
    uint32_t* cursor = p.im.tile_cursor + (size_t)v * p.T;
    if (LDS_AGG) {
        for (int i = threadIdx.x; i < p.T; i += 256) lcnt[i] = 0;
        __syncthreads();
        if (vis)
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) atomicAdd(&lcnt[y * p.gx + x], 1u);
        __syncthreads();
        for (int i = threadIdx.x; i < p.T; i += 256) {
            const uint32_t c = lcnt[i];
            // the tile's first slot rides along with the cursor's round trip (as a load of its own in the loop below it was one more
            // dependent trip per instance: the kernel is a chain of them, 0.82 of its wave cycles parked -- profiles/r04_raster_sq_pmc.txt)
            if (c) { lbase[i] = ranges[i].x + atomicAdd(&cursor[i], c); lcnt[i] = 0; }
        }
        __syncthreads();
        if (vis)
            for (int y = y0; y < y1; ++y)
                for (int x = x0; x < x1; ++x) {
                    const int t = y * p.gx + x;
                    const uint32_t slot = lbase[t] + atomicAdd(&lcnt[t], 1u);
                    if (wide) p.bn.inst_key[slot] = key; else p.bn.inst_rank[slot] = rank;
                }
    } else if (vis) {
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                const int t = y * p.gx + x;
                const uint32_t slot = ranges[t].x + atomicAdd(&cursor[t], 1u);
                if (wide) p.bn.inst_key[slot] = key; else p.bn.inst_rank[slot] = rank;
            }
    }
}

// grid (T, V), 256 threads, dynamic LDS = window words * 4.  Sorts one tile's instances by depth rank:
// ranks of one view are distinct integers in [0,P), so a bitmap (atomicOr, order-independent) plus a popcount
// prefix sum IS the sorted sequence.  Windows of `wwords*32` ranks keep LDS bounded for any P.
__global__ __launch_bounds__(256) void tile_sort_kernel(FwdParams p, int wwords) {
    DGS_DYNAMIC_LDS(smem);
    uint32_t* bm = reinterpret_cast<uint32_t*>(smem);
    __shared__ uint32_t scratch[8];
    if (p.im.totals[1] != 0 || binning_form(p) != kFormRankSort) return;
    const int t = blockIdx.x, v = blockIdx.y, tid = threadIdx.x;
    const uint2 rg = p.im.ranges[(size_t)v * p.T + t];
    if (rg.x == rg.y) return;
    const uint32_t* order = p.g.vals[0] + (size_t)v * p.P;   // rank -> Gaussian index (after 4 passes the result is in buffer 0)
    uint32_t emitted = rg.x;
    for (uint32_t w0 = 0; w0 < (uint32_t)p.P; w0 += (uint32_t)wwords * 32u) {
        for (int i = tid; i < wwords; i += 256) bm[i] = 0;
        __syncthreads();
        // 8 independent loads in flight per thread: the loop is latency-bound otherwise (one L2 round trip per 256 instances)
        for (uint32_t i0 = rg.x; i0 < rg.y; i0 += 8u * 256u) {
            uint32_t r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = i0 + (uint32_t)u * 256u + (uint32_t)tid;
                r[u] = i < rg.y ? p.bn.inst_rank[i] - w0 : 0xffffffffu;     // unsigned wrap puts other windows out of range
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r[u] < (uint32_t)wwords * 32u) atomicOr(&bm[r[u] >> 5], 1u << (r[u] & 31u));
        }
        __syncthreads();
        // Emission is wave-cooperative so that stores (and the rank -> index gathers) are coalesced: every wave owns a
        // contiguous quarter of the window's words and expands them 64 bits (two words) per step -- lane l tests bit l & 31
        // of word (l >> 5), a ballot + popcount prefix gives each set lane its output slot.  (One thread walking its own
        // words writes 64 scattered cache lines per store instruction: 34x write amplification on long tile lists.)
        const int lane = tid & 63, wave = tid >> 6;
        const int wpw = wwords / 4;                       // words per wave (wwords is a multiple of 256)
        uint32_t cnt = 0;
        for (int k = lane; k < wpw; k += 64) cnt += (uint32_t)__popc(bm[wave * wpw + k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
        if (lane == 0) scratch[wave] = cnt;
        __syncthreads();
        uint32_t off = emitted, total = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = scratch[w];
            if (w < wave) off += c;
            total += c;
        }
        const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        // sixteen 64-bit chunks per step: the rank -> index gathers of all of them are in flight before the first store needs one
        // (wwords is a multiple of 256, so wpw is a multiple of 32).  A lane-per-word variant (popcount scan + per-lane bit
        // peeling) was measured 30 % slower: its stores and gathers scatter.
        constexpr int EB = 16;
        for (int k = 0; k < wpw; k += 2 * EB) {
            uint32_t slot[EB], val[EB];
            unsigned long long setm = 0;
#pragma unroll
            for (int u = 0; u < EB; ++u) {
                const uint32_t word = bm[wave * wpw + k + 2 * u + (lane >> 5)];
                const bool set = (word >> (lane & 31)) & 1u;
                const unsigned long long m = __ballot(set);
                slot[u] = off + (uint32_t)__popcll(m & lt_mask);
                off += (uint32_t)__popcll(m);
                val[u] = set ? order[w0 + (uint32_t)(wave * wpw + k + 2 * u) * 32u + (uint32_t)lane] : 0u;
                setm |= (unsigned long long)set << u;
            }
#pragma unroll
            for (int u = 0; u < EB; ++u)
                if ((setm >> u) & 1ull) p.bn.point_list[slot[u]] = val[u];
        }
        emitted += total;
        __syncthreads();
    }
}

// grid (T, V), 256 threads, dynamic LDS = cap * 8 + 2 * kBuckets * 4 bytes.  Sparse scenes: a tile's list is a few thousand
// entries, and the rank-sort form above pays O(P / 32) per tile (it expands a bitmap over ALL depth ranks) plus a 12-launch
// radix sort of the P depth keys first.  Here a tile sorts ITS OWN (depth bits << 32 | index) keys in LDS -- O(n) for n = the
// tile's list, no global sort at all, and the keys' order IS the reference's (tile, depth bits, Gaussian index):
//   1. min / max of the tile's depth bits;  2. histogram over B ~ n / 4 buckets under the monotone map
//   b = floor((bits - min) * B / (max - min + 1)) (LDS atomics);  3. exclusive scan -> bucket starts;  4. keys scattered into
//   their buckets in LDS (arrival order inside a bucket is arbitrary);  5. every KEY finds its place inside its bucket -- ~4 keys --
//   by counting the bucket's smaller keys (full 64-bit compare) and leaves for the tile's list from there.  The map is monotone, so
//   bucket order + in-bucket order is the total order; the keys of a tile are distinct, so the result does not depend on the order
//   atomics happened to arrive in.
// A tile whose depths pile up (a bucket above kBucketLimit keys: the counting is quadratic) takes the bitonic network over
// the whole list instead -- O(n log^2 n), data-independent.  (The network alone was measured first: 0.34 ms per 4 views at
// 256^2 and 3.7 ms at 512^2, where 8 k-key tiles need 91 LDS round trips each.)
constexpr int kBuckets = 2048, kBucketLimit = 40;
__device__ __forceinline__ void cswap(uint64_t& a, uint64_t& b, bool up) {
    const bool sw = (a > b) == up;
    const uint64_t t = sw ? b : a;
    b = sw ? a : b;
    a = t;
}
template <int NT>
__device__ __forceinline__ void bitonic_sort_lds(uint64_t* keys, uint32_t m, int tid) {      // m: power of two >= 8 * NT
    for (uint32_t k = 2; k <= m; k <<= 1) {
        uint32_t j = k >> 1;
        for (; j > 4; j >>= 1) {                               // partners further than 4 apart: one compare-exchange per LDS round trip
            for (uint32_t i = tid; i < m / 2; i += NT) {
                const uint32_t lo = ((i / j) * 2 * j) + (i % j), hi = lo + j;
                uint64_t a = keys[lo], b = keys[hi];
                cswap(a, b, (lo & k) == 0);
                keys[lo] = a; keys[hi] = b;
            }
            __syncthreads();
        }
        // strides 4, 2, 1 (those that remain): 8 consecutive keys per thread, in registers
        for (uint32_t c = tid; c < m / 8; c += NT) {
            uint64_t r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) r[u] = keys[8 * c + u];
            const bool up = ((8 * c) & k) == 0;
            if (k >= 8) {
                if (j >= 4) { cswap(r[0], r[4], up); cswap(r[1], r[5], up); cswap(r[2], r[6], up); cswap(r[3], r[7], up); }
                if (j >= 2) { cswap(r[0], r[2], up); cswap(r[1], r[3], up); cswap(r[4], r[6], up); cswap(r[5], r[7], up); }
                cswap(r[0], r[1], up); cswap(r[2], r[3], up); cswap(r[4], r[5], up); cswap(r[6], r[7], up);
            } else if (k == 2) {
                cswap(r[0], r[1], true); cswap(r[2], r[3], false); cswap(r[4], r[5], true); cswap(r[6], r[7], false);
            } else {   // k == 4: strides 2, 1, direction per group of 4
                cswap(r[0], r[2], true); cswap(r[1], r[3], true); cswap(r[4], r[6], false); cswap(r[5], r[7], false);
                cswap(r[0], r[1], true); cswap(r[2], r[3], true); cswap(r[4], r[5], false); cswap(r[6], r[7], false);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) keys[8 * c + u] = r[u];
        }
        __syncthreads();
    }
}

// NT threads per tile: 256 while two or more tiles fit a CU's LDS, 1024 for the long lists that own a CU each (512^2 scenes).
template <int NT>
__global__ __launch_bounds__(NT) void tile_bitonic_kernel(FwdParams p) {
    DGS_DYNAMIC_LDS(smem);
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem + (size_t)p.bitonic_cap * 8);
    uint32_t* cur = cnt + kBuckets;
    constexpr int NWV = NT / 64;
    __shared__ uint32_t s_red[3][NWV];
    __shared__ uint32_t scratch[NWV + 4];
    if (p.im.totals[1] != 0 || binning_form(p) != kFormBitonic) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint2 rg = p.im.ranges[p.im.tile_order[blockIdx.x]];
    const uint32_t n = rg.y - rg.x;
    if (n == 0) return;
    const uint64_t* src = p.bn.inst_key + rg.x;
    constexpr int KPT = NT == 256 ? 32 : 16;                   // 8,192 / 256, 8,192 / 512 (both launched up to that capacity), 16,384 / 1,024
    {
        // A list longer than this launch's LDS (only when the form was launched alone -- `bitonic_any` -- for a caller that expected
        // shorter lists; any other launch sends such a tile to the rank sort): chunks of C keys are sorted in LDS and written back
        // in place, then every key's place in the tile's list is its place in its own chunk + the number of smaller keys in each other
        // chunk (binary search; the keys of a tile are distinct).  O(n (n / C) log C): the rare path of a plan that went stale -- the
        // reference pays a full radix sort of all instances for every call (rasterizer_impl.cu:303).
        const uint32_t C = min((uint32_t)p.bitonic_cap, (uint32_t)(KPT * NT));
        if (n > C) {
            uint64_t* sorted = p.bn.inst_key + rg.x;
            const uint32_t nchunks = (n + C - 1) / C;
            for (uint32_t c = 0; c < nchunks; ++c) {
                const uint32_t len = min(C, n - c * C);
                uint32_t m = 8 * NT;
                while (m < len) m <<= 1;
                for (uint32_t i = tid; i < m; i += NT) keys[i] = i < len ? sorted[c * C + i] : ~0ull;
                __syncthreads();
                bitonic_sort_lds<NT>(keys, m, tid);
                __syncthreads();
                for (uint32_t i = tid; i < len; i += NT) sorted[c * C + i] = keys[i];
                __syncthreads();
            }
            __threadfence_block();                             // one workgroup: its own global writes are visible behind the barrier
            __syncthreads();
            for (uint32_t i = tid; i < n; i += NT) {
                const uint64_t k = sorted[i];
                const uint32_t mine = i / C;
                uint32_t place = i - mine * C;
                for (uint32_t c = 0; c < nchunks; ++c) {
                    if (c == mine) continue;
                    const uint64_t* ch = sorted + c * C;
                    uint32_t lo = 0, hi = min(C, n - c * C);   // first index with ch[index] > k == number of smaller keys (distinct)
                    while (lo < hi) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if (ch[mid] < k) lo = mid + 1; else hi = mid;
                    }
                    place += lo;
                }
                p.bn.point_list[rg.x + place] = (uint32_t)k;
            }
            return;
        }
    }
    // The tile's keys, ONCE, into registers: thread t holds keys t, t + NT, ... (at most KPT = the launch's LDS capacity / NT; all
    // loads in flight together).  The three passes below (range, histogram, scatter) each used to walk `src` with the next load
    // behind the use of the previous one -- ~12 dependent L2 round trips per pass for a 3,000-entry list, 0.67 of the kernel's wave
    // cycles parked (profiles/r04_raster_sq_pmc.txt).
    uint64_t kreg[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const uint32_t i = (uint32_t)tid + (uint32_t)u * NT;
        kreg[u] = i < n ? src[i] : ~0ull;
    }
    // 1. range of the depth bits
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        if ((uint32_t)tid + (uint32_t)u * NT < n) {
            const uint32_t d = (uint32_t)(kreg[u] >> 32);
            lo = min(lo, d); hi = max(hi, d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o)); }
    if (lane == 0) { s_red[0][wave] = lo; s_red[1][wave] = hi; }
    uint32_t B = 256;
    while (B * 4 < n && B < (uint32_t)kBuckets) B <<= 1;
    for (uint32_t i = tid; i < (uint32_t)kBuckets; i += NT) cnt[i] = 0;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NWV; ++w) { lo = min(lo, s_red[0][w]); hi = max(hi, s_red[1][w]); }
    // monotone map of the depth bits onto [0, B): float conversion and the multiplication by a positive constant are monotone
    const float scale = (float)B / ((float)(hi - lo) + 1.0f);
    auto bucket = [&](uint32_t d) { return min((uint32_t)((float)(d - lo) * scale), B - 1u); };
    // 2. histogram
#pragma unroll
    for (int u = 0; u < KPT; ++u)
        if ((uint32_t)tid + (uint32_t)u * NT < n) atomicAdd(&cnt[bucket((uint32_t)(kreg[u] >> 32))], 1u);
    __syncthreads();
    // 3. exclusive scan of the counts (kBuckets / NT consecutive buckets per thread; buckets >= B are empty) and the largest bucket
    const uint32_t per = kBuckets / NT;
    uint32_t local = 0, big = 0;
    for (uint32_t u = 0; u < per; ++u) { const uint32_t c = cnt[tid * per + u]; local += c; big = max(big, c); }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan<NT>(local, scratch, &tot);
    for (uint32_t u = 0; u < per; ++u) { const uint32_t c = cnt[tid * per + u]; cur[tid * per + u] = ex; cnt[tid * per + u] = ex; ex += c; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) big = max(big, (uint32_t)__shfl_xor((int)big, o));
    if (lane == 0) s_red[2][wave] = big;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < NWV; ++w) big = max(big, s_red[2][w]);
    // 4. scatter into the buckets (cnt[] keeps the bucket starts, cur[] are the cursors)
#pragma unroll
    for (int u = 0; u < KPT; ++u)
        if ((uint32_t)tid + (uint32_t)u * NT < n) keys[atomicAdd(&cur[bucket((uint32_t)(kreg[u] >> 32))], 1u)] = kreg[u];
    __syncthreads();
    if (big <= (uint32_t)kBucketLimit) {
        // 5. one thread per KEY: its place inside its bucket is the number of smaller keys there (keys are distinct), and the tile's
        //    list leaves from here.  (One thread per BUCKET running an insertion sort left the workgroup waiting for the thread with the
        //    largest bucket -- a chain of ~s^2 / 2 dependent LDS round trips for s keys, s up to 19 in the trained-like regime -- and
        //    needed one more pass to write the list out.)  cur[b] is the bucket's end by now.
        for (uint32_t i = tid; i < n; i += NT) {
            const uint64_t k = keys[i];
            const uint32_t b = bucket((uint32_t)(k >> 32));
            const uint32_t s0 = cnt[b], e0 = cur[b];
            uint32_t below = 0;
            for (uint32_t j = s0; j < e0; ++j) below += keys[j] < k ? 1u : 0u;
            p.bn.point_list[rg.x + s0 + below] = (uint32_t)k;
        }
        return;
    }
    {
        uint32_t m = 8 * NT;
        while (m < n) m <<= 1;                                 // <= bitonic_cap (the host sized the LDS for it)
        for (uint32_t i = n + tid; i < m; i += NT) keys[i] = ~0ull;
        __syncthreads();
        bitonic_sort_lds<NT>(keys, m, tid);
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += NT) p.bn.point_list[rg.x + i] = (uint32_t)keys[i];
}

// The final stage of the depth range sort (see range_count_kernel): kRangeGroup consecutive coarse buckets of one view, sorted completely.
// grid (kRangeBuckets / kRangeGroup, V), 256 threads, dynamic LDS = kRangeSortCap * 8 + 2 * kBuckets * 4 bytes (two workgroups per CU).
constexpr int kRangeSortCap = 8192;
template <int NT>
__global__ __launch_bounds__(NT) void range_sort_kernel(const uint32_t* ws, uint32_t* pkeys, uint32_t* pvals, uint32_t* order, uint32_t* rank_of,
                                                       int P, int stride, int seg_off) {
    DGS_DYNAMIC_LDS(smem);
    constexpr int NWV = NT / 64, KPT = kRangeSortCap / NT;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem + (size_t)kRangeSortCap * 8);
    uint32_t* cur = cnt + kBuckets;
    __shared__ uint32_t s_red[3][NWV];
    __shared__ uint32_t scratch[NWV + 4];
    const int v = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t* w = ws + (size_t)v * stride;
    const uint32_t begin = w[seg_off + blockIdx.x], end = w[seg_off + blockIdx.x + 1];   // range_plan_kernel's segment table
    const uint32_t n = end - begin;
    if (n == 0) return;
    const size_t vo = (size_t)v * P;
    uint32_t* pk = pkeys + vo + begin;
    uint32_t* pv = pvals + vo + begin;
    auto emit = [&](uint32_t place, uint64_t k) {
        order[vo + begin + place] = (uint32_t)k;
        rank_of[vo + (uint32_t)k] = begin + place;
    };
    if (n > (uint32_t)kRangeSortCap) {
        // depths piled up beyond the LDS (thousands of Gaussians inside a thousandth of the view's depth range): chunks of the pairs are
        // sorted in LDS and written back in place, then a pair's rank is its place in its chunk + the number of smaller pairs in every
        // other chunk (binary search; the (key, index) pairs are distinct).  O(n (n / C) log C): slow, never wrong.
        const uint32_t C = (uint32_t)kRangeSortCap, nchunks = (n + C - 1) / C;
        for (uint32_t c = 0; c < nchunks; ++c) {
            const uint32_t len = min(C, n - c * C);
            uint32_t m = 8 * NT;
            while (m < len) m <<= 1;
            for (uint32_t i = tid; i < m; i += NT) keys[i] = i < len ? ((uint64_t)pk[c * C + i] << 32) | pv[c * C + i] : ~0ull;
            __syncthreads();
            bitonic_sort_lds<NT>(keys, m, tid);
            __syncthreads();
            for (uint32_t i = tid; i < len; i += NT) { pk[c * C + i] = (uint32_t)(keys[i] >> 32); pv[c * C + i] = (uint32_t)keys[i]; }
            __syncthreads();
        }
        __threadfence_block();                                 // one workgroup: its own global writes are visible behind the barrier
        __syncthreads();
        for (uint32_t i = tid; i < n; i += NT) {
            const uint64_t k = ((uint64_t)pk[i] << 32) | pv[i];
            const uint32_t mine = i / C;
            uint32_t place = i - mine * C;
            for (uint32_t c = 0; c < nchunks; ++c) {
                if (c == mine) continue;
                uint32_t lo = 0, hi = min(C, n - c * C);
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    const uint64_t o = ((uint64_t)pk[c * C + mid] << 32) | pv[c * C + mid];
                    if (o < k) lo = mid + 1; else hi = mid;
                }
                place += lo;
            }
            emit(place, k);
        }
        return;
    }
    // the group's pairs, once, into registers (all loads in flight together), then tile_bitonic_kernel's five steps
    uint64_t kreg[KPT];
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        const uint32_t i = (uint32_t)tid + (uint32_t)u * NT;
        kreg[u] = i < n ? ((uint64_t)pk[i] << 32) | pv[i] : ~0ull;
    }
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
#pragma unroll
    for (int u = 0; u < KPT; ++u) {
        if ((uint32_t)tid + (uint32_t)u * NT < n) {
            const uint32_t d = (uint32_t)(kreg[u] >> 32);
            lo = min(lo, d); hi = max(hi, d);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (uint32_t)__shfl_xor((int)lo, o)); hi = max(hi, (uint32_t)__shfl_xor((int)hi, o)); }
    if (lane == 0) { s_red[0][wave] = lo; s_red[1][wave] = hi; }
    uint32_t B = 256;
    while (B * 4 < n && B < (uint32_t)kBuckets) B <<= 1;
    for (uint32_t i = tid; i < (uint32_t)kBuckets; i += NT) cnt[i] = 0;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NWV; ++q) { lo = min(lo, s_red[0][q]); hi = max(hi, s_red[1][q]); }
    const float scale = (float)B / ((float)(hi - lo) + 1.0f);
    auto bucket = [&](uint32_t d) { return min((uint32_t)((float)(d - lo) * scale), B - 1u); };
#pragma unroll
    for (int u = 0; u < KPT; ++u)
        if ((uint32_t)tid + (uint32_t)u * NT < n) atomicAdd(&cnt[bucket((uint32_t)(kreg[u] >> 32))], 1u);
    __syncthreads();
    const uint32_t per = kBuckets / NT;
    uint32_t local = 0, big = 0;
    for (uint32_t u = 0; u < per; ++u) { const uint32_t c = cnt[tid * per + u]; local += c; big = max(big, c); }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan<NT>(local, scratch, &tot);
    for (uint32_t u = 0; u < per; ++u) { const uint32_t c = cnt[tid * per + u]; cur[tid * per + u] = ex; cnt[tid * per + u] = ex; ex += c; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) big = max(big, (uint32_t)__shfl_xor((int)big, o));
    if (lane == 0) s_red[2][wave] = big;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NWV; ++q) big = max(big, s_red[2][q]);
#pragma unroll
    for (int u = 0; u < KPT; ++u)
        if ((uint32_t)tid + (uint32_t)u * NT < n) keys[atomicAdd(&cur[bucket((uint32_t)(kreg[u] >> 32))], 1u)] = kreg[u];
    __syncthreads();
    if (big <= (uint32_t)kBucketLimit) {
        for (uint32_t i = tid; i < n; i += NT) {
            const uint64_t k = keys[i];
            const uint32_t b = bucket((uint32_t)(k >> 32));
            const uint32_t s0 = cnt[b], e0 = cur[b];
            uint32_t below = 0;
            for (uint32_t j = s0; j < e0; ++j) below += keys[j] < k ? 1u : 0u;
            emit(s0 + below, k);
        }
        return;
    }
    {
        uint32_t m = 8 * NT;
        while (m < n) m <<= 1;                                 // <= kRangeSortCap
        for (uint32_t i = n + tid; i < m; i += NT) keys[i] = ~0ull;
        __syncthreads();
        bitonic_sort_lds<NT>(keys, m, tid);
    }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += NT) emit(i, keys[i]);
}

// ---- binning, dense form ("scan") -------------------------------------------------------------------------------------
// When Gaussians are large (random-init weights: ~50 tiles each, a tile is touched by a fifth of all Gaussians) listing the
// instances first (emit) and sorting every tile's list moves each instance three times.  Here a tile filters the
// depth-ORDERED Gaussians directly: rank_rects_kernel leaves each Gaussian's tile rectangle, packed in 32 bits, at its depth
// rank; the blend kernel (scan_more below) walks the ranks 64 at a time -- test, ballot, popcount -- and takes the hits in
// order: the list the sort produces, with no atomics and no instance list.  At most T x P tests per view (a tile that never
// saturates tests every rank), so the host rules the form out beyond T x P = 2^31 (512^2: 2^30).
__global__ __launch_bounds__(256) void rank_rects_kernel(FwdParams p) {
    if (p.im.totals[1] != 0 || !binning_is_scan(p)) return;
    const int v = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.P) return;
    const size_t gi = (size_t)v * p.P + idx;
    uint32_t packed = 0;                                   // empty rectangle: never hit
    if (p.radii[gi] > 0) {
        int x0, y0, x1, y1;
        const float2 m = p.g.means2D[gi];
        tile_rect(m.x, m.y, p.radii[gi], p.gx, p.gy, &x0, &y0, &x1, &y1);
        packed = (uint32_t)x0 | ((uint32_t)y0 << 8) | ((uint32_t)x1 << 16) | ((uint32_t)y1 << 24);     // gx, gy <= 255 (host)
    }
    p.g.vals[1][(size_t)v * p.P + p.g.rank_of[gi]] = packed;   // the sort's spare value buffer
}

// ------------------------------------------------------------------------------------------------
// Scan form: the tile's list is produced ON DEMAND, inside the blend kernel.  A tile filters the depth-ORDERED Gaussians
// (rank_rects_kernel left every Gaussian's tile rectangle at its depth rank) a window of ranks at a time -- 64 ranks per test:
// two compares, two ballots, popcount -- and appends the hits to a ring in LDS (the blend's staging reads from there) and to
// the tile's segment of point_list (what the backward replays).  It asks for more only when the ring runs dry, and a tile
// whose 256 pixels are all saturated never asks again: in the regime this form is picked for (dense scenes: random-init
// weights put 50,000 Gaussians on a tile and its pixels saturate after 800-2,200 of them) a tile tests a few percent of the
// ranks, where a separate pass that builds complete lists tested all T x P pairs (0.38 ms of a 1.0 ms call at 256^2, 4
// views).  The list in memory is therefore a PREFIX of the reference's list -- as long as the forward walked, which is all the
// backward reads (tile_cursor holds its length).  Windows are sized from the tile's density for ~768 hits; a wave owns a
// contiguous quarter of the window (count pass, prefix over the waves, emit pass that re-tests: no masks kept).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kRing = 2048;                          // entries; a window adds at most kRing - 255

struct TileScan {                                         // wave-uniform, every thread keeps a copy
    uint32_t next_rank;                                   // first depth rank not tested yet
    uint32_t found;                                       // entries produced so far (= written to point_list)
    uint32_t head, waiting;                               // ring: next entry to consume, entries waiting
};

template <bool EMIT>
__device__ __forceinline__ uint32_t scan_groups(const uint32_t* rects, const uint32_t* order, uint32_t rank_begin, int g0, int g1, uint32_t P,
                                                uint32_t tx, uint32_t ty, int lane, unsigned long long lanes_before, uint32_t off, uint32_t* ring,
                                                uint32_t ring_at, uint32_t* list, uint32_t list_at, uint32_t list_cap) {
    uint32_t cnt = 0;
    for (int g = g0; g < g1; g += 8) {
        const uint32_t rank0 = rank_begin + (uint32_t)g * 64u + (uint32_t)lane;
        uint32_t r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = (g + u < g1 && rank0 + 64u * u < P) ? rects[rank0 + 64u * u] : 0u;
        uint32_t slot[8], val[8], setm = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t x0 = r[u] & 255u, y0 = (r[u] >> 8) & 255u, x1 = (r[u] >> 16) & 255u, y1 = r[u] >> 24;
            const unsigned long long m = __ballot(tx - x0 < x1 - x0) & __ballot(ty - y0 < y1 - y0);   // unsigned: x0 <= tx < x1, y0 <= ty < y1
            if (EMIT) {
                const bool set = (m >> lane) & 1ull;
                slot[u] = off + cnt + (uint32_t)__popcll(m & lanes_before);
                val[u] = set ? order[rank0 + 64u * u] : 0u;
                setm |= (uint32_t)set << u;
            }
            cnt += (uint32_t)__popcll(m);
        }
        if (EMIT) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (((setm >> u) & 1u) && list_at + slot[u] < list_cap) {
                    ring[(ring_at + slot[u]) & (kRing - 1u)] = val[u];
                    list[list_at + slot[u]] = val[u];
                }
        }
    }
    return cnt;
}

// All 256 threads.  Tests at least one more window of ranks and appends its hits; returns with s advanced.
__device__ __forceinline__ void scan_more(const FwdParams& p, TileScan& s, uint32_t* ring, uint32_t* scratch, const uint32_t* rects,
                                          const uint32_t* order, uint32_t* list, uint32_t count, uint32_t tx, uint32_t ty, int lane, int wave,
                                          unsigned long long lanes_before) {
    const uint32_t P = (uint32_t)p.P;
    unsigned long long want = 768ull * P / (count ? count : 1u);
    uint32_t win = (uint32_t)(want > (1ull << 24) ? (1ull << 24) : want);
    win = max(1024u, (win + 255u) & ~255u);
    const uint32_t room = kRing - s.waiting;
    uint32_t total = 0, off = 0;
    int g0 = 0, g1 = 0;
    for (;;) {
        const int groups = (int)((min(win, P - s.next_rank) + 63u) / 64u);
        const int gpw = (groups + 3) / 4;
        g0 = wave * gpw; g1 = min(groups, g0 + gpw);
        const uint32_t cnt = scan_groups<false>(rects, order, s.next_rank, g0, g1, P, tx, ty, lane, lanes_before, 0u, nullptr, 0u, nullptr, 0u, 0u);
        if (lane == 0) scratch[wave] = cnt;
        __syncthreads();
        total = 0; off = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = scratch[w];
            if (w < wave) off += c;
            total += c;
        }
        __syncthreads();
        if (total <= room || win <= 256u) break;           // 256 ranks cannot overflow the ring (waiting <= 255 when called)
        win = max(256u, (win / 2u + 255u) & ~255u);
    }
    scan_groups<true>(rects, order, s.next_rank, g0, g1, P, tx, ty, lane, lanes_before, off, ring, s.head + s.waiting, list, s.found, count);
    __syncthreads();
    s.next_rank += min(win, P - s.next_rank);
    s.found += total;
    s.waiting += total;
}

// grid V*T (tile_order picks the tile), 256 threads = 4 wave64.  forward.cu:261-374.
//
// Lanes and pixels.  Wave g owns the 16 x 4 strip g of the tile; its four 16-lane rows own the strip's four 4 x 4 CELLS (lane
// l: cell column l >> 4, pixel (l & 3, (l >> 2) & 3) inside it).  The tile's list is the reference's -- every Gaussian whose
// ceil(3 sigma_max) square overlaps the tile -- but a pair contributes only inside the ellipse power >= ln(1 / (255 opacity)),
// and in the trained-like regime (~3,000 entries per tile, ellipses of a few pixels) a cell meets a fifth of its
// tile's list.  So a batch of 256 entries is staged in LDS; while staging, every thread works out which of the sixteen cells
// ITS entry can reach (cell_mask: the ellipse's bounding box, inflated -- conservative, so skipping changes no bit); ballots
// compact the batch into sixteen index lists, front to back; and the four rows of a wave walk their four lists in lockstep,
// each lane row reading its own entry.  Per (pixel, entry) the arithmetic is the reference's, in the reference's order
// (forward.cu:332-358); the body has no per-lane branches: one wave-uniform branch leaves when no lane passes the alpha
// cut-off, everything behind it is selects.
template <bool SCAN, bool FAST_EXP>
__global__ __launch_bounds__(256) void blend_forward_kernel(FwdParams p) {
    __shared__ float2 s_xy[256];
    __shared__ float4 s_co[256];
    __shared__ float4 s_rgbc[256];
    __shared__ uint4 s_cnt[16];                           // [cell] entries of the batch the cell keeps, per staging wave
    __shared__ uint8_t s_list[17][256];                   // [cell] their batch indices, front to back (+ one row: the walk reads a group ahead)
    __shared__ uint32_t s_walk[4];
    __shared__ uint2 s_stat[kRasterStats ? 4 : 1];
    __shared__ uint32_t s_ring[SCAN ? kRing : 1];         // scan form: list entries found, not yet staged
    __shared__ uint32_t s_scan[4];
    if (binning_is_scan(p) != SCAN) return;               // async mode launches both forms
    const uint32_t vt = p.im.tile_order[blockIdx.x];           // (view, tile) this workgroup works on: scan_tiles_kernel
    const int v = (int)(vt / (uint32_t)p.T), tile = (int)(vt % (uint32_t)p.T);
    const int bx = tile % p.gx, by = tile / p.gx;
    const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cell = wave * 4 + row;
    const int pxi = bx * kTile + 4 * row + (lane & 3), pyi = by * kTile + 4 * wave + ((lane >> 2) & 3);
    const bool inside = pxi < p.W && pyi < p.H;
    const float pfx = (float)pxi, pfy = (float)pyi;
    const float tx0 = (float)(bx * kTile), ty0 = (float)(by * kTile);
    const unsigned long long lanes_before = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const bool ok = p.im.totals[1] == 0;
    uint2 rg = p.im.ranges[vt];
    if (!ok) rg.y = rg.x;
    const int rounds = (int)((rg.y - rg.x + 255u) / 256u);
    const size_t vo = (size_t)v * p.P;
    const uint32_t* rects = p.g.vals[1] + vo;             // scan form: tile rectangle at depth rank (rank_rects_kernel)
    const uint32_t* order = p.g.vals[0] + vo;             //            depth rank -> Gaussian index
    TileScan ts{0u, 0u, 0u, 0u};
    bool done = !inside;
    float T = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    uint32_t last_contributor = 0;
    uint32_t st_entries = 0, st_trips = 0, st_batches = 0;      // tile_stats (measurement): per lane its cell's entries, per wave its loop trips
    for (int i = 0; i < rounds; ++i) {
        if (__syncthreads_count(done) == 256) break;
        if constexpr (kRasterStats) ++st_batches;
        const uint32_t pos = rg.x + (uint32_t)i * 256u + (uint32_t)tid;
        const uint32_t need = min(256u, rg.y - rg.x - (uint32_t)i * 256u);
        if (SCAN) {
            while (ts.waiting < need && ts.next_rank < (uint32_t)p.P)
                scan_more(p, ts, s_ring, s_scan, rects, order, p.bn.point_list + rg.x, rg.y - rg.x, (uint32_t)bx, (uint32_t)by, lane, wave, lanes_before);
        }
        unsigned m16 = 0u;
        if (pos < rg.y && (!SCAN || (uint32_t)tid < ts.waiting)) {
            const uint32_t id = SCAN ? s_ring[(ts.head + (uint32_t)tid) & (kRing - 1u)] : p.bn.point_list[pos];
            const BlendRecord* rec = p.g.blend + vo + id;      // one line per entry (raster_state.h)
            const float4 co = rec->co;
            const float4 rc = rec->rc;
            const float2 xy = rec->xy;
            s_xy[tid] = xy; s_co[tid] = co; s_rgbc[tid] = rc;
            m16 = cell_mask(xy, co, rc.w, tx0, ty0);
        } else {
            // a slot without an entry holds a finite record: the walk reads ahead of its lists (stale indices), and the product-default
            // arithmetic multiplies a masked-out lane's colour by a zero weight instead of selecting -- 0 * NaN from LDS left by an
            // earlier kernel would poison the pixel
            s_xy[tid] = make_float2(0.f, 0.f); s_co[tid] = make_float4(0.f, 0.f, 0.f, 0.f); s_rgbc[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        unsigned long long keeps[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            keeps[c] = __ballot((m16 >> c) & 1u);
            if (lane == 0) reinterpret_cast<uint32_t*>(&s_cnt[c])[wave] = (uint32_t)__popcll(keeps[c]);
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if ((m16 >> c) & 1u) {
                const uint4 cn = s_cnt[c];
                const uint32_t ahead = (wave > 0 ? cn.x : 0u) + (wave > 1 ? cn.y : 0u) + (wave > 2 ? cn.z : 0u);
                s_list[c][ahead + (uint32_t)__popcll(keeps[c] & lanes_before)] = (uint8_t)tid;
            }
        }
        __syncthreads();
        if (SCAN) { const uint32_t took = min(need, ts.waiting); ts.head += took; ts.waiting -= took; }
        const uint32_t base = (uint32_t)i * 256u;
        const unsigned long long alive = __ballot(!done);
        if (alive != 0ull) {                                    // a wave whose 64 pixels are all finished only keeps the barriers
            const uint4 cn = s_cnt[cell];
            const uint32_t tot = (((alive >> (16 * row)) & 0xFFFFull) != 0ull) ? cn.x + cn.y + cn.z + cn.w : 0u;
            // a lane's OWN list length: 0 once its pixel is finished, so "still walking" is the index compare the step makes anyway
            // (a `done` flag tested per step is a VGPR 0/1 and costs three VALU per step to test and keep)
            uint32_t tot_l = done ? 0u : tot;
            // software pipeline: the cell's indices arrive four at a time (one 32-bit word, the next word a group ahead), the
            // entry itself one step ahead, in two register sets that take turns -- the loop is unrolled by the word, so there is
            // no copy between steps and every shift is a literal (the rotating form spent 13 of its ~45 VALU per step on moves)
            // (bytes behind `tot` are stale indices of earlier batches: any of them addresses a staged record, none is used)
            struct Entry { float2 xy; float4 co; float4 rc; };
            auto load = [&](uint32_t j) { return Entry{s_xy[j], s_co[j], s_rgbc[j]}; };
            auto step = [&](uint32_t k, uint32_t j, const Entry& e) {
                const float dx = e.xy.x - pfx, dy = e.xy.y - pfy;
                const float power = -0.5f * (e.co.x * dx * dx + e.co.z * dy * dy) - e.co.y * dx * dy;
                // (bitwise &: the operands are three compares; && would make the second and third a branch under a saved exec mask)
                const bool pass = (k < tot_l) & !(power > 0.0f) & !(power < e.rc.w);  // alpha < 1/255 guaranteed below the cut (preprocess_one)
                if (wave_ballot(pass) != 0ull) {
                    const float alpha = alpha_clamp(e.co.w * blend_exp<FAST_EXP>(pass ? power : 0.0f, e.co.w));
                    const float test_T = T * (1 - alpha);
                    const bool contributes = pass & !(alpha < 1.0f / 255.0f);
                    const bool finishes = contributes & (test_T < 0.0001f);
                    const bool blends = contributes & !finishes;
                    tot_l = finishes ? 0u : tot_l;
                    if constexpr (FAST_EXP) {
                        // product default: one weight, three fused multiply-adds (5 VALU for 12; the exact mode keeps the
                        // reference's (c alpha) T products and separate adds, forward.cu:352-353, for bit-identity with the oracle)
                        const float w = blends ? alpha * T : 0.0f;
                        C0 = __builtin_fmaf(e.rc.x, w, C0); C1 = __builtin_fmaf(e.rc.y, w, C1); C2 = __builtin_fmaf(e.rc.z, w, C2);
                    } else {
                        const float c0 = C0 + e.rc.x * alpha * T, c1 = C1 + e.rc.y * alpha * T, c2 = C2 + e.rc.z * alpha * T;
                        C0 = blends ? c0 : C0;
                        C1 = blends ? c1 : C1;
                        C2 = blends ? c2 : C2;
                    }
                    T = blends ? test_T : T;
                    last_contributor = blends ? base + j + 1u : last_contributor;
                }
            };
            const uint32_t* lst = reinterpret_cast<const uint32_t*>(s_list[cell]);
            uint32_t word = lst[0];
            Entry ea = load(word & 255u), eb;
            uint32_t k = 0;
            for (; wave_ballot(k < tot_l) != 0ull; k += 4) {
                const uint32_t word_next = lst[(k >> 2) + 1u];
                eb = load((word >> 8) & 255u);  step(k, word & 255u, ea);
                ea = load((word >> 16) & 255u); step(k + 1u, (word >> 8) & 255u, eb);
                eb = load(word >> 24);          step(k + 2u, (word >> 16) & 255u, ea);
                ea = load(word_next & 255u);    step(k + 3u, word >> 24, eb);
                word = word_next;
            }
            done = done | (tot_l != tot);                       // finished in this batch (tot >= 1 then), or before it
            if constexpr (kRasterStats) { st_entries += min(tot, k); st_trips += k; }    // the walk leaves a batch once the wave's 64 pixels are finished
        }
    }
    if (inside) {
        const size_t pid = (size_t)p.W * pyi + pxi;
        const size_t HW = (size_t)p.H * p.W;
        p.im.final_T[(size_t)v * HW + pid] = T;
        p.im.n_contrib[(size_t)v * HW + pid] = last_contributor;
        float* out = p.out_color + (size_t)v * 3 * HW;
        // a call that failed on the device (async mode: more instances than the binning buffer holds; prefiltered + a culled point)
        // has no host to report to before its output is consumed: the image is NaN, not a plausible background
        const float bad = __uint_as_float(0x7FC00000u);
        out[pid] = ok ? C0 + T * p.bg[0] : bad;
        out[HW + pid] = ok ? C1 + T * p.bg[1] : bad;
        out[2 * HW + pid] = ok ? C2 + T * p.bg[2] : bad;
    }
    // how far into its list the tile got: what the backward replays, and what it ranks its launch order by
    uint32_t walked = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) walked = max(walked, (uint32_t)__shfl_xor((int)walked, o));
    if constexpr (kRasterStats) {
        uint32_t ent = (lane & 15) == 0 ? st_entries : 0u;             // one lane per 16-lane row: the row's cell
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ent += (uint32_t)__shfl_xor((int)ent, o);
        if (lane == 0) s_stat[wave] = make_uint2(ent, st_trips);
    }
    if (lane == 0) s_walk[wave] = walked;
    __syncthreads();
    if (tid == 0) {
        p.im.tile_work[vt] = max(max(s_walk[0], s_walk[1]), max(s_walk[2], s_walk[3]));
        p.im.tile_cursor[vt] = SCAN ? ts.found : rg.y - rg.x;          // entries of the tile's list that exist in point_list
        p.im.tile_scanned[vt] = SCAN ? ts.next_rank : 0u;              // depth ranks the tile tested (scan form)
        if constexpr (kRasterStats)
            p.im.tile_stats[vt] = make_uint4(s_stat[0].x + s_stat[1].x + s_stat[2].x + s_stat[3].x, s_stat[0].y + s_stat[1].y + s_stat[2].y + s_stat[3].y,
                                             SCAN ? ts.next_rank : 0u, st_batches);
    }
}

// tile_count .. totals (ImageState::carve lays them out back to back) start from zero: a kernel of the library's own, not
// hipMemsetAsync (a memset NODE of a captured graph stopped zeroing once the process had issued an eager hipMemsetAsync elsewhere;
// and this is the shorter launch)
__global__ __launch_bounds__(1024) void zero_words_kernel(uint32_t* dst, int n) {
    for (int i = threadIdx.x; i < n; i += 1024) dst[i] = 0u;
}

// dgs_raster_state_read("conic_opacity" | "rgb"): the parity tests' view of one field of the blend records
__global__ void record_field_kernel(const BlendRecord* rec, float4* dst, size_t n, int field) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = field ? rec[i].rc : rec[i].co;
}

__global__ void mark_visible_kernel(int P, const float* means, const float* vm, uint8_t* present) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    const float tz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
    present[i] = !(tz <= 0.2f);   // in_frustum, auxiliary.h:154
}

static int check(hipStream_t st, int debug) {
    hipError_t e = debug ? hipStreamSynchronize(st) : hipSuccess;
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { fprintf(stderr, "[dgs] rasterizer: HIP error %d (%s)\n", (int)e, hipGetErrorString(e)); return DGS_ERR_DEVICE; }
    return DGS_OK;
}

static int pick_window_words(int P) {
    int words = (P + 31) / 32;
    words = ((words + 255) / 256) * 256;
    const int kMax = 16128;   // 63 KiB of LDS per tile workgroup (static scratch shares the 64 KiB default limit)
    return words < kMax ? words : kMax;
}

}  // namespace dgs

using namespace dgs;

extern "C" {

int dgs_abi_version(void) { return DGS_ABI_VERSION; }

const char* dgs_status_string(int s) {
    switch (s) {
        case DGS_OK: return "ok";
        case DGS_ERR_INVALID_ARGUMENT: return "invalid argument (means3D must have dimensions (num_points, 3); sizes must be positive)";
        case DGS_ERR_NEED_COLORS: return "provide exactly one of SHs or precomputed colors";
        case DGS_ERR_NEED_COVARIANCE: return "provide exactly one of scale/rotation pair or precomputed 3D covariance";
        case DGS_ERR_ALLOC: return "state-buffer allocation failed";
        case DGS_ERR_DEVICE: return "HIP device error";
        case DGS_ERR_PREFILTERED_CULLED: return "point is filtered although prefiltered is set";
        case DGS_ERR_BINNING_OVERFLOW: return "num_rendered exceeds binning capacity";
        default: return "unknown status";
    }
}

size_t dgs_raster_geom_bytes(int32_t P, int32_t V) { size_t b; GeomState::carve(nullptr, (size_t)P, (size_t)V, &b); return b; }
size_t dgs_raster_image_bytes(int32_t W, int32_t H, int32_t V) { size_t b; ImageState::carve(nullptr, (size_t)W, (size_t)H, (size_t)V, &b); return b; }
size_t dgs_raster_binning_bytes(int64_t N) { size_t b; BinningState::carve(nullptr, (size_t)(N < 1 ? 1 : N), &b); return b; }

int dgs_raster_forward(DgsRasterForwardArgs* a, dgs_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!a || a->P < 0 || a->width <= 0 || a->height <= 0 || a->V < 1 || a->views_per_set < 1) return DGS_ERR_INVALID_ARGUMENT;
    const int P = a->P, V = a->V, W = a->width, H = a->height;
    a->num_rendered = 0;
    (void)hipGetLastError();       // hipGetLastError is sticky per thread: an unrelated earlier failure (another library's probing) must
                                   // not be reported as ours by the checks below
    if (!a->out_color || !a->geom_alloc || !a->img_alloc || !a->binning_alloc) return DGS_ERR_INVALID_ARGUMENT;
    const size_t HW = (size_t)W * H;
    if (P == 0) {   // rasterize_points.cu:68: outputs stay zero-initialised
        hipMemsetAsync(a->out_color, 0, (size_t)V * 3 * HW * sizeof(float), st);
        return check(st, a->debug);
    }
    if (!a->means3D || !a->opacities || !a->viewmatrix || !a->projmatrix || !a->campos || !a->background || !a->radii) return DGS_ERR_INVALID_ARGUMENT;
    if ((a->shs == nullptr) == (a->colors_precomp == nullptr)) return DGS_ERR_NEED_COLORS;
    if (a->cov3D_precomp ? (a->scales || a->rotations) : !(a->scales && a->rotations)) return DGS_ERR_NEED_COVARIANCE;
    if (a->shs && (a->M < 1 || a->D < 0 || (a->D + 1) * (a->D + 1) > a->M || a->D > 3)) return DGS_ERR_INVALID_ARGUMENT;

    FwdParams p{};
    p.P = P; p.D = a->D; p.M = a->M; p.W = W; p.H = H; p.V = V; p.vps = a->views_per_set;
    p.gx = (W + kTile - 1) / kTile; p.gy = (H + kTile - 1) / kTile; p.T = p.gx * p.gy;
    p.bg = a->background; p.means3D = a->means3D; p.shs = a->shs; p.colors_pre = a->colors_precomp; p.opac = a->opacities;
    p.scales = a->scales; p.rots = a->rotations; p.cov_pre = a->cov3D_precomp; p.viewm = a->viewmatrix; p.projm = a->projmatrix;
    p.campos = a->campos; p.tanfov = a->tanfov; p.tanfovx = a->tanfovx; p.tanfovy = a->tanfovy; p.scale_mod = a->scale_modifier;
    p.prefiltered = a->prefiltered; p.raw_act = a->raw_activations; p.radii = a->radii; p.out_color = a->out_color;
    p.exact_exp = a->exact_exp ? 1 : 0;
    p.debug = a->debug ? 1 : 0;

    size_t gbytes, ibytes;
    GeomState::carve(nullptr, (size_t)P, (size_t)V, &gbytes);
    ImageState::carve(nullptr, (size_t)W, (size_t)H, (size_t)V, &ibytes);
    void* gbuf = a->geom_alloc(gbytes, a->geom_user);
    void* ibuf = a->img_alloc(ibytes, a->img_user);
    if (!gbuf || !ibuf) return DGS_ERR_ALLOC;
    p.g = GeomState::carve(gbuf, (size_t)P, (size_t)V, nullptr);
    p.im = ImageState::carve(ibuf, (size_t)W, (size_t)H, (size_t)V, nullptr);
    const bool async = a->binning_capacity > 0;
    void* bbuf = nullptr;
    if (async) {
        bbuf = a->binning_alloc(dgs_raster_binning_bytes(a->binning_capacity), a->binning_user);
        if (!bbuf) return DGS_ERR_ALLOC;
        p.bn = BinningState::carve(bbuf, (size_t)a->binning_capacity, nullptr);
    }

    const int VT = V * p.T;
    constexpr int kBitonicMax = 16384;                             // 128 KiB of LDS
    // once per process, before anything is enqueued (a stream capture must not meet it): the per-tile LDS sort's dynamic LDS size
    static const bool lds_ok =
        hipFuncSetAttribute(reinterpret_cast<const void*>(tile_bitonic_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kBitonicMax * 8 + 2 * kBuckets * 4) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(tile_bitonic_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kBitonicMax * 8 + 2 * kBuckets * 4) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(tile_bitonic_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kBitonicMax * 8 + 2 * kBuckets * 4) == hipSuccess &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(range_sort_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            kRangeSortCap * 8 + 2 * kBuckets * 4) == hipSuccess;
    if (!lds_ok) { fprintf(stderr, "[dgs] rasterizer: hipFuncSetAttribute(tile_bitonic_kernel, %d bytes of LDS) failed\n", kBitonicMax * 8 + 2 * kBuckets * 4); return DGS_ERR_DEVICE; }
    // tile_count, totals and the radix sort's arrival counters are neighbours in the image state (ImageState::carve): one fill
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(1024), 0, st, p.im.tile_count, (int)p.im.zero_span((size_t)V));
    const dim3 gridP((P + 255) / 256, V);
    const bool lds_tiles = p.T <= 4096;
    if (lds_tiles) hipLaunchKernelGGL((preprocess_kernel<true>), gridP, dim3(256), (size_t)p.T * 4, st, p);
    else hipLaunchKernelGGL((preprocess_kernel<false>), gridP, dim3(256), 0, st, p);
    int rc = check(st, a->debug);
    if (rc) return rc;

    // binning form (binning_form_of): the host only rules forms out; the choice itself is made from the instance statistics --
    // on the host in the sync mode (it has just read them back), on the device in the async mode
    const bool can_scan = scan_form_possible(p.gx, p.gy, p.T, P);
    p.bin_mode = a->binning_form;
    if (p.bin_mode < 0 || p.bin_mode > 3 || (p.bin_mode == 0 && !can_scan) || (p.bin_mode == kFormScan && !can_scan)) p.bin_mode = can_scan ? 0 : kFormBitonic;
    // async + the LDS sort named + the longest list the caller expects: that kernel alone (no radix sort, no rank-sort fallback in
    // the launch sequence); its LDS is sized for 1.5 x the expectation, and a list that outgrows it is sorted in LDS-sized chunks merged
    // by rank (tile_bitonic_kernel's long-list path: slower, never wrong)
    int bitonic_only_cap = 0;
    if (async && p.bin_mode == kFormBitonic && a->longest_hint > 0) {
        bitonic_only_cap = 2048;
        while (bitonic_only_cap < a->longest_hint + a->longest_hint / 2 && bitonic_only_cap < kBitonicMax) bitonic_only_cap <<= 1;
    }
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, st, p.im.tile_count, p.im.ranges, p.im.tile_cursor, VT, p.im.totals,
                       async ? (long long)a->binning_capacity : -1LL, p.im.tile_order, p.gx, p.gy, a->num_rendered_dev, a->num_rendered_host);
    rc = check(st, a->debug);
    if (rc) return rc;

    int forms = 0;                                                 // bit f set: form f has to be launched
    // The forms that work on depth ranks need the radix sort of the P depth keys (12 launches, ~0.17 ms at 256^2 x 4 views).  In
    // the sync mode it is worth having it in flight while the host waits for the statistics -- but only if it will be needed:
    // the previous call's form is the (performance-only) guess.
    static thread_local int last_form = 0;
    bool radix_done = false;
    auto radix_sort = [&]() {
        const int NB = sort_blocks(P);
        // default: the depth RANGE sort (four plain kernels, range_count_kernel).  DGS_RASTER_SORT=radix: the four-pass radix sort below,
        // three plain kernels per pass (cross-check: both give the same order bit for bit; also what P > 2 M takes)
        static const bool use_radix = getenv("DGS_RASTER_SORT") && !strcmp(getenv("DGS_RASTER_SORT"), "radix");
        const int RB = range_blocks(P);
        if (!use_radix && RB <= 1024) {
            const int stride = (int)range_ws_stride(P);
            hipLaunchKernelGGL(range_count_kernel, dim3(RB, V), dim3(256), 0, st, p.g.keys[0], p.im.depth_range, p.g.range_ws, P, V, stride);
            const int NWIN = range_windows(P), seg_off = 2 * kRangeBuckets + 1 + RB;
            hipLaunchKernelGGL(range_plan_kernel, dim3(V), dim3(1024), 0, st, p.g.range_ws, RB, NWIN, stride);
            hipLaunchKernelGGL(range_scatter_kernel, dim3(RB, V), dim3(256), 0, st, p.g.keys[0], p.im.depth_range, p.g.range_ws, p.g.keys[1], p.g.vals[1],
                               p.g.vals[0], p.g.rank_of, P, V, stride);
            hipLaunchKernelGGL(range_sort_kernel<512>, dim3(NWIN, V), dim3(512), kRangeSortCap * 8 + 2 * kBuckets * 4, st, p.g.range_ws, p.g.keys[1], p.g.vals[1],
                               p.g.vals[0], p.g.rank_of, P, stride, seg_off);
            radix_done = true;
            return;
        }
        for (int pass = 0; pass < 4; ++pass) {
            const int in = pass & 1, out = in ^ 1;
            hipLaunchKernelGGL(radix_hist_kernel, dim3(NB, V), dim3(256), 0, st, p.g.keys[in], p.g.radix_hist, P, NB, 8 * pass);
            hipLaunchKernelGGL(radix_colscan_kernel, dim3(1, V), dim3(256), 0, st, p.g.radix_hist, p.g.radix_base, NB);
            hipLaunchKernelGGL(radix_scatter_kernel, dim3(NB, V), dim3(256), 0, st, p.g.keys[in], pass == 0 ? (const uint32_t*)nullptr : p.g.vals[in],
                               p.g.keys[out], p.g.vals[out], pass == 3 ? p.g.rank_of : (uint32_t*)nullptr, p.g.radix_hist, p.g.radix_base, P, NB, 8 * pass);
        }
        radix_done = true;
    };
    if (!async) {   // the reference's blocking read of num_rendered (rasterizer_impl.cu:281), plus the longest tile list
        if (last_form != kFormBitonic && p.bin_mode != kFormBitonic) radix_sort();
        int32_t tot[4] = {0, 0, 0, 0};
        if (hipMemcpyAsync(tot, p.im.totals, sizeof(tot), hipMemcpyDeviceToHost, st) != hipSuccess) return DGS_ERR_DEVICE;
        if (hipStreamSynchronize(st) != hipSuccess) return DGS_ERR_DEVICE;
        if (tot[1] != 0) return tot[1];
        a->num_rendered = (int64_t)(uint32_t)tot[0];
        a->longest_list = (int64_t)(uint32_t)tot[2];
        bbuf = a->binning_alloc(dgs_raster_binning_bytes(a->num_rendered), a->binning_user);
        if (!bbuf) return DGS_ERR_ALLOC;
        p.bn = BinningState::carve(bbuf, (size_t)(a->num_rendered < 1 ? 1 : a->num_rendered), nullptr);
        int cap = 2048;
        while (cap < tot[2] && cap < kBitonicMax) cap <<= 1;
        p.bitonic_cap = cap;
        last_form = binning_form_of(p.bin_mode, p.bitonic_cap, a->num_rendered, (uint32_t)tot[2], p.T, P, V);
        forms = 1 << last_form;
    } else {
        // nothing is read back: the statistics go to the caller's device words (it reads them when it likes -- dgs_amd/raster.py
        // copies them to pinned host memory behind the call and looks at them before the NEXT call), every kernel of a form the
        // device does not pick returns at once.  A forced form (binning_form != 0: the caller's knowledge of the previous call of
        // this shape) launches that form only -- scan and rank sort are always valid, the LDS sort falls back to the rank sort on
        // the device when a list does not fit (both are launched).  All forms produce the same lists bit for bit.
        a->num_rendered = -1;
        a->longest_list = -1;
        // LDS the per-tile sort is launched with: for the longest list the caller expects (+ 50 %) -- then it is launched alone and
        // takes a longer list through its chunked path --, else the maximum with the rank sort beside it as the device's fallback
        p.bitonic_cap = bitonic_only_cap ? bitonic_only_cap : kBitonicMax;
        p.bitonic_any = bitonic_only_cap ? 1 : 0;
        forms = p.bin_mode ? (1 << p.bin_mode) | (p.bin_mode == kFormBitonic && !bitonic_only_cap ? 1 << kFormRankSort : 0)
                           : (1 << kFormRankSort) | (1 << kFormScan) | (1 << kFormBitonic);
    }

    // tools' library: DGS_RASTER_FWD_LDS_PAD = bytes of unused LDS per blend workgroup (fewer of them per CU: is the walk bound by
    // throughput or by the number of waves?  raster_common.h kRasterAblate)
    size_t blend_pad = 0;
    if constexpr (kRasterAblate) { static const int pad = [] { const char* e = getenv("DGS_RASTER_FWD_LDS_PAD"); return e ? atoi(e) : 0; }(); blend_pad = (size_t)pad; }
    if ((forms & ((1 << kFormRankSort) | (1 << kFormScan))) && !radix_done) radix_sort();
    if (forms & (1 << kFormScan)) {
        hipLaunchKernelGGL(rank_rects_kernel, gridP, dim3(256), 0, st, p);
        if (p.exact_exp) hipLaunchKernelGGL((blend_forward_kernel<true, false>), dim3(VT), dim3(256), blend_pad, st, p);
        else hipLaunchKernelGGL((blend_forward_kernel<true, true>), dim3(VT), dim3(256), blend_pad, st, p);
    }
    if (forms & ((1 << kFormRankSort) | (1 << kFormBitonic))) {
        if (lds_tiles) hipLaunchKernelGGL((emit_instances_kernel<true>), gridP, dim3(256), (size_t)p.T * 8, st, p);
        else hipLaunchKernelGGL((emit_instances_kernel<false>), gridP, dim3(256), 0, st, p);
    }
    if (forms & (1 << kFormRankSort)) {
        const int wwords = pick_window_words(P);
        hipLaunchKernelGGL(tile_sort_kernel, dim3(p.T, V), dim3(256), (size_t)wwords * 4, st, p, wwords);
    }
    if (forms & (1 << kFormBitonic)) {
        const size_t lds = (size_t)p.bitonic_cap * 8 + 2 * kBuckets * 4;
        // 4,096 .. 8,192-entry lists (the trained-like 256^2 regime): 512 threads -- the same two workgroups per CU (80 KiB of LDS each)
        // with twice the waves to hide the kernel's chain of LDS phases behind (its network's smallest size is 8 x 512 = 4,096 keys,
        // so shorter capacities keep 256 threads); DGS_RASTER_BITONIC_NT=256: measurement aid (profiles/r06_tile_sort_threads_ab.txt)
        static const int nt_env = getenv("DGS_RASTER_BITONIC_NT") ? atoi(getenv("DGS_RASTER_BITONIC_NT")) : 0;
        if (p.bitonic_cap > 8192) hipLaunchKernelGGL(tile_bitonic_kernel<1024>, dim3(VT), dim3(1024), lds, st, p);
        else if (p.bitonic_cap >= 4096 && nt_env != 256) hipLaunchKernelGGL(tile_bitonic_kernel<512>, dim3(VT), dim3(512), lds, st, p);
        else hipLaunchKernelGGL(tile_bitonic_kernel<256>, dim3(VT), dim3(256), lds, st, p);
    }
    if (forms & ((1 << kFormRankSort) | (1 << kFormBitonic))) {
        if (p.exact_exp) hipLaunchKernelGGL((blend_forward_kernel<false, false>), dim3(VT), dim3(256), blend_pad, st, p);
        else hipLaunchKernelGGL((blend_forward_kernel<false, true>), dim3(VT), dim3(256), blend_pad, st, p);
    }
    return check(st, a->debug);
}

int dgs_raster_binning_form(int32_t binning_form, int64_t num_rendered, int64_t longest_list, int32_t P, int32_t W, int32_t H, int32_t V) {
    if (P < 0 || W <= 0 || H <= 0 || V < 1 || num_rendered < 0 || longest_list < 0) return DGS_ERR_INVALID_ARGUMENT;
    const int gx = (W + kTile - 1) / kTile, gy = (H + kTile - 1) / kTile;
    const long long T = (long long)gx * gy;
    const bool can_scan = scan_form_possible(gx, gy, T, P);
    int mode = binning_form;
    if (mode < 0 || mode > 3 || (mode == 0 && !can_scan) || (mode == kFormScan && !can_scan)) mode = can_scan ? 0 : kFormBitonic;
    return binning_form_of(mode, 16384, num_rendered, longest_list, T, P, V);
}

int dgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present,
                     dgs_stream_t stream) {
    (void)projmatrix;
    if (P < 0) return DGS_ERR_INVALID_ARGUMENT;
    if (P == 0) return DGS_OK;
    if (!means3D || !viewmatrix || !present) return DGS_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mark_visible_kernel, dim3((P + 255) / 256), dim3(256), 0, st, P, means3D, viewmatrix, present);
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

int64_t dgs_raster_state_read(const char* name, int32_t P, int32_t W, int32_t H, int32_t V, int64_t N, const void* gbuf,
                              const void* bbuf, const void* ibuf, void* dst, int64_t dst_bytes, dgs_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const GeomState g = GeomState::carve(const_cast<void*>(gbuf), (size_t)P, (size_t)V, nullptr);
    const ImageState im = ImageState::carve(const_cast<void*>(ibuf), (size_t)W, (size_t)H, (size_t)V, nullptr);
    const BinningState bn = BinningState::carve(const_cast<void*>(bbuf), (size_t)(N < 1 ? 1 : N), nullptr);
    const size_t n = (size_t)P * V, HW = (size_t)W * H * V;
    const size_t T = (size_t)((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile) * V;
    const void* src = nullptr;
    size_t bytes = 0;
    const auto is = [&](const char* s) { return strcmp(name, s) == 0; };
    if (is("depths")) { src = g.depths; bytes = n * 4; }
    else if (is("means2D")) { src = g.means2D; bytes = n * 8; }
    else if (is("conic_opacity") || is("rgb")) {
        bytes = n * 16;
        if ((int64_t)bytes > dst_bytes) return DGS_ERR_INVALID_ARGUMENT;
        if (n) hipLaunchKernelGGL(record_field_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g.blend, static_cast<float4*>(dst), n, is("rgb") ? 1 : 0);
        return hipGetLastError() == hipSuccess ? (int64_t)bytes : (int64_t)DGS_ERR_DEVICE;
    }
    else if (is("tiles_touched")) { src = g.tiles_touched; bytes = n * 4; }
    else if (is("clamped")) { src = g.clamped; bytes = n; }
    else if (is("cov3D")) { src = g.cov3D; bytes = n * 24; }
    else if (is("rank_of")) { src = g.rank_of; bytes = n * 4; }
    else if (is("order")) { src = g.vals[0]; bytes = n * 4; }
    else if (is("ranges")) { src = im.ranges; bytes = T * 8; }
    else if (is("n_contrib")) { src = im.n_contrib; bytes = HW * 4; }
    else if (is("final_T")) { src = im.final_T; bytes = HW * 4; }
    else if (is("list_len")) { src = im.tile_cursor; bytes = T * 4; }       // entries of each tile's list present in point_list
    else if (is("tile_work")) { src = im.tile_work; bytes = T * 4; }
    else if (is("tile_scanned")) { src = im.tile_scanned; bytes = T * 4; }
    else if (is("tile_stats")) { src = im.tile_stats; bytes = T * 16; }                // forward blend: see raster_state.h
    else if (is("tile_stats_bwd")) { src = im.tile_stats + T; bytes = T * 16; }
    else if (is("point_list")) { src = bn.point_list; bytes = (size_t)(N < 0 ? 0 : N) * 4; }
    else return DGS_ERR_INVALID_ARGUMENT;
    if ((int64_t)bytes > dst_bytes) return DGS_ERR_INVALID_ARGUMENT;
    if (bytes && hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return DGS_ERR_DEVICE;
    return (int64_t)bytes;
}

}  // extern "C"
