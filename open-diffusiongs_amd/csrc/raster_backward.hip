// raster_backward.hip -- backward pass of the MI355X-native 3D-Gaussian rasterizer (gfx950, wave64).
//
// What it computes is fixed by the reference (cuda_rasterizer/backward.cu, rasterizer_impl.cu:340-434); HOW is
// redesigned for CDNA4:
//
//   reference (CUDA)                                         this file
//   -------------------------------------------------------  ---------------------------------------------------------
//   renderCUDA backward  backward.cu:399-557                 blend_backward_kernel: same back-to-front replay per 16x16
//     9-10 global atomicAdd per contributing (pixel,           tile on the forward's 4 x 4 cells and per-cell lists; the nine
//     Gaussian) pair                                           per-Gaussian sums are reduced across a cell's 16 lanes (DPP
//                                                              adds), across the tile's cells in LDS, and leave as ONE atomic
//                                                              INSTRUCTION per (tile, Gaussian) into the Gaussian's 64-byte
//                                                              gradient record (lane z adds value z: one L2 request) instead
//                                                              of <= 256 x 9; the replay starts at the tile's deepest
//                                                              contributor.  blend_backward_pair_kernel: the same with two
//                                                              pixels per lane (packed fp32), for grids of >= 3,072 tiles
//                                                            DETERMINISTIC form (DgsRasterBackwardArgs.scratch, opt-in):
//                                                              no atomic at all.  A (tile, Gaussian) instance has a slot
//                                                              of its own, Gaussian-major -- slot = (exclusive scan of
//                                                              tiles_touched)[Gaussian] + index of the tile in the Gaussian's
//                                                              rectangle --, the tile STORES its nine sums there, and
//                                                              gather_partials_kernel adds a Gaussian's slots in rectangle
//                                                              order.  Which slots a tile wrote needs no flag and no fill: a
//                                                              tile's list is sorted by (depth bits, index), it replays the
//                                                              first `todo` entries, so it wrote Gaussian g iff key(g) <= the
//                                                              key of its deepest replayed entry (8 bytes per tile)
//   computeCov2DCUDA     backward.cu:144-274                 preprocess_backward_kernel: ONE thread per (set, Gaussian)
//   preprocessCUDA bwd   backward.cu:346-396 (+ SH :20-139,    walks the views of its set, so gradients of the views of a
//     computeCov3D :278-341)                                   set are summed in registers in a fixed order -- no atomics,
//   torch autograd of exp / normalize / sigmoid                deterministic -- and the activation Jacobians are applied in
//     (gs_core.py:330-334) when raw_activations = 1            the same pass
#include "raster_common.h"

namespace dgs {

struct BwdParams {
    int P, D, M, W, H, V, vps, gx, gy, T, raw_act, exact_exp, ablate;
    const float *bg, *means3D, *shs, *colors_pre, *opac, *scales, *rots, *cov_pre, *viewm, *projm, *campos, *tanfov;
    float tanfovx, tanfovy, scale_mod;
    const int* radii;
    const float* dL_dpix;
    GeomState g;
    ImageState im;
    BinningState bn;
    float *dL_dmean2D, *dL_dconic, *dL_dcolors, *dL_dcov3D, *dL_dopacity, *dL_dmeans3D, *dL_dsh, *dL_dscales, *dL_drots;
    // deterministic form (null: the atomic form): see BwdScratch
    uint32_t* slot_base;         // [V*P + 1]   exclusive scan of tiles_touched: first slot of (view, Gaussian)
    unsigned long long* last_key;// [V*T]       (depth bits << 32 | Gaussian) of the deepest entry the tile replayed; 0: none
    float4* slot_a;              // [slots]     {colour r, g, b, mean2D x}
    float4* slot_b;              // [slots]     {mean2D y, conic xx, conic xy, conic yy}
    float* slot_c;               // [slots]     opacity
};

// Scratch of the deterministic backward, carved from ONE caller-owned buffer (dgs_raster_backward_scratch_bytes).
struct BwdScratch {
    uint32_t* slot_base; uint32_t* block_sums; unsigned long long* last_key; float4* slot_a; float4* slot_b; float* slot_c;
    static BwdScratch carve(void* buf, size_t P, size_t V, size_t T, size_t slots, size_t* bytes) {
        Carver c(buf);
        BwdScratch s;
        s.slot_base = c.take<uint32_t>(P * V + 1);
        s.block_sums = c.take<uint32_t>((P * V + 4095) / 4096 + 1);
        s.last_key = c.take<unsigned long long>(V * T);
        s.slot_a = c.take<float4>(slots);
        s.slot_b = c.take<float4>(slots);
        s.slot_c = c.take<float>(slots);
        if (bytes) *bytes = c.bytes();
        return s;
    }
};

// ---- exclusive scan of tiles_touched over all (view, Gaussian): two launches, 4,096 elements per workgroup ----
__global__ __launch_bounds__(256) void touched_block_sums_kernel(const uint32_t* touched, size_t n, uint32_t* block_sums) {
    __shared__ uint32_t scratch[8];
    const size_t base = (size_t)blockIdx.x * 4096;
    uint32_t sum = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const size_t i = base + (size_t)r * 256 + threadIdx.x;
        sum += i < n ? touched[i] : 0u;
    }
    uint32_t total;
    (void)block_exclusive_scan<256>(sum, scratch, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void touched_scan_kernel(const uint32_t* touched, size_t n, const uint32_t* block_sums, uint32_t* slot_base) {
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t s_base;
    // the workgroups in front of this one
    uint32_t before = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) before += block_sums[b];
    uint32_t total;
    (void)block_exclusive_scan<256>(before, scratch, &total);
    if (threadIdx.x == 0) s_base = total;
    __syncthreads();
    // thread t owns 16 CONSECUTIVE elements (the scan is over the element order)
    const size_t first = (size_t)blockIdx.x * 4096 + (size_t)threadIdx.x * 16;
    uint32_t c[16], mine = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) { c[e] = first + e < n ? touched[first + e] : 0u; mine += c[e]; }
    uint32_t run = s_base + block_exclusive_scan<256>(mine, scratch, &total);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        if (first + e < n) slot_base[first + e] = run;
        run += c[e];
    }
    if (first <= n && n < first + 16) slot_base[n] = run;     // the end: first slot behind the last Gaussian (n % 16 elements of this thread counted)
}

// Row length of the per-batch accumulators in LDS, [value][entry]: 256 entries + 4, so that the flush below -- lane = (entry, value),
// sixteen lanes an entry -- reads bank (4 z + e) mod 32: two lanes a bank at worst (256 would put the nine values of an entry in one).
constexpr int kAccRow = 260;

// Flush of the atomic form: the nine sums of a (tile, Gaussian) instance go into the Gaussian's 64-byte gradient record with ONE atomic
// instruction, lane z of a 16-lane group adding value z.  The lanes of one instruction that fall into the same line travel to L2 as one
// request: measured (tools/ubench/atomic_layout_bench.hip, 2.7 M instances = the trained-like regime, 4 views at 256^2) 134 us against
// 1,178 us for nine instructions into four arrays with a lane per instance -- the form of rounds 1-4, whose cost inside the blend
// kernel was 0.19 ms of 0.88 (ablation, profiles/r04_raster_backward_ablation.txt).  A wave flushes the 64 entries per 128 it staged itself
// (it reads their ids), so no barrier separates the flush from the next batch's staging.  NPASS passes of four entries.
template <int NPASS>
__device__ __forceinline__ void flush_records(float (*acc)[kAccRow], const uint32_t* s_id, float* grad_acc, size_t vo, int first_entry, int lane) {
    const int z = lane & 15, sub = lane >> 4;
    if (z < 9) {
#pragma unroll 4
        for (int pass = 0; pass < NPASS; ++pass) {
            const int e = first_entry + 4 * pass + sub;
            const float val = acc[z][e];
            if (val != 0.f) {
                acc[z][e] = 0.f;
                atomicAdd(grad_acc + 16 * (vo + s_id[e]) + z, val);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();      // the wave's next staging overwrites the ids its other lanes read here (no instruction on the GPU)
}

// grid V*T (workgroup b takes tile tile_order[b]), 256 threads = 4 wave64.  backward.cu:399-557.
//
// Lanes, pixels and lists as in the forward (blend_forward_kernel): wave g owns strip g, its four 16-lane rows own the strip's
// four 4 x 4 cells; a batch of 256 list entries is staged in LDS back to front, compacted into one index list per cell
// (cell_mask: the cells the Gaussian's ellipse can reach), and the four rows of a wave walk their lists in lockstep.
//
// Per (tile, Gaussian) the nine sums have to be reduced over up to 256 pixels.  A 16-lane row reduces its cell with four DPP
// adds per value; lane 15 of the row adds the nine partial sums into the batch entry's accumulators in LDS (ds_add_f32: up
// to sixteen cells meet there); when the batch is done, thread j reads entry j's nine sums and issues ONE global atomic per
// value -- 256 Gaussians' atomics in flight at once, a sixteenth (at most) of the per-cell count.  The body has no per-lane
// branches: two wave-uniform exits (no lane inside the ellipse / above the alpha cut-off), selects behind them.
// Measured per 4 views at 256^2, trained-like / random-init regime, forward + backward: one atomic per wave per value 4.14 /
// 2.49 ms; 16 x 4 strip masks 3.45 / 2.11; + combining a tile's four strips in LDS 3.00 / 1.88; + launch order by work 2.33 /
// 1.70; cells: see DESIGN.md.  (One wave per tile with four pixels per lane lost: a latency chain.)
template <bool FAST_EXP, bool DET>
__global__ __launch_bounds__(256) void blend_backward_kernel(BwdParams p) {
    __shared__ uint32_t s_id[256];
    __shared__ uint2 s_stat[kRasterStats ? 4 : 1];
    __shared__ float2 s_xy[256];
    __shared__ float4 s_co[256];
    __shared__ float4 s_rgbc[256];                        // colour, alpha cut-off on `power`
    __shared__ uint32_t s_max[4];
    __shared__ uint4 s_cnt[16];                           // [cell] entries of the batch the cell keeps, per staging wave
    __shared__ uint8_t s_list[17][256];                   // [cell] their batch indices, back to front (+ one row: read-ahead)
    // [wave copy][value][entry] sums over the tile's pixels.  Atomic form: ONE copy, the sixteen cells' ds_add_f32 meet in it in whatever
    // order the four waves get there (fp32: the sum's last bit depends on that order).  Deterministic form: a copy per wave -- inside
    // a wave the adds happen in program order -- and the four copies are added in wave order when the batch is done.
    constexpr int NACC = DET ? 4 : 1;
    __shared__ float s_acc[NACC][9][kAccRow];
    const uint32_t vt = p.im.tile_order[blockIdx.x];           // order_tiles_kernel: most replayed entries first
    const int v = (int)(vt / (uint32_t)p.T), tile = (int)(vt % (uint32_t)p.T), s = v / p.vps;
    const int bx = tile % p.gx, by = tile / p.gx;
    const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cell = wave * 4 + row;
    const int pxi = bx * kTile + 4 * row + (lane & 3), pyi = by * kTile + 4 * wave + ((lane >> 2) & 3);
    const bool inside = pxi < p.W && pyi < p.H;
    const float pfx = (float)pxi, pfy = (float)pyi;
    const float tx0 = (float)(bx * kTile), ty0 = (float)(by * kTile);
    const unsigned long long lanes_before = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const size_t HW = (size_t)p.H * p.W, pid = (size_t)p.W * pyi + pxi;
    const uint2 rg = p.im.ranges[vt];
    const size_t vo = (size_t)v * p.P;

    const float T_final = inside ? p.im.final_T[(size_t)v * HW + pid] : 0.0f;
    float T = T_final;
    const uint32_t last_contributor = inside ? p.im.n_contrib[(size_t)v * HW + pid] : 0u;
    float dpix[3] = {0.f, 0.f, 0.f};
    if (inside) {
        const float* g = p.dL_dpix + (size_t)v * 3 * HW;
        dpix[0] = g[pid]; dpix[1] = g[HW + pid]; dpix[2] = g[2 * HW + pid];
    }
    float bg_dot = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) bg_dot += p.bg[c] * dpix[c];
    float accum_rec[3] = {0.f, 0.f, 0.f}, last_color[3] = {0.f, 0.f, 0.f}, last_alpha = 0.f;
    const float ddelx_dx = (float)(0.5 * p.W), ddely_dy = (float)(0.5 * p.H);

    // the deepest contributor of the tile: everything behind it in the list contributes to no pixel
    uint32_t mc = last_contributor;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mc = max(mc, (uint32_t)__shfl_xor((int)mc, o));
    if (lane == 0) s_max[wave] = mc;
#pragma unroll
    for (int w = 0; w < NACC; ++w)
#pragma unroll
        for (int k = 0; k < 9; ++k) s_acc[w][k][tid] = 0.f;
    __syncthreads();
    const uint32_t todo = max(max(s_max[0], s_max[1]), max(s_max[2], s_max[3]));   // <= rg.y - rg.x
    const int rounds = (int)((todo + 255u) / 256u);
    const bool colors_per_set = p.colors_pre != nullptr;
    uint32_t st_entries = 0, st_trips = 0;                      // tile_stats (measurement)
    for (int i = 0; i < rounds; ++i) {
        // batch entry e (0..255) = 1-based list index contributor = todo - (i * 256 + e), list position rg.x + contributor - 1
        const int idx = (int)todo - 1 - (i * 256 + tid);
        unsigned m16 = 0u;
        if (idx >= 0) {
            const uint32_t id = p.bn.point_list[rg.x + (uint32_t)idx];
            const BlendRecord* rec = p.g.blend + vo + id;      // one line per entry (raster_state.h)
            const float4 co = rec->co;
            float4 rc = rec->rc;
            const float2 xy = rec->xy;
            m16 = cell_mask(xy, co, rc.w, tx0, ty0);
            if (colors_per_set) {
                const float* c = p.colors_pre + 3 * ((size_t)s * p.P + id);
                rc.x = c[0]; rc.y = c[1]; rc.z = c[2];
            }
            s_id[tid] = id; s_xy[tid] = xy; s_co[tid] = co; s_rgbc[tid] = rc;
        } else {
            // finite records in the slots without an entry (the walk reads ahead of its lists), as in the forward
            s_id[tid] = 0u; s_xy[tid] = make_float2(0.f, 0.f); s_co[tid] = make_float4(0.f, 0.f, 0.f, 0.f); s_rgbc[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
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
        {
            const uint4 cn = s_cnt[cell];
            const uint32_t tot = cn.x + cn.y + cn.z + cn.w;
            const uint32_t first = todo - (uint32_t)(i * 256);          // contributor (1-based list index) of batch entry 0
            // software pipeline as in the forward: a cell's indices four at a time (one word, the next word a group ahead), the
            // entry one step ahead in two register sets that take turns; unrolled by the word: no copies, literal shifts
            struct Entry { float2 xy; float4 co; float4 rc; };
            auto load = [&](uint32_t j) { return Entry{s_xy[j], s_co[j], s_rgbc[j]}; };
            auto step = [&](uint32_t k, uint32_t j, const Entry& e) {
                // pixel took part iff index <= last_contributor (backward.cu:463-468); cheap rejects first (outside the ellipse,
                // or below the Gaussian's alpha cut-off: alpha < 1/255 guaranteed, see preprocess_one -- the test the forward
                // used to drop the pair)
                const float dx = e.xy.x - pfx, dy = e.xy.y - pfy;
                const float power = -0.5f * (e.co.x * dx * dx + e.co.z * dy * dy) - e.co.y * dx * dy;
                // (bitwise &: plain compares; && would branch under a saved exec mask)
                const bool near = (k < tot) & inside & (first - j <= last_contributor) & !(power > 0.0f) & !(power < e.rc.w);
                if (wave_ballot(near) != 0ull) {
                    const float G = blend_exp<FAST_EXP>(near ? power : 0.0f, e.co.w);
                    const float alpha = alpha_clamp(e.co.w * G);
                    const bool take = near & !(alpha < 1.0f / 255.0f);
                    const unsigned long long takers = wave_ballot(take);
                    if (takers != 0ull) {                               // wave-uniform: somebody in this strip touches its Gaussian
                        // product default: one v_rcp_f32 serves both divisions by (1 - alpha) (the exact mode keeps the reference's two
                        // correctly rounded divisions, backward.cu:478,506: ~10 VALU each)
                        const float inv1ma = FAST_EXP ? hw_rcp(1.f - alpha) : 0.f;
                        const float Tn = FAST_EXP ? T * inv1ma : T / (1.f - alpha);
                        const float dchannel_dcolor = alpha * Tn;
                        const float col[3] = {e.rc.x, e.rc.y, e.rc.z};
                        float c9[9];
                        float dL_dalpha = 0.0f;
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) {
                            const float rec = last_alpha * last_color[ch] + (1.f - last_alpha) * accum_rec[ch];
                            dL_dalpha += (col[ch] - rec) * dpix[ch];
                            c9[ch] = dchannel_dcolor * dpix[ch];
                            accum_rec[ch] = take ? rec : accum_rec[ch];
                            last_color[ch] = take ? col[ch] : last_color[ch];
                        }
                        dL_dalpha *= Tn;
                        dL_dalpha += (FAST_EXP ? -T_final * inv1ma : -T_final / (1.f - alpha)) * bg_dot;
                        T = take ? Tn : T;
                        last_alpha = take ? alpha : last_alpha;
                        const float dL_dG = e.co.w * dL_dalpha;
                        const float gdx = G * dx, gdy = G * dy;
                        const float dG_ddelx = -gdx * e.co.x - gdy * e.co.y;
                        const float dG_ddely = -gdy * e.co.z - gdx * e.co.y;
                        c9[3] = dL_dG * dG_ddelx * ddelx_dx;
                        c9[4] = dL_dG * dG_ddely * ddely_dy;
                        c9[5] = -0.5f * gdx * dx * dL_dG;
                        c9[6] = -0.5f * gdx * dy * dL_dG;
                        c9[7] = -0.5f * gdy * dy * dL_dG;
                        c9[8] = G * dL_dalpha;
                        const bool reduce = !(kRasterAblate && (p.ablate & 16)), add = !(kRasterAblate && (p.ablate & 2));
#pragma unroll
                        for (int q = 0; q < 9; ++q) c9[q] = reduce ? row_sum_to_lane15(take ? c9[q] : 0.f) : (take ? c9[q] : 0.f);
                        if (add && (lane & 15) == 15 && ((takers >> (16 * row)) & 0xFFFFull) != 0ull) {
#pragma unroll
                            for (int q = 0; q < 9; ++q) lds_add(&s_acc[DET ? wave : 0][q][j], c9[q]);
                        }
                    }
                }
            };
            const uint32_t* lst = reinterpret_cast<const uint32_t*>(s_list[cell]);
            uint32_t word = lst[0];
            Entry ea = load(word & 255u), eb;
            if constexpr (kRasterStats) st_entries += tot;
            uint32_t k = 0;
            for (; wave_ballot(k < tot) != 0ull && !(kRasterAblate && (p.ablate & 4)); k += 4) {
                const uint32_t word_next = lst[(k >> 2) + 1u];
                eb = load((word >> 8) & 255u);  step(k, word & 255u, ea);
                ea = load((word >> 16) & 255u); step(k + 1u, (word >> 8) & 255u, eb);
                eb = load(word >> 24);          step(k + 2u, (word >> 16) & 255u, ea);
                ea = load(word_next & 255u);    step(k + 3u, word >> 24, eb);
                word = word_next;
            }
            if constexpr (kRasterStats) st_trips += k;
        }
        __syncthreads();
        if constexpr (!DET) {
            if (!(kRasterAblate && (p.ablate & 1))) flush_records<16>(s_acc[0], s_id, p.g.grad_acc, vo, 64 * wave, lane);
        } else {
            // entry `tid`: the tile's sums, stored into the slot of (Gaussian, this tile)
            float c9[9];
            bool any = false;                                      // any of the four per-wave copies non-zero: from the COPIES, not from their
#pragma unroll                                                     // sum (partials that cancel exactly would otherwise stay behind for the next batch)
            for (int q = 0; q < 9; ++q) {
                const float a0 = s_acc[0][q][tid], a1 = s_acc[1][q][tid], a2 = s_acc[2][q][tid], a3 = s_acc[3][q][tid];
                c9[q] = ((a0 + a1) + a2) + a3;
                any = any || a0 != 0.f || a1 != 0.f || a2 != 0.f || a3 != 0.f;
            }
            // every replayed entry STORES its sums (zeros too): Gaussian-major, the tile's index inside the Gaussian's rectangle (the
            // forward's tile_rect on the same state: the same rectangle)
            if (idx >= 0) {
                if (any) {
#pragma unroll
                    for (int w = 0; w < NACC; ++w)
#pragma unroll
                        for (int q = 0; q < 9; ++q) s_acc[w][q][tid] = 0.f;
                }
                const uint32_t id = s_id[tid];
                const size_t gv = vo + id;
                int x0, y0, x1, y1;
                tile_rect(s_xy[tid].x, s_xy[tid].y, p.radii[gv], p.gx, p.gy, &x0, &y0, &x1, &y1);
                const size_t slot = (size_t)p.slot_base[gv] + (size_t)((by - y0) * (x1 - x0) + (bx - x0));
                p.slot_a[slot] = make_float4(c9[0], c9[1], c9[2], c9[3]);
                p.slot_b[slot] = make_float4(c9[4], c9[5], c9[6], c9[7]);
                p.slot_c[slot] = c9[8];
            }
        }
    }
    if constexpr (DET) {
        // what this tile replayed: the first `todo` entries of its list, which is sorted by (depth bits, Gaussian index)
        if (tid == 0) {
            unsigned long long key = 0ull;
            if (todo > 0) {
                const uint32_t id = p.bn.point_list[rg.x + todo - 1u];
                key = ((unsigned long long)__float_as_uint(p.g.depths[vo + id]) << 32) | id;
            }
            p.last_key[vt] = key;
        }
    }
    if constexpr (kRasterStats) {
        uint32_t ent = (lane & 15) == 0 ? st_entries : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ent += (uint32_t)__shfl_xor((int)ent, o);
        if (lane == 0) s_stat[wave] = make_uint2(ent, st_trips);
        __syncthreads();
        if (tid == 0)
            p.im.tile_stats[(size_t)p.V * p.T + vt] = make_uint4(s_stat[0].x + s_stat[1].x + s_stat[2].x + s_stat[3].x,
                                                                 s_stat[0].y + s_stat[1].y + s_stat[2].y + s_stat[3].y, 0u, (uint32_t)rounds);
    }
}

__device__ __forceinline__ v2f sel2(bool a, bool b, v2f t, v2f f) { return v2f{a ? t.x : f.x, b ? t.y : f.y}; }

// The same replay with TWO pixels per lane.  grid V*T (workgroup b takes tile tile_order[b]), 128 threads = 2 wave64.
//
// Why.  The walk of blend_backward_kernel is bound by VALU issue, not by memory or LDS: a full step is ~165 VALU instructions per
// wave (ISA count; 2.35 M wave steps x 660 cycles on 1,024 SIMDs = 0.63 ms of the kernel's 0.80 ms in the trained-like regime, 4 views at
// 256^2), 45 of them the cross-lane reduction of the nine sums, ~100 plain fp32 multiplies / adds / selects.  gfx950 issues fp32
// multiplies, adds and FMAs on register PAIRS at the same cost (v_pk_mul_f32, v_pk_add_f32): a lane that owns two pixels does the
// arithmetic of both with one instruction each, and the reduction (one add, then three DPP steps over 8 lanes instead of four over
// 16) serves 16 pixels with 4 instructions per value instead of 5 per value.  Cells, lists and the per-(cell, entry) culling are
// unchanged: a wave's eight 8-lane groups own eight 4 x 4 cells (wave w: strips 2w, 2w+1), lane q of a group the pixels
// (q & 3, 2 (q >> 2)) and (q & 3, 2 (q >> 2) + 1) of its cell -- same x, so dx is shared; a step covers 128 (pixel, entry) pairs.
// Per pixel the arithmetic is the reference's, in the reference's order (every operation on a pair is the scalar operation
// on each half); the masks are applied to one factor (0 x finite) instead of to the nine products, which adds (+-)0 to a sum
// where the other kernel adds +0.  The order in which a cell's 16 pixels are added differs from the other kernel's.
// Batches of 256 entries are staged by 128 threads, two entries each (entry e of the batch by thread e & 127).
template <bool FAST_EXP, bool DET>
__global__ __launch_bounds__(128) void blend_backward_pair_kernel(BwdParams p) {
    __shared__ uint32_t s_id[256];
    __shared__ uint2 s_stat[kRasterStats ? 2 : 1];
    __shared__ float2 s_xy[256];
    __shared__ float4 s_co[256];
    __shared__ float4 s_rgbc[256];
    __shared__ uint32_t s_max[2];
    __shared__ uint4 s_cnt[16];                           // [cell] entries of the batch the cell keeps, per 64 entries of the batch
    __shared__ uint8_t s_list[17][256];
    constexpr int NACC = DET ? 2 : 1;                     // deterministic form: a copy per wave, added in wave order (see blend_backward_kernel)
    __shared__ float s_acc[NACC][9][kAccRow];
    const uint32_t vt = p.im.tile_order[blockIdx.x];
    const int v = (int)(vt / (uint32_t)p.T), tile = (int)(vt % (uint32_t)p.T), s = v / p.vps;
    const int bx = tile % p.gx, by = tile / p.gx;
    const int tid = threadIdx.x, lane = tid & 63, grp = lane >> 3, q = lane & 7;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cell = wave * 8 + grp;                      // cell_mask bit: 4 * strip + column
    const int pxi = bx * kTile + 4 * (cell & 3) + (q & 3), pyi = by * kTile + 4 * (cell >> 2) + 2 * (q >> 2);
    const bool in0 = pxi < p.W && pyi < p.H, in1 = pxi < p.W && pyi + 1 < p.H;
    const float pfx = (float)pxi;
    const v2f pfy = {(float)pyi, (float)(pyi + 1)};
    const float tx0 = (float)(bx * kTile), ty0 = (float)(by * kTile);
    const unsigned long long lanes_before = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    const size_t HW = (size_t)p.H * p.W, pid0 = (size_t)p.W * pyi + pxi, pid1 = pid0 + (size_t)p.W;
    const uint2 rg = p.im.ranges[vt];
    const size_t vo = (size_t)v * p.P;

    const v2f T_final = {in0 ? p.im.final_T[(size_t)v * HW + pid0] : 0.0f, in1 ? p.im.final_T[(size_t)v * HW + pid1] : 0.0f};
    v2f T = T_final;
    const uint32_t lc0 = in0 ? p.im.n_contrib[(size_t)v * HW + pid0] : 0u, lc1 = in1 ? p.im.n_contrib[(size_t)v * HW + pid1] : 0u;
    v2f dpix[3];
    {
        const float* g = p.dL_dpix + (size_t)v * 3 * HW;
#pragma unroll
        for (int c = 0; c < 3; ++c) dpix[c] = v2f{in0 ? g[c * HW + pid0] : 0.f, in1 ? g[c * HW + pid1] : 0.f};
    }
    v2f bg_dot = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) bg_dot += p.bg[c] * dpix[c];
    v2f accum_rec[3], last_color[3], last_alpha = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) { accum_rec[c] = v2f{0.f, 0.f}; last_color[c] = v2f{0.f, 0.f}; }
    const float ddelx_dx = (float)(0.5 * p.W), ddely_dy = (float)(0.5 * p.H);

    uint32_t mc = max(lc0, lc1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mc = max(mc, (uint32_t)__shfl_xor((int)mc, o));
    if (lane == 0) s_max[wave] = mc;
#pragma unroll
    for (int w = 0; w < NACC; ++w)
#pragma unroll
        for (int k = 0; k < 9; ++k) { s_acc[w][k][tid] = 0.f; s_acc[w][k][tid + 128] = 0.f; }
    __syncthreads();
    const uint32_t todo = max(s_max[0], s_max[1]);             // <= rg.y - rg.x
    const int rounds = (int)((todo + 255u) / 256u);
    const bool colors_per_set = p.colors_pre != nullptr;
    uint32_t st_entries = 0, st_trips = 0;
    for (int i = 0; i < rounds; ++i) {
        // batch entry e (0..255) = 1-based list index contributor = todo - (i * 256 + e), staged by thread e & 127
        unsigned m16[2] = {0u, 0u};
        int idxs[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int e = h * 128 + tid;
            const int idx = (int)todo - 1 - (i * 256 + e);
            idxs[h] = idx;
            if (idx >= 0) {
                const uint32_t id = p.bn.point_list[rg.x + (uint32_t)idx];
                const BlendRecord* rec = p.g.blend + vo + id;
                const float4 co = rec->co;
                float4 rc = rec->rc;
                const float2 xy = rec->xy;
                m16[h] = cell_mask(xy, co, rc.w, tx0, ty0);
                if (colors_per_set) {
                    const float* c = p.colors_pre + 3 * ((size_t)s * p.P + id);
                    rc.x = c[0]; rc.y = c[1]; rc.z = c[2];
                }
                s_id[e] = id; s_xy[e] = xy; s_co[e] = co; s_rgbc[e] = rc;
            } else {
                s_id[e] = 0u; s_xy[e] = make_float2(0.f, 0.f); s_co[e] = make_float4(0.f, 0.f, 0.f, 0.f); s_rgbc[e] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const unsigned long long keep = __ballot((m16[h] >> c) & 1u);
                if (lane == 0) reinterpret_cast<uint32_t*>(&s_cnt[c])[2 * h + wave] = (uint32_t)__popcll(keep);
            }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const unsigned long long keep = __ballot((m16[h] >> c) & 1u);
                if ((m16[h] >> c) & 1u) {
                    const uint4 cn = s_cnt[c];
                    const int part = 2 * h + wave;                      // which 64 entries of the batch
                    const uint32_t ahead = (part > 0 ? cn.x : 0u) + (part > 1 ? cn.y : 0u) + (part > 2 ? cn.z : 0u);
                    s_list[c][ahead + (uint32_t)__popcll(keep & lanes_before)] = (uint8_t)(h * 128 + tid);
                }
            }
        __syncthreads();
        {
            const uint4 cn = s_cnt[cell];
            const uint32_t tot = cn.x + cn.y + cn.z + cn.w;
            const uint32_t first = todo - (uint32_t)(i * 256);
            struct Entry { float2 xy; float4 co; float4 rc; };
            auto load = [&](uint32_t j) { return Entry{s_xy[j], s_co[j], s_rgbc[j]}; };
            auto step = [&](uint32_t k, uint32_t j, const Entry& e) {
                const float dx = e.xy.x - pfx;
                const v2f dy = e.xy.y - pfy;
                const v2f power = -0.5f * (e.co.x * dx * dx + e.co.z * dy * dy) - e.co.y * dx * dy;
                const uint32_t back = first - j;                         // 1-based list index of the entry (backward.cu:463-468)
                const bool near0 = (k < tot) & in0 & (back <= lc0) & !(power.x > 0.0f) & !(power.x < e.rc.w);
                const bool near1 = (k < tot) & in1 & (back <= lc1) & !(power.y > 0.0f) & !(power.y < e.rc.w);
                if (wave_ballot(near0 | near1) != 0ull) {
                    const v2f G = {blend_exp<FAST_EXP>(near0 ? power.x : 0.0f, e.co.w), blend_exp<FAST_EXP>(near1 ? power.y : 0.0f, e.co.w)};
                    v2f alpha = e.co.w * G;
                    alpha.x = alpha_clamp(alpha.x); alpha.y = alpha_clamp(alpha.y);
                    const bool take0 = near0 & !(alpha.x < 1.0f / 255.0f), take1 = near1 & !(alpha.y < 1.0f / 255.0f);
                    const unsigned long long takers = wave_ballot(take0 | take1);
                    if (takers != 0ull) {
                        const v2f one_m = 1.f - alpha;
                        v2f inv1ma = {0.f, 0.f}, Tn;
                        if constexpr (FAST_EXP) { inv1ma = v2f{hw_rcp(one_m.x), hw_rcp(one_m.y)}; Tn = T * inv1ma; }
                        else Tn = T / one_m;
                        const v2f dch = sel2(take0, take1, alpha * Tn, v2f{0.f, 0.f});
                        const v2f la1 = 1.f - last_alpha;
                        v2f c9[9];
                        v2f dLa = {0.f, 0.f};
                        auto channel = [&](int ch, float col) {
                            const v2f rec = last_alpha * last_color[ch] + la1 * accum_rec[ch];
                            dLa += (col - rec) * dpix[ch];
                            c9[ch] = dch * dpix[ch];
                            accum_rec[ch] = sel2(take0, take1, rec, accum_rec[ch]);
                            last_color[ch] = sel2(take0, take1, v2f{col, col}, last_color[ch]);
                        };
                        channel(0, e.rc.x); channel(1, e.rc.y); channel(2, e.rc.z);
                        dLa *= Tn;
                        if constexpr (FAST_EXP) dLa += (-T_final * inv1ma) * bg_dot;
                        else dLa += (-T_final / one_m) * bg_dot;
                        T = sel2(take0, take1, Tn, T);
                        last_alpha = sel2(take0, take1, alpha, last_alpha);
                        const v2f dLa_m = sel2(take0, take1, dLa, v2f{0.f, 0.f});
                        const v2f dL_dG = e.co.w * dLa_m;
                        const v2f gdx = G * dx, gdy = G * dy;
                        const v2f dG_ddelx = -gdx * e.co.x - gdy * e.co.y;
                        const v2f dG_ddely = -gdy * e.co.z - gdx * e.co.y;
                        c9[3] = dL_dG * dG_ddelx * ddelx_dx;
                        c9[4] = dL_dG * dG_ddely * ddely_dy;
                        c9[5] = -0.5f * gdx * dx * dL_dG;
                        c9[6] = -0.5f * gdx * dy * dL_dG;
                        c9[7] = -0.5f * gdy * dy * dL_dG;
                        c9[8] = G * dLa_m;
                        const bool reduce = !(kRasterAblate && (p.ablate & 16)), add = !(kRasterAblate && (p.ablate & 2));
                        float s9[9];
#pragma unroll
                        for (int z = 0; z < 9; ++z) s9[z] = reduce ? oct_sum_to_lane7(c9[z].x + c9[z].y) : c9[z].x + c9[z].y;
                        if (add && q == 7 && ((takers >> (8 * grp)) & 0xFFull) != 0ull) {
#pragma unroll
                            for (int z = 0; z < 9; ++z) lds_add(&s_acc[DET ? wave : 0][z][j], s9[z]);
                        }
                    }
                }
            };
            const uint32_t* lst = reinterpret_cast<const uint32_t*>(s_list[cell]);
            uint32_t word = lst[0];
            Entry ea = load(word & 255u), eb;
            if constexpr (kRasterStats) st_entries += tot;
            uint32_t k = 0;
            for (; wave_ballot(k < tot) != 0ull && !(kRasterAblate && (p.ablate & 4)); k += 4) {
                const uint32_t word_next = lst[(k >> 2) + 1u];
                eb = load((word >> 8) & 255u);  step(k, word & 255u, ea);
                ea = load((word >> 16) & 255u); step(k + 1u, (word >> 8) & 255u, eb);
                eb = load(word >> 24);          step(k + 2u, (word >> 16) & 255u, ea);
                ea = load(word_next & 255u);    step(k + 3u, word >> 24, eb);
                word = word_next;
            }
            if constexpr (kRasterStats) st_trips += 2u * k;          // a trip issues 128 pixel slots: counted as two of the 64-slot trips
        }
        __syncthreads();
        if constexpr (!DET) {
            if (!(kRasterAblate && (p.ablate & 1))) {
                flush_records<16>(s_acc[0], s_id, p.g.grad_acc, vo, 64 * wave, lane);            // the entries this wave staged: e = h * 128 + tid
                flush_records<16>(s_acc[0], s_id, p.g.grad_acc, vo, 128 + 64 * wave, lane);
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int e = h * 128 + tid;
                float c9[9];
                bool any = false;
#pragma unroll
                for (int z = 0; z < 9; ++z) {
                    const float a0 = s_acc[0][z][e], a1 = s_acc[1][z][e];
                    c9[z] = a0 + a1;
                    any = any || a0 != 0.f || a1 != 0.f;           // the copies, not their sum (see blend_backward_kernel)
                }
                if (idxs[h] >= 0) {
                    if (any) {
#pragma unroll
                        for (int w = 0; w < NACC; ++w)
#pragma unroll
                            for (int z = 0; z < 9; ++z) s_acc[w][z][e] = 0.f;
                    }
                    const uint32_t id = s_id[e];
                    const size_t gv = vo + id;
                    int x0, y0, x1, y1;
                    tile_rect(s_xy[e].x, s_xy[e].y, p.radii[gv], p.gx, p.gy, &x0, &y0, &x1, &y1);
                    const size_t slot = (size_t)p.slot_base[gv] + (size_t)((by - y0) * (x1 - x0) + (bx - x0));
                    p.slot_a[slot] = make_float4(c9[0], c9[1], c9[2], c9[3]);
                    p.slot_b[slot] = make_float4(c9[4], c9[5], c9[6], c9[7]);
                    p.slot_c[slot] = c9[8];
                }
            }
        }
    }
    if constexpr (DET) {
        if (tid == 0) {
            unsigned long long key = 0ull;
            if (todo > 0) {
                const uint32_t id = p.bn.point_list[rg.x + todo - 1u];
                key = ((unsigned long long)__float_as_uint(p.g.depths[vo + id]) << 32) | id;
            }
            p.last_key[vt] = key;
        }
    }
    if constexpr (kRasterStats) {
        uint32_t ent = q == 0 ? st_entries : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ent += (uint32_t)__shfl_xor((int)ent, o);
        if (lane == 0) s_stat[wave] = make_uint2(ent, st_trips);
        __syncthreads();
        if (tid == 0)
            p.im.tile_stats[(size_t)p.V * p.T + vt] = make_uint4(s_stat[0].x + s_stat[1].x, s_stat[0].y + s_stat[1].y, 0u, (uint32_t)rounds);
    }
}

// backward.cu:20-139 for one Gaussian of one view.  dRGB already has the clamp mask applied.  Adds into dsh[M][3] and dmean.
__device__ __forceinline__ void sh_backward(int deg, const float* sh, float ox, float oy, float oz, const float* dRGB, float* dsh,
                                            float* dmean) {
    const float len = sqrtf(ox * ox + oy * oy + oz * oz);
    const float x = ox / len, y = oy / len, z = oz / len;
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    float dx[3] = {0, 0, 0}, dy[3] = {0, 0, 0}, dz[3] = {0, 0, 0};
#define SHV(k, c) sh[3 * (k) + (c)]
#define OUT(k, w)                                          \
    do {                                                   \
        dsh[3 * (k)] += (w) * dRGB[0];                     \
        dsh[3 * (k) + 1] += (w) * dRGB[1];                 \
        dsh[3 * (k) + 2] += (w) * dRGB[2];                 \
    } while (0)
    OUT(0, C0);
    if (deg > 0) {
        OUT(1, -C1 * y); OUT(2, C1 * z); OUT(3, -C1 * x);
#pragma unroll
        for (int c = 0; c < 3; ++c) { dx[c] = -C1 * SHV(3, c); dy[c] = -C1 * SHV(1, c); dz[c] = C1 * SHV(2, c); }
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            OUT(4, C2[0] * xy); OUT(5, C2[1] * yz); OUT(6, C2[2] * (2.f * zz - xx - yy)); OUT(7, C2[3] * xz); OUT(8, C2[4] * (xx - yy));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dx[c] += C2[0] * y * SHV(4, c) + C2[2] * 2.f * -x * SHV(6, c) + C2[3] * z * SHV(7, c) + C2[4] * 2.f * x * SHV(8, c);
                dy[c] += C2[0] * x * SHV(4, c) + C2[1] * z * SHV(5, c) + C2[2] * 2.f * -y * SHV(6, c) + C2[4] * 2.f * -y * SHV(8, c);
                dz[c] += C2[1] * y * SHV(5, c) + C2[2] * 2.f * 2.f * z * SHV(6, c) + C2[3] * x * SHV(7, c);
            }
            if (deg > 2) {
                OUT(9, C3[0] * y * (3.f * xx - yy)); OUT(10, C3[1] * xy * z); OUT(11, C3[2] * y * (4.f * zz - xx - yy));
                OUT(12, C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)); OUT(13, C3[4] * x * (4.f * zz - xx - yy));
                OUT(14, C3[5] * z * (xx - yy)); OUT(15, C3[6] * x * (xx - 3.f * yy));
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    dx[c] += C3[0] * SHV(9, c) * 3.f * 2.f * xy + C3[1] * SHV(10, c) * yz + C3[2] * SHV(11, c) * -2.f * xy +
                             C3[3] * SHV(12, c) * -3.f * 2.f * xz + C3[4] * SHV(13, c) * (-3.f * xx + 4.f * zz - yy) +
                             C3[5] * SHV(14, c) * 2.f * xz + C3[6] * SHV(15, c) * 3.f * (xx - yy);
                    dy[c] += C3[0] * SHV(9, c) * 3.f * (xx - yy) + C3[1] * SHV(10, c) * xz + C3[2] * SHV(11, c) * (-3.f * yy + 4.f * zz - xx) +
                             C3[3] * SHV(12, c) * -3.f * 2.f * yz + C3[4] * SHV(13, c) * -2.f * xy + C3[5] * SHV(14, c) * -2.f * yz +
                             C3[6] * SHV(15, c) * -3.f * 2.f * xy;
                    dz[c] += C3[1] * SHV(10, c) * xy + C3[2] * SHV(11, c) * 4.f * 2.f * yz + C3[3] * SHV(12, c) * 3.f * (2.f * zz - xx - yy) +
                             C3[4] * SHV(13, c) * 4.f * 2.f * xz + C3[5] * SHV(14, c) * (xx - yy);
                }
            }
        }
    }
#undef OUT
#undef SHV
    const float ddx = dx[0] * dRGB[0] + dx[1] * dRGB[1] + dx[2] * dRGB[2];
    const float ddy = dy[0] * dRGB[0] + dy[1] * dRGB[1] + dy[2] * dRGB[2];
    const float ddz = dz[0] * dRGB[0] + dz[1] * dRGB[1] + dz[2] * dRGB[2];
    // dnormvdv, auxiliary.h:107-117
    const float sum2 = ox * ox + oy * oy + oz * oz;
    const float inv = 1.0f / sqrtf(sum2 * sum2 * sum2);
    dmean[0] += ((+sum2 - ox * ox) * ddx - oy * ox * ddy - oz * ox * ddz) * inv;
    dmean[1] += (-ox * oy * ddx + (sum2 - oy * oy) * ddy - oz * oy * ddz) * inv;
    dmean[2] += (-ox * oz * ddx - oy * oz * ddy + (sum2 - oz * oz) * ddz) * inv;
}

// The backward of a tile replays exactly the entries its forward walked (tile_work): its launch order is dealt by that.
__global__ __launch_bounds__(1024) void order_tiles_kernel(const uint32_t* work, int n, uint32_t* order) {
    __shared__ uint32_t scratch[20];
    __shared__ uint32_t smax;
    __shared__ uint32_t s_class[1024];
    if (threadIdx.x == 0) smax = 0;
    __syncthreads();
    uint32_t mx = 0;
    for (int i = threadIdx.x; i < n; i += 1024) mx = max(mx, work[i]);
    atomicMax(&smax, mx);
    __syncthreads();
    deal_tiles(work, n, smax, order, s_class, scratch);
}

// Deterministic form: grid (ceil(P / 256), V), one thread per (view, Gaussian).  It adds the slots of the tiles that replayed the
// Gaussian, in rectangle order (row-major), and writes the nine sums as the (view, Gaussian)'s gradient record -- what the atomic form
// accumulates with atomics; preprocess_backward_kernel reads either.  The tiles' keys sit in LDS (8 bytes per tile): a Gaussian of the
// random-init regime tests ~50 of them.
template <bool LDS_KEYS>
__global__ __launch_bounds__(256) void gather_partials_kernel(BwdParams p) {
    DGS_DYNAMIC_LDS(smem);
    const int v = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
    const unsigned long long* lk = p.last_key + (size_t)v * p.T;
    if (LDS_KEYS) {
        unsigned long long* s_lk = reinterpret_cast<unsigned long long*>(smem);
        for (int i = threadIdx.x; i < p.T; i += 256) s_lk[i] = lk[i];
        __syncthreads();
        lk = s_lk;
    }
    if (idx >= p.P) return;
    const size_t gi = (size_t)v * p.P + idx;
    float a[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int radius = p.radii[gi];
    if (radius > 0) {
        const float2 m = p.g.means2D[gi];
        int x0, y0, x1, y1;
        tile_rect(m.x, m.y, radius, p.gx, p.gy, &x0, &y0, &x1, &y1);
        const unsigned long long key = ((unsigned long long)__float_as_uint(p.g.depths[gi]) << 32) | (uint32_t)idx;
        size_t slot = p.slot_base[gi];
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x, ++slot) {
                if (key <= lk[y * p.gx + x]) {
                    const float4 sa = p.slot_a[slot], sb = p.slot_b[slot];
                    a[0] += sa.x; a[1] += sa.y; a[2] += sa.z; a[3] += sa.w;
                    a[4] += sb.x; a[5] += sb.y; a[6] += sb.z; a[7] += sb.w;
                    a[8] += p.slot_c[slot];
                }
            }
    }
    if (radius > 0) {      // the record of a culled Gaussian is never read
        float4* rec = reinterpret_cast<float4*>(p.g.grad_acc + 16 * gi);
        rec[0] = make_float4(a[0], a[1], a[2], a[3]);
        rec[1] = make_float4(a[4], a[5], a[6], a[7]);
        rec[2] = make_float4(a[8], 0.f, 0.f, 0.f);
    }
}

// grid ceil(S*P / 256).  One thread per (set, Gaussian); loops over the views of the set.
// MAXM: SH coefficients the per-thread gradient array is sized for.  DiffusionGS trains degree 0 (one coefficient: `gaussians_sh_degree 0`,
// denoiser.py:96): with the 48-float array of degree 3 the kernel needs 213 registers -- two waves per SIMD on a kernel that is a chain
// of memory round trips; sized for what the call has it runs at twice the occupancy.
template <int MAXM>
__global__ __launch_bounds__(256) void preprocess_backward_kernel(BwdParams p, int S) {
    const size_t si = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (si >= (size_t)S * p.P) return;
    const int s = (int)(si / p.P), idx = (int)(si % p.P);
    const float mx = p.means3D[3 * si], my = p.means3D[3 * si + 1], mz = p.means3D[3 * si + 2];
    float dmean[3] = {0.f, 0.f, 0.f};
    float dcov_sum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float dsh[3 * MAXM];
    const int nsh = p.shs ? 3 * p.M : 0;
#pragma unroll
    for (int k = 0; k < 3 * MAXM; ++k) dsh[k] = 0.f;
    const int v0 = s * p.vps, v1 = min(p.V, v0 + p.vps);
    // the 3D covariance does not depend on the view: once per Gaussian, from the inputs (the forward's instruction sequence, raster_common.h)
    float c6[6];
    if (p.cov_pre) {
#pragma unroll
        for (int k = 0; k < 6; ++k) c6[k] = p.cov_pre[6 * si + k];
    } else {
        cov3d_from_scale_rot(p.scales + 3 * si, p.rots + 4 * si, p.raw_act != 0, p.scale_mod, c6);
    }
    // The blend kernel's nine sums per (view, Gaussian) arrive as one 64-byte record (GeomState::grad_acc: accumulated by atomics, or
    // written by gather_partials_kernel); the per-view rows the reference returns (dL_dmeans2D, dL_dconic, dL_dcolors) are written from it
    // here -- every element, nothing is pre-filled -- and the per-set sums (opacity; colours when they are per set) are added over the
    // views of the set in view order.
    float op_sum = 0.f, col_sum[3] = {0.f, 0.f, 0.f};
    for (int v = v0; v < v1; ++v) {
        const size_t gi = (size_t)v * p.P + idx;
        if (!(p.radii[gi] > 0)) {                  // culled in this view: its rows are zero
#pragma unroll
            for (int k = 0; k < 6; ++k) p.dL_dcov3D[6 * gi + k] = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) p.dL_dmean2D[3 * gi + k] = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) p.dL_dconic[4 * gi + k] = 0.f;
            if (!p.colors_pre) { p.dL_dcolors[3 * gi] = 0.f; p.dL_dcolors[3 * gi + 1] = 0.f; p.dL_dcolors[3 * gi + 2] = 0.f; }
            continue;
        }
        const float4* rec = reinterpret_cast<const float4*>(p.g.grad_acc + 16 * gi);
        const float4 r0 = rec[0], r1 = rec[1];
        op_sum += p.g.grad_acc[16 * gi + 8];
        p.dL_dmean2D[3 * gi] = r0.w; p.dL_dmean2D[3 * gi + 1] = r1.x; p.dL_dmean2D[3 * gi + 2] = 0.f;
        p.dL_dconic[4 * gi] = r1.y; p.dL_dconic[4 * gi + 1] = r1.z; p.dL_dconic[4 * gi + 2] = 0.f; p.dL_dconic[4 * gi + 3] = r1.w;
        if (p.colors_pre) { col_sum[0] += r0.x; col_sum[1] += r0.y; col_sum[2] += r0.z; }
        else { p.dL_dcolors[3 * gi] = r0.x; p.dL_dcolors[3 * gi + 1] = r0.y; p.dL_dcolors[3 * gi + 2] = r0.z; }
        const float* vm = p.viewm + 16 * v;
        const float* proj = p.projm + 16 * v;
        float tanx, tany;
        if (p.tanfov) { tanx = p.tanfov[2 * v]; tany = p.tanfov[2 * v + 1]; } else { tanx = p.tanfovx; tany = p.tanfovy; }
        const float focal_y = p.H / (2.0f * tany), focal_x = p.W / (2.0f * tanx);
        // ---- computeCov2DCUDA, backward.cu:144-274 ----
        float tx = vm[0] * mx + vm[4] * my + vm[8] * mz + vm[12];
        float ty = vm[1] * mx + vm[5] * my + vm[9] * mz + vm[13];
        const float tz = vm[2] * mx + vm[6] * my + vm[10] * mz + vm[14];
        const float limx = 1.3f * tanx, limy = 1.3f * tany;
        const float txtz = tx / tz, tytz = ty / tz;
        tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
        ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
        const float xgm = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
        const float ygm = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
        const M3 J = m3_cols(focal_x / tz, 0.0f, -(focal_x * tx) / (tz * tz), 0.0f, focal_y / tz, -(focal_y * ty) / (tz * tz), 0, 0, 0);
        const M3 Wm = m3_cols(vm[0], vm[4], vm[8], vm[1], vm[5], vm[9], vm[2], vm[6], vm[10]);
        const M3 Tm = m3_mul(Wm, J);
        const M3 Vrk = m3_cols(c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]);
        const M3 cov = m3_mul(m3_mul(m3_t(Tm), m3_t(Vrk)), Tm);
        const float a = cov.c[0][0] + 0.3f, b = cov.c[0][1], c = cov.c[1][1] + 0.3f;
        const float denom = a * c - b * b;
        const float dcx = r1.y, dcy = r1.z, dcz = r1.w;
        float dL_da = 0, dL_db = 0, dL_dc = 0;
        const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
        float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const float (*T)[3] = Tm.c;
        if (denom2inv != 0) {
            dL_da = denom2inv * (-c * c * dcx + 2 * b * c * dcy + (denom - a * c) * dcz);
            dL_dc = denom2inv * (-a * a * dcz + 2 * a * b * dcy + (denom - a * c) * dcx);
            dL_db = denom2inv * 2 * (b * c * dcx - (denom + 2 * b * b) * dcy + a * b * dcz);
            dcov[0] = (T[0][0] * T[0][0] * dL_da + T[0][0] * T[1][0] * dL_db + T[1][0] * T[1][0] * dL_dc);
            dcov[3] = (T[0][1] * T[0][1] * dL_da + T[0][1] * T[1][1] * dL_db + T[1][1] * T[1][1] * dL_dc);
            dcov[5] = (T[0][2] * T[0][2] * dL_da + T[0][2] * T[1][2] * dL_db + T[1][2] * T[1][2] * dL_dc);
            dcov[1] = 2 * T[0][0] * T[0][1] * dL_da + (T[0][0] * T[1][1] + T[0][1] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][1] * dL_dc;
            dcov[2] = 2 * T[0][0] * T[0][2] * dL_da + (T[0][0] * T[1][2] + T[0][2] * T[1][0]) * dL_db + 2 * T[1][0] * T[1][2] * dL_dc;
            dcov[4] = 2 * T[0][2] * T[0][1] * dL_da + (T[0][1] * T[1][2] + T[0][2] * T[1][1]) * dL_db + 2 * T[1][1] * T[1][2] * dL_dc;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) { p.dL_dcov3D[6 * gi + k] = dcov[k]; dcov_sum[k] += dcov[k]; }
        const float (*Vm)[3] = Vrk.c;
        const float dT00 = 2 * (T[0][0] * Vm[0][0] + T[0][1] * Vm[0][1] + T[0][2] * Vm[0][2]) * dL_da + (T[1][0] * Vm[0][0] + T[1][1] * Vm[0][1] + T[1][2] * Vm[0][2]) * dL_db;
        const float dT01 = 2 * (T[0][0] * Vm[1][0] + T[0][1] * Vm[1][1] + T[0][2] * Vm[1][2]) * dL_da + (T[1][0] * Vm[1][0] + T[1][1] * Vm[1][1] + T[1][2] * Vm[1][2]) * dL_db;
        const float dT02 = 2 * (T[0][0] * Vm[2][0] + T[0][1] * Vm[2][1] + T[0][2] * Vm[2][2]) * dL_da + (T[1][0] * Vm[2][0] + T[1][1] * Vm[2][1] + T[1][2] * Vm[2][2]) * dL_db;
        const float dT10 = 2 * (T[1][0] * Vm[0][0] + T[1][1] * Vm[0][1] + T[1][2] * Vm[0][2]) * dL_dc + (T[0][0] * Vm[0][0] + T[0][1] * Vm[0][1] + T[0][2] * Vm[0][2]) * dL_db;
        const float dT11 = 2 * (T[1][0] * Vm[1][0] + T[1][1] * Vm[1][1] + T[1][2] * Vm[1][2]) * dL_dc + (T[0][0] * Vm[1][0] + T[0][1] * Vm[1][1] + T[0][2] * Vm[1][2]) * dL_db;
        const float dT12 = 2 * (T[1][0] * Vm[2][0] + T[1][1] * Vm[2][1] + T[1][2] * Vm[2][2]) * dL_dc + (T[0][0] * Vm[2][0] + T[0][1] * Vm[2][1] + T[0][2] * Vm[2][2]) * dL_db;
        const float dJ00 = Wm.c[0][0] * dT00 + Wm.c[0][1] * dT01 + Wm.c[0][2] * dT02;
        const float dJ02 = Wm.c[2][0] * dT00 + Wm.c[2][1] * dT01 + Wm.c[2][2] * dT02;
        const float dJ11 = Wm.c[1][0] * dT10 + Wm.c[1][1] * dT11 + Wm.c[1][2] * dT12;
        const float dJ12 = Wm.c[2][0] * dT10 + Wm.c[2][1] * dT11 + Wm.c[2][2] * dT12;
        const float itz = 1.f / tz, tz2 = itz * itz, tz3 = tz2 * itz;
        const float dtx = xgm * -focal_x * tz2 * dJ02;
        const float dty = ygm * -focal_y * tz2 * dJ12;
        const float dtz = -focal_x * tz2 * dJ00 - focal_y * tz2 * dJ11 + (2 * focal_x * tx) * tz3 * dJ02 + (2 * focal_y * ty) * tz3 * dJ12;
        // transformVec4x3Transpose, auxiliary.h:97-105
        float dmv[3];
        dmv[0] = vm[0] * dtx + vm[1] * dty + vm[2] * dtz;
        dmv[1] = vm[4] * dtx + vm[5] * dty + vm[6] * dtz;
        dmv[2] = vm[8] * dtx + vm[9] * dty + vm[10] * dtz;
        // ---- preprocessCUDA backward, backward.cu:346-396 ----
        const float hw = proj[3] * mx + proj[7] * my + proj[11] * mz + proj[15];
        const float m_w = 1.0f / (hw + 0.0000001f);
        const float mul1 = (proj[0] * mx + proj[4] * my + proj[8] * mz + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mx + proj[5] * my + proj[9] * mz + proj[13]) * m_w * m_w;
        const float d2x = r0.w, d2y = r1.x;
        dmv[0] += (proj[0] * m_w - proj[3] * mul1) * d2x + (proj[1] * m_w - proj[3] * mul2) * d2y;
        dmv[1] += (proj[4] * m_w - proj[7] * mul1) * d2x + (proj[5] * m_w - proj[7] * mul2) * d2y;
        dmv[2] += (proj[8] * m_w - proj[11] * mul1) * d2x + (proj[9] * m_w - proj[11] * mul2) * d2y;
        if (p.shs) {
            const unsigned cb = p.g.clamped[gi];
            const float dRGB[3] = {(cb & 1u) ? 0.f : r0.x, (cb & 2u) ? 0.f : r0.y, (cb & 4u) ? 0.f : r0.z};
            const float* cam = p.campos + 3 * v;
            sh_backward(MAXM == 1 ? 0 : p.D, p.shs + 3 * (size_t)p.M * si, mx - cam[0], my - cam[1], mz - cam[2], dRGB, dsh, dmv);
        }
        dmean[0] += dmv[0]; dmean[1] += dmv[1]; dmean[2] += dmv[2];
    }
    p.dL_dmeans3D[3 * si] = dmean[0]; p.dL_dmeans3D[3 * si + 1] = dmean[1]; p.dL_dmeans3D[3 * si + 2] = dmean[2];
    if (p.dL_dsh) {
#pragma unroll
        for (int k = 0; k < 3 * MAXM; ++k)
            if (k < nsh) p.dL_dsh[(size_t)nsh * si + k] = dsh[k];
    }
    if (p.colors_pre) { p.dL_dcolors[3 * si] = col_sum[0]; p.dL_dcolors[3 * si + 1] = col_sum[1]; p.dL_dcolors[3 * si + 2] = col_sum[2]; }
    if (p.raw_act) {   // d sigmoid, gs_core.py:334
        const float op = 1.0f / (1.0f + det_expf(-p.opac[si]));
        op_sum = op_sum * (op * (1.0f - op));
    }
    p.dL_dopacity[si] = op_sum;
    if (!p.cov_pre && p.dL_dscales && p.dL_drots) {
        // ---- computeCov3D backward, backward.cu:278-341, on the view-summed dL/dSigma ----
        float sx = p.scales[3 * si], sy = p.scales[3 * si + 1], sz = p.scales[3 * si + 2];
        float qr = p.rots[4 * si], qx = p.rots[4 * si + 1], qy = p.rots[4 * si + 2], qz = p.rots[4 * si + 3];
        const float raw_q[4] = {qr, qx, qy, qz};
        float nrm = 1.0f;
        if (p.raw_act) {
            sx = det_expf(sx); sy = det_expf(sy); sz = det_expf(sz);
            nrm = fmaxf(sqrtf(qr * qr + qx * qx + qy * qy + qz * qz), 1e-12f);
            qr = qr / nrm; qx = qx / nrm; qy = qy / nrm; qz = qz / nrm;
        }
        const float r = qr, x = qx, y = qy, z = qz;
        const M3 R = m3_cols(1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y),
                             2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x),
                             2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y));
        M3 S = m3_cols(1, 0, 0, 0, 1, 0, 0, 0, 1);
        const float s3[3] = {p.scale_mod * sx, p.scale_mod * sy, p.scale_mod * sz};
        S.c[0][0] = s3[0]; S.c[1][1] = s3[1]; S.c[2][2] = s3[2];
        const M3 Mm = m3_mul(S, R);
        const float* d6 = dcov_sum;
        const M3 dSig = m3_cols(d6[0], 0.5f * d6[1], 0.5f * d6[2], 0.5f * d6[1], d6[3], 0.5f * d6[4], 0.5f * d6[2], 0.5f * d6[4], d6[5]);
        M3 M2;
#pragma unroll
        for (int cc = 0; cc < 3; ++cc)
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) M2.c[cc][rr] = Mm.c[cc][rr] * 2.0f;
        const M3 dM = m3_mul(M2, dSig);
        const M3 Rt = m3_t(R);
        M3 dMt = m3_t(dM);
        float gs[3];
        gs[0] = Rt.c[0][0] * dMt.c[0][0] + Rt.c[0][1] * dMt.c[0][1] + Rt.c[0][2] * dMt.c[0][2];
        gs[1] = Rt.c[1][0] * dMt.c[1][0] + Rt.c[1][1] * dMt.c[1][1] + Rt.c[1][2] * dMt.c[1][2];
        gs[2] = Rt.c[2][0] * dMt.c[2][0] + Rt.c[2][1] * dMt.c[2][1] + Rt.c[2][2] * dMt.c[2][2];
#pragma unroll
        for (int k = 0; k < 3; ++k) { dMt.c[0][k] *= s3[0]; dMt.c[1][k] *= s3[1]; dMt.c[2][k] *= s3[2]; }
        float gq[4];
        gq[0] = 2 * z * (dMt.c[0][1] - dMt.c[1][0]) + 2 * y * (dMt.c[2][0] - dMt.c[0][2]) + 2 * x * (dMt.c[1][2] - dMt.c[2][1]);
        gq[1] = 2 * y * (dMt.c[1][0] + dMt.c[0][1]) + 2 * z * (dMt.c[2][0] + dMt.c[0][2]) + 2 * r * (dMt.c[1][2] - dMt.c[2][1]) - 4 * x * (dMt.c[2][2] + dMt.c[1][1]);
        gq[2] = 2 * x * (dMt.c[1][0] + dMt.c[0][1]) + 2 * r * (dMt.c[2][0] - dMt.c[0][2]) + 2 * z * (dMt.c[1][2] + dMt.c[2][1]) - 4 * y * (dMt.c[2][2] + dMt.c[0][0]);
        gq[3] = 2 * r * (dMt.c[0][1] - dMt.c[1][0]) + 2 * x * (dMt.c[2][0] + dMt.c[0][2]) + 2 * y * (dMt.c[1][2] + dMt.c[2][1]) - 4 * z * (dMt.c[1][1] + dMt.c[0][0]);
        if (p.raw_act) {
            // gs is d/d(scale_mod * s) per component (backward.cu:329-331 writes it as dL_dscale): chain d exp = s
            gs[0] *= sx; gs[1] *= sy; gs[2] *= sz;
            // F.normalize backward (q = raw / max(|raw|, eps)): (g - q (q . g)) / |raw|
            const float qg = qr * gq[0] + qx * gq[1] + qy * gq[2] + qz * gq[3];
            const float qn[4] = {qr, qx, qy, qz};
#pragma unroll
            for (int k = 0; k < 4; ++k) gq[k] = (gq[k] - qn[k] * qg) / nrm;
            (void)raw_q;
        }
        p.dL_dscales[3 * si] = gs[0]; p.dL_dscales[3 * si + 1] = gs[1]; p.dL_dscales[3 * si + 2] = gs[2];
        p.dL_drots[4 * si] = gq[0]; p.dL_drots[4 * si + 1] = gq[1]; p.dL_drots[4 * si + 2] = gq[2]; p.dL_drots[4 * si + 3] = gq[3];
    }
}

// Which walk.  Two pixels per lane halves the VALU work of a step but also the number of waves (a tile is two waves instead of four):
// measured (profiles/r04_raster_backward_walks.txt, forward + backward per call) 4 views at 256^2 -- 1,024 workgroups, two waves per
// SIMD -- 1.194 ms against 1.144 (trained-like) and 1.080 against 0.970 (random init); 16 views at 256^2 3.299 against 3.429; 4 views
// at 512^2 3.093 against 3.202.  So: the two-pixel walk from 3,072 workgroups on (every CU then holds its six), the one-pixel walk
// below.  The deterministic form takes the two-pixel walk always (its LDS holds a copy of the accumulators per wave: two instead of
// four).  DGS_RASTER_BWD_WALK=1 | 2 forces one (A/B runs).
template <bool DET>
static bool pair_walk(int workgroups) {
    static const int forced = [] { const char* e = getenv("DGS_RASTER_BWD_WALK"); return e ? atoi(e) : 0; }();
    if (forced == 1 || forced == 2) return forced == 2;
    return DET || workgroups >= 3072;
}

template <bool DET>
static void launch_blend_backward(const BwdParams& p, int V, hipStream_t st) {
    const dim3 grid((unsigned)(V * p.T));
    size_t pad = 0;                // tools' library: DGS_RASTER_BWD_LDS_PAD = bytes of unused LDS per workgroup (fewer workgroups per CU)
    if constexpr (kRasterAblate) { static const int e = [] { const char* v = getenv("DGS_RASTER_BWD_LDS_PAD"); return v ? atoi(v) : 0; }(); pad = (size_t)e; }
    if (pair_walk<DET>(V * p.T)) {
        if (p.exact_exp) hipLaunchKernelGGL((blend_backward_pair_kernel<false, DET>), grid, dim3(128), pad, st, p);
        else hipLaunchKernelGGL((blend_backward_pair_kernel<true, DET>), grid, dim3(128), pad, st, p);
    } else {
        if (p.exact_exp) hipLaunchKernelGGL((blend_backward_kernel<false, DET>), grid, dim3(256), pad, st, p);
        else hipLaunchKernelGGL((blend_backward_kernel<true, DET>), grid, dim3(256), pad, st, p);
    }
}

static void launch_preprocess_backward(const BwdParams& p, int S, size_t ns, hipStream_t st) {
    const dim3 grid((unsigned)((ns + 255) / 256));
    // one coefficient (degree 0, or precomputed colours: no SH at all) / up to 16
    if (!p.shs || (p.M == 1 && p.D == 0)) hipLaunchKernelGGL(preprocess_backward_kernel<1>, grid, dim3(256), 0, st, p, S);
    else hipLaunchKernelGGL(preprocess_backward_kernel<16>, grid, dim3(256), 0, st, p, S);
}

}  // namespace dgs

using namespace dgs;

extern "C" int dgs_raster_backward(const DgsRasterBackwardArgs* a, dgs_stream_t stream) {
    (void)hipGetLastError();       // sticky per-thread error state of unrelated earlier calls is not ours to report
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!a || a->P < 0 || a->width <= 0 || a->height <= 0 || a->V < 1 || a->views_per_set < 1) return DGS_ERR_INVALID_ARGUMENT;
    const int P = a->P, V = a->V, W = a->width, H = a->height;
    const int S = (V + a->views_per_set - 1) / a->views_per_set;
    if (P == 0) return DGS_OK;
    if (!a->means3D || !a->viewmatrix || !a->projmatrix || !a->campos || !a->background || !a->radii || !a->dL_dpix ||
        !a->geom_buffer || !a->binning_buffer || !a->img_buffer)
        return DGS_ERR_INVALID_ARGUMENT;
    if (!a->dL_dmeans2D || !a->dL_dconic || !a->dL_dcolors || !a->dL_dcov3D || !a->dL_dopacity || !a->dL_dmeans3D)
        return DGS_ERR_INVALID_ARGUMENT;
    if ((a->shs == nullptr) == (a->colors_precomp == nullptr)) return DGS_ERR_NEED_COLORS;
    if (a->cov3D_precomp ? (a->scales || a->rotations) : !(a->scales && a->rotations)) return DGS_ERR_NEED_COVARIANCE;
    if (a->shs && (!a->dL_dsh || a->M < 1 || a->M > 16)) return DGS_ERR_INVALID_ARGUMENT;
    if (!a->cov3D_precomp && (!a->dL_dscales || !a->dL_drotations)) return DGS_ERR_INVALID_ARGUMENT;
    if (a->raw_activations && !a->opacities) return DGS_ERR_INVALID_ARGUMENT;

    BwdParams p{};
    p.P = P; p.D = a->D; p.M = a->M; p.W = W; p.H = H; p.V = V; p.vps = a->views_per_set;
    p.gx = (W + kTile - 1) / kTile; p.gy = (H + kTile - 1) / kTile; p.T = p.gx * p.gy; p.raw_act = a->raw_activations; p.exact_exp = a->exact_exp ? 1 : 0;
    p.bg = a->background; p.means3D = a->means3D; p.shs = a->shs; p.colors_pre = a->colors_precomp; p.opac = a->opacities;
    p.scales = a->scales; p.rots = a->rotations; p.cov_pre = a->cov3D_precomp; p.viewm = a->viewmatrix; p.projm = a->projmatrix;
    p.campos = a->campos; p.tanfov = a->tanfov; p.tanfovx = a->tanfovx; p.tanfovy = a->tanfovy; p.scale_mod = a->scale_modifier;
    p.radii = a->radii; p.dL_dpix = a->dL_dpix;
    if constexpr (kRasterAblate) { static const int ab = [] { const char* e = getenv("DGS_RASTER_BWD_ABLATE"); return e ? atoi(e) : 0; }(); p.ablate = ab; }
    p.g = GeomState::carve(const_cast<void*>(a->geom_buffer), (size_t)P, (size_t)V, nullptr);
    p.im = ImageState::carve(const_cast<void*>(a->img_buffer), (size_t)W, (size_t)H, (size_t)V, nullptr);
    p.bn = BinningState::carve(const_cast<void*>(a->binning_buffer), (size_t)(a->num_rendered < 1 ? 1 : a->num_rendered), nullptr);
    p.dL_dmean2D = a->dL_dmeans2D; p.dL_dconic = a->dL_dconic; p.dL_dcolors = a->dL_dcolors; p.dL_dcov3D = a->dL_dcov3D;
    p.dL_dopacity = a->dL_dopacity; p.dL_dmeans3D = a->dL_dmeans3D; p.dL_dsh = a->dL_dsh; p.dL_dscales = a->dL_dscales;
    p.dL_drots = a->dL_drotations;

    const size_t nv = (size_t)V * P, ns = (size_t)S * P;
    const bool det = a->scratch != nullptr;
    if (det) {
        // ---- deterministic form: slots instead of atomics, nothing to fill ----
        const size_t slots = (size_t)(a->num_rendered < 1 ? 1 : a->num_rendered);
        size_t need = 0;
        const BwdScratch sc = BwdScratch::carve(a->scratch, (size_t)P, (size_t)V, (size_t)p.T, slots, &need);
        if (a->scratch_bytes < need) return DGS_ERR_ALLOC;
        p.slot_base = sc.slot_base; p.last_key = sc.last_key; p.slot_a = sc.slot_a; p.slot_b = sc.slot_b; p.slot_c = sc.slot_c;
        const unsigned nb = (unsigned)((nv + 4095) / 4096);
        hipLaunchKernelGGL(touched_block_sums_kernel, dim3(nb), dim3(256), 0, st, p.g.tiles_touched, nv, sc.block_sums);
        hipLaunchKernelGGL(touched_scan_kernel, dim3(nb), dim3(256), 0, st, p.g.tiles_touched, nv, sc.block_sums, sc.slot_base);
        hipLaunchKernelGGL(order_tiles_kernel, dim3(1), dim3(1024), 0, st, p.im.tile_work, V * p.T, p.im.tile_order);
        launch_blend_backward<true>(p, V, st);
        if (a->debug && hipStreamSynchronize(st) != hipSuccess) return DGS_ERR_DEVICE;
        const dim3 gridPV((unsigned)((P + 255) / 256), (unsigned)V);
        if (p.T <= 4096) hipLaunchKernelGGL((gather_partials_kernel<true>), gridPV, dim3(256), (size_t)p.T * 8, st, p);
        else hipLaunchKernelGGL((gather_partials_kernel<false>), gridPV, dim3(256), 0, st, p);
        launch_preprocess_backward(p, S, ns, st);
        if (a->debug && hipStreamSynchronize(st) != hipSuccess) return DGS_ERR_DEVICE;
        return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
    }
    // The gradient records the blend kernel adds into start from zero (the reference's torch::zeros of its four accumulators,
    // rasterize_points.cu:148-156); the tensors of the interface are written in full by preprocess_backward_kernel.
    if (hipMemsetAsync(p.g.grad_acc, 0, nv * 16 * sizeof(float), st) != hipSuccess) return DGS_ERR_DEVICE;
    if (a->num_rendered != 0) hipLaunchKernelGGL(order_tiles_kernel, dim3(1), dim3(1024), 0, st, p.im.tile_work, V * p.T, p.im.tile_order);
    if (a->num_rendered != 0) launch_blend_backward<false>(p, V, st);
    if (a->debug && hipStreamSynchronize(st) != hipSuccess) return DGS_ERR_DEVICE;
    launch_preprocess_backward(p, S, ns, st);
    if (a->debug && hipStreamSynchronize(st) != hipSuccess) return DGS_ERR_DEVICE;
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}

extern "C" size_t dgs_raster_backward_scratch_bytes(int32_t P, int32_t width, int32_t height, int32_t V, int64_t num_rendered) {
    if (P <= 0 || width <= 0 || height <= 0 || V < 1) return 0;
    const size_t T = (size_t)((width + kTile - 1) / kTile) * ((height + kTile - 1) / kTile);
    size_t b = 0;
    BwdScratch::carve(nullptr, (size_t)P, (size_t)V, T, (size_t)(num_rendered < 1 ? 1 : num_rendered), &b);
    return b;
}
