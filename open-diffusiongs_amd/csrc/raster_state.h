// raster_state.h -- layout of the three opaque state buffers (geometry / image / binning).
//
// The reference keeps GeometryState / ImageState / BinningState (rasterizer_impl.h:29-65) inside three
// caller-owned byte buffers; the contents are private to the library, so the layout here is our own
// (data laid out for the MI355X pipeline: SoA arrays per (view, Gaussian) for the streaming kernels, one 64-byte
// record per (view, Gaussian) for the blend kernels' gathers, per-tile ranges instead of 64-bit sort keys).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

namespace dgs {

constexpr int kTile = 16;
constexpr int kSortItems = 16;                    // keys per thread in the depth radix sort
constexpr int kSortTile = 256 * kSortItems;       // keys per workgroup
constexpr int kRangeBuckets = 1024;               // depth range sort: coarse buckets per view (monotone map of the depth bits' range)
constexpr int kRangeWindow = 4096;                //                   a workgroup of the final LDS sort takes the buckets that START inside one window of
                                                  //                   this many ranks: at most kRangeWindow + the largest bucket keys, whatever the depths' distribution
constexpr int kRangeSlots = 16;                   //                   copies of a view's {max key, max ~key} words the preprocess workgroups spread their atomics over
constexpr int kRangeItems = 8;                    //                   keys per thread of its count / scatter kernels (2,048 per workgroup)
constexpr int kRangeTile = 256 * kRangeItems;
constexpr size_t kAlign = 256;

struct Carver {
    char* base;
    size_t off;
    explicit Carver(void* b) : base(static_cast<char*>(b)), off(0) {}
    template <class T>
    T* take(size_t count) {
        off = (off + kAlign - 1) & ~(kAlign - 1);
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
    size_t bytes() const { return ((off + kAlign - 1) & ~(kAlign - 1)) + kAlign; }
};

__host__ __device__ inline int sort_blocks(int P) { return (P + kSortTile - 1) / kSortTile; }
// words per view of GeomState::range_ws: [0, B] bucket counts -> exclusive starts (entry B = keys that are not culled),
// [B + 1, 2 B] scatter cursors, [2 B + 1, 2 B + 1 + blocks) culled keys per 2,048-key block -> their first rank, then windows + 1 words:
// the first rank of every window's segment (the last: the keys that are not culled)
__host__ __device__ inline int range_blocks(int P) { return (P + kRangeTile - 1) / kRangeTile; }
__host__ __device__ inline int range_windows(int P) { return (P + kRangeWindow - 1) / kRangeWindow; }
__host__ __device__ inline size_t range_ws_stride(int P) { return (size_t)((2 * kRangeBuckets + 1 + range_blocks(P) + range_windows(P) + 1 + 3) & ~3); }

// Everything the blend kernels need of one (view, Gaussian), in ONE 64-byte line.  The per-tile lists are depth ordered, so consecutive
// entries are unrelated Gaussians: every field read is its own memory request.  As three arrays (8 + 16 + 16 bytes) a staged entry cost
// three 64-byte requests (measured, round 4: 198 bytes fetched per walked entry, blend_forward 537 MB per 4 views in the trained-like
// regime -- 2.3 TB/s of a ~3.5 TB/s random-request ceiling, profiles/pmc_traffic.json `fetch_calibration.records`); as one aligned
// record it is one.
struct alignas(64) BlendRecord {
    float4 co;                     // (conic.x, conic.y, conic.z, opacity)
    float4 rc;                     // (r, g, b, alpha-skip threshold on `power`)
    float2 xy;                     // pixel coordinates (also in means2D, which the binning kernels stream)
    float2 unused;
};
static_assert(sizeof(BlendRecord) == 64, "one record, one line");

struct GeomState {                 // arrays indexed [view * P + gaussian]
    float* depths;                 // p_view.z
    float2* means2D;               // pixel coordinates
    BlendRecord* blend;            // conic + opacity, colour + cut-off, pixel coordinates
    float* cov3D;                  // 6 per Gaussian: written only by a `debug` forward (inspection); nothing reads it
    uint8_t* clamped;              // bit c set: colour channel c was clamped at 0
    int32_t* internal_radii;
    uint32_t* tiles_touched;
    uint32_t* keys[2];             // depth radix sort ping-pong (key = depth bits, 0xFFFFFFFF if culled)
    uint32_t* vals[2];
    uint32_t* rank_of;             // position of the Gaussian in its view's depth order
    uint32_t* radix_hist;          // [V][blocks][256]
    uint32_t* radix_base;          // [V][256]
    uint32_t* range_ws;            // [V][range_ws_stride(P)] depth range sort (raster_forward.hip range_*_kernel)
    float* grad_acc;               // [V*P*16] backward only: the nine sums of the blend backward per (view, Gaussian) in ONE 64-byte line
                                   //          {colour r g b, mean2D x y, conic xx xy yy, opacity, 7 unused} (raster_backward.hip)
    static GeomState carve(void* buf, size_t P, size_t V, size_t* bytes) {
        Carver c(buf);
        GeomState g;
        const size_t n = P * V;
        g.depths = c.take<float>(n);
        g.means2D = c.take<float2>(n);
        g.blend = c.take<BlendRecord>(n);
        g.cov3D = c.take<float>(6 * n);
        g.clamped = c.take<uint8_t>(n);
        g.internal_radii = c.take<int32_t>(n);
        g.tiles_touched = c.take<uint32_t>(n);
        for (int i = 0; i < 2; ++i) { g.keys[i] = c.take<uint32_t>(n); g.vals[i] = c.take<uint32_t>(n); }
        g.rank_of = c.take<uint32_t>(n);
        g.radix_hist = c.take<uint32_t>(V * (size_t)sort_blocks((int)P) * 256);
        g.radix_base = c.take<uint32_t>(V * 256);
        g.range_ws = c.take<uint32_t>(V * range_ws_stride((int)P));
        g.grad_acc = c.take<float>(16 * n);
        if (bytes) *bytes = c.bytes();
        return g;
    }
};

struct ImageState {
    float* final_T;                // [V*H*W]
    uint32_t* n_contrib;           // [V*H*W]
    uint32_t* tile_count;          // [V*T]   instances per tile
    uint32_t* tile_cursor;         // [V*T]   scatter cursors
    uint2* ranges;                 // [V*T]   [start,end) into the packed instance list
    int32_t* totals;               // [4]     {num_rendered, status, longest tile list, -}
    uint32_t* depth_range;         // [2][V][kRangeSlots]  max depth key | max of ~(depth key) over the view's keys that are not culled, in kRangeSlots partial
                                   //          copies (preprocess_kernel: workgroup b adds to copy b % kRangeSlots; zero = none)
    uint32_t* tile_order;          // [V*T]   launch order of the per-tile kernels (workgroup b works on tile tile_order[b])
    uint32_t* tile_work;           // [V*T]   list entries the forward blend walked before the tile was finished
    uint32_t* tile_scanned;        // [V*T]   depth ranks the forward blend tested for the tile (scan form; else 0)
    uint4* tile_stats;             // [2][V*T] measurement (forward, backward), tools' / emulator build only (kRasterStats), else zero:
                                   //          {cell-list entries walked (x 16 pixels = pair evaluations), wave loop trips (x 64 lanes = lane
                                   //          slots issued), depth ranks scanned (scan form), batches}
    // words the forward zeroes in one fill: tile_count .. depth_range
    size_t zero_span(size_t V) const { return (size_t)(depth_range + 2 * V * kRangeSlots - tile_count); }
    static ImageState carve(void* buf, size_t W, size_t H, size_t V, size_t* bytes) {
        Carver c(buf);
        ImageState s;
        const size_t T = ((W + kTile - 1) / kTile) * ((H + kTile - 1) / kTile);
        s.final_T = c.take<float>(V * W * H);
        s.n_contrib = c.take<uint32_t>(V * W * H);
        s.tile_count = c.take<uint32_t>(V * T);
        s.totals = c.take<int32_t>(4);             // directly behind tile_count: the forward zeroes both with one fill
        s.depth_range = c.take<uint32_t>(2 * V * kRangeSlots);     // ... and these (zero_span() words from tile_count on)
        s.tile_cursor = c.take<uint32_t>(V * T);
        s.ranges = c.take<uint2>(V * T);
        s.tile_order = c.take<uint32_t>(V * T);
        s.tile_work = c.take<uint32_t>(V * T);
        s.tile_scanned = c.take<uint32_t>(V * T);
        s.tile_stats = c.take<uint4>(2 * V * T);
        if (bytes) *bytes = c.bytes();
        return s;
    }
};

struct BinningState {
    uint64_t* inst_key;            // [N] unsorted, grouped by tile: (depth bits << 32) | Gaussian index   (bitonic form)
    uint32_t* inst_rank;           //     the same storage as 32-bit depth ranks                           (rank-sort form)
    uint32_t* point_list;          // [N] per tile, front to back: Gaussian index
    static BinningState carve(void* buf, size_t N, size_t* bytes) {
        Carver c(buf);
        BinningState b;
        b.inst_key = c.take<uint64_t>(N);
        b.inst_rank = reinterpret_cast<uint32_t*>(b.inst_key);
        b.point_list = c.take<uint32_t>(N);
        if (bytes) *bytes = c.bytes();
        return b;
    }
};

}  // namespace dgs
