// dit_kernels.h -- parameter blocks and launchers shared by the DiT translation units (internal, not part of the C ABI).
#pragma once
#include "dit_common.h"
#include "dgs_dit.h"

namespace dgs {

struct LnParams {
    int rows, width, mod_stride, rows_per_batch, out_f32;
    float eps;
    const float *x, *weight, *shift, *scale;
    void* out;
};

struct RowLinParams {
    int M, N, K, silu_in, silu_out;
    const float* x;
    const bf16_t* W;
    const float* bias;
    float* out;
};

struct EmbedParams {
    int B, V, H, W, ps, lpad, relative_plk;
    const float *images, *ray_o, *ray_d;
    bf16_t* out;   // [B*lpad, 9*ps*ps]
};

struct GsParams {
    int B, V, H, W, ps, lpad, ng, C, scene, relative_plk;
    float range_near, range_far;
    const float *dec, *up, *ray_o, *ray_d;
    float *xyz, *features, *scaling, *rotation, *opacity, *aligned;
};

// Second role of a LayerNorm + modulate launch: the consumer GEMM's output rows [row0, row0 + nrows) of every sample (nrows <= 2: the
// rows behind the sample's last full 256-row tile), see layernorm_rows_gemv_kernel.  `items` is set by the launcher.
struct LnRowsGemv {
    const bf16_t* W;           // the GEMM's weight [N, ldw], K = the LayerNorm's width
    const float* bias;
    void* out;
    void* aux;
    bf16_t* vt;
    int N, ldw, ldo, epilogue, row0, nrows, items;
    float q_scale;
};

int launch_layernorm(const DgsDitLayerNormArgs* a, hipStream_t st);
// the same launch + the rows of `g` (an inference GEMM the caller launches next with launch_gemm_external_rows); false: shape not taken
bool layernorm_rows_gemv_ok(const DgsDitLayerNormArgs* a, const DgsDitGemmArgs* g);
int launch_layernorm_rows_gemv(const DgsDitLayerNormArgs* a, const DgsDitGemmArgs* g, hipStream_t st);
// dgs_dit_gemm without the one or two live rows behind every sample's last full 256-row tile (somebody else writes those outputs);
// gemm_leaves_rows_out: whether this shape runs on the kernel that can (the sliced 256-row kernel with two-row GEMV side jobs)
bool gemm_leaves_rows_out(const DgsDitGemmArgs* a);
int launch_gemm_external_rows(const DgsDitGemmArgs* a, dgs_stream_t stream);
int launch_rowlinear(const DgsDitRowLinearArgs* a, hipStream_t st);
int launch_timestep(const int64_t* t, float* emb, int B, hipStream_t st);
// zero `bytes` bytes (a multiple of 16, 16-byte aligned) with a kernel of the library's own.  NOT hipMemsetAsync: inside a captured
// hipGraph a memset NODE stopped zeroing its destination as soon as the process issued an eager hipMemsetAsync elsewhere (ROCm 7.2,
// profiles/r04_graph_memset_node_debug.txt), and a plain kernel is also the shorter launch (2 us against ~5 for the runtime's fill)
int launch_zero_fill(void* dst, size_t bytes, hipStream_t st);
int launch_embed(const EmbedParams& p, hipStream_t st);
int launch_pos_embed(const float* pe, float* x, int B, int lpad, int L, int ng, int width, hipStream_t st);
int launch_gather_tokens(const float* x, float* out, int B, int lpad, int L, int ng, int width, hipStream_t st);
int launch_scatter_tokens(const float* in, float* x, int B, int lpad, int L, int ng, int width, hipStream_t st);
int launch_gaussians(const GsParams& p, hipStream_t st);

// ---- backward (dit_backward_elementwise.hip) ----
struct LnBwdParams {
    int rows, width, mod_stride, rows_per_batch, rows_per_block, dh_f32;
    float eps;
    const float* x;
    const void* dh;          // bf16 (or f32) [rows, width]
    const float *weight, *scale;
    const float* dx_in;      // optional residual-path gradient added to the result
    float* dx_out;
    float *dshift, *dscale;  // [batch, mod_stride], may be NULL: which column sums are wanted (and, without `part`, where they are added)
    float* dweight;          // [width], may be NULL
    float* part;             // slab [workgroups][part_stride >= 3 width]: row = [shift | scale | weight] partial sums of one workgroup
    int part_stride;         //   (col_reduce finishes them; NULL is only valid for single-workgroup launches, which ADD to the above)
};

struct RowLinBwdParams {
    int M, N, K, silu_in;
    const float* x;       // [M, K] pre-activation input
    const bf16_t* W;      // [N, K]
    const float* dy;      // [M, N], row stride ldy (0: N)
    int ldy;
    float* dW;            // [N, K] or NULL
    float* db;            // [N] or NULL
    float* dx;            // [M, K] or NULL (with `part`: only says that dx is wanted)
    float* part;          // slab [workgroups][M * K] partial dx rows, or NULL (single workgroup: ADDED to dx)
    int rows_per_block;   // output features per workgroup (0: ROWLINEAR_BWD_ROWS)
};

// One column-sum job of col_reduce_kernel: out[b * out_bstride + c] = sum_{s < slots} part[(b * slots + s) * part_stride + c]
struct ColReduceJob {
    const float* part;
    float* out;
    int slots, part_stride, cols, batches, out_bstride;
};
struct ColReduceParams { ColReduceJob job[8]; };

constexpr int ROWLINEAR_BWD_ROWS = 256;     // output features per workgroup of rowlinear_backward_kernel (default; the slab of dx partial rows is sized by it)
constexpr int ROWLINEAR_DW_ROWS = 32;       // ... of a launch that only writes dW / db (a block's adaLN Linear, 6W rows: 192 workgroups at W = 1024)
// Rows per workgroup of layernorm_backward_kernel (never straddles samples; the column-sum slab has one row per workgroup).  The
// kernel runs one 512-thread workgroup per CU (160 VGPRs): with 32 rows each, the training shape (4 x 4224 rows) was 528 workgroups
// on 256 CUs -- two full rounds and a third for 16 of them, 83 us at 2.9 TB/s.  The smallest divisor of the sample's rows that
// gets every row into ONE round (66 there: 256 workgroups) takes a third off.  `ncu` 0 (unknown): 32 rows as before.
inline int ln_backward_rows_per_block(int rows_per_batch, int rows, int ncu) {
    if (ncu > 0 && rows > 0) {
        const int want = (rows + ncu - 1) / ncu < 16 ? 16 : (rows + ncu - 1) / ncu;
        for (int d = want; d <= rows_per_batch && d <= 8 * want; ++d)
            if (rows_per_batch % d == 0) return d;
    }
    return rows_per_batch % 32 == 0 ? 32 : rows_per_batch;
}
int ln_backward_compute_units();           // cached device query (dit_backward_elementwise.hip)

struct GsBwdParams {
    int B, V, H, W, ps, lpad, ng, C, scene, relative_plk;
    float range_near, range_far;
    const float *dec, *up, *ray_d;                                   // forward values needed for the activation masks
    const float *dxyz, *dfeatures, *dscaling, *drotation, *dopacity; // incoming gradients [B, P, ...]
    bf16_t* ddec;
    float* dup;
};

int launch_transpose(const bf16_t* in, int ld, bf16_t* out, int B, int rows, int F, hipStream_t st);
// part: [B * rows / 64][2 W] (gate gradient | bias gradient partial sums)
int launch_gate_mul(const float* dx, const bf16_t* y, const float* gate, int gate_stride, bf16_t* dy, bf16_t* dyT, float* part, int B,
                    int rows, int W, hipStream_t st);
int launch_col_reduce(const ColReduceJob* jobs, int njobs, hipStream_t st);
int colsum_slots(int M, int N, int ld);
int launch_layernorm_backward(const LnBwdParams& p, hipStream_t st);
int launch_colsum(const bf16_t* dy, int ld, int M, int N, float* part, hipStream_t st);
int launch_rowlinear_backward(const RowLinBwdParams& p, hipStream_t st);
int launch_gaussians_backward(const GsBwdParams& p, hipStream_t st);

}  // namespace dgs
