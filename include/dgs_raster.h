/* ============================================================================
 * dgs_raster.h -- C ABI of the MI355X-native differentiable 3D-Gaussian rasterizer.
 *
 * Drop-in boundary for the reference's diff-gaussian-rasterization submodule: each
 * entry point replaces one function the reference's pybind module `_C` exports
 * (paths relative to /root/reference/submodules/diff-gaussian-rasterization/):
 *
 *   dgs_raster_forward   <- RasterizeGaussiansCUDA          rasterize_points.cu:35-115
 *                           / CudaRasterizer::Rasterizer::forward   cuda_rasterizer/rasterizer_impl.cu:198-336
 *   dgs_raster_backward  <- RasterizeGaussiansBackwardCUDA  rasterize_points.cu:117-196
 *                           / CudaRasterizer::Rasterizer::backward  cuda_rasterizer/rasterizer_impl.cu:340-434
 *   dgs_mark_visible     <- markVisible                     rasterize_points.cu:198-217
 *   dgs_alloc_fn         <- std::function<char*(size_t)> resizeFunctional   rasterize_points.cu:27-33
 *
 * Plain C: raw device pointers, sizes, a HIP stream; no torch / pybind types.
 * All pointers are DEVICE pointers unless a field says "host".  NULL == absent
 * optional input (the reference uses empty tensors, rasterizer_impl.cu:321,389,411).
 * Return value: DGS_OK (0) or a negative DgsStatus; nothing throws across the ABI.
 *
 * Extension over the reference (MI355X-first): one call renders V views of S Gaussian
 * sets (`V > 1`), so a whole (batch x views) step is ONE launch sequence instead of
 * b*v Python-level calls; V == 1 / S == 1 is the verbatim drop-in case.
 * ==========================================================================*/
#ifndef DGS_RASTER_H_
#define DGS_RASTER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGS_ABI_VERSION 8   /* bumped with every change of a struct or prototype in the include directory (round 2 = 2, unversioned) */
#define DGS_TILE 16 /* cuda_rasterizer/config.h:14-15 */

typedef void* dgs_stream_t; /* hipStream_t */

typedef enum DgsStatus {
    DGS_OK = 0,
    DGS_ERR_INVALID_ARGUMENT = -1,  /* AT_ERROR shape checks, rasterize_points.cu:57-59 */
    DGS_ERR_NEED_COLORS = -2,       /* neither SH nor precomputed colours            */
    DGS_ERR_NEED_COVARIANCE = -3,   /* neither scale/rotation nor precomputed cov3D   */
    DGS_ERR_ALLOC = -4,             /* allocation callback returned NULL / capacity too small */
    DGS_ERR_DEVICE = -5,            /* hipGetLastError() != hipSuccess (debug=1 synchronises first, auxiliary.h:166-173) */
    DGS_ERR_PREFILTERED_CULLED = -6,/* prefiltered=1 but a point was culled (auxiliary.h:156-160 traps) */
    DGS_ERR_BINNING_OVERFLOW = -7   /* async mode: num_rendered exceeded binning capacity */
} DgsStatus;

/* Grows (or returns) a device buffer of at least `bytes` bytes; `user` is passed through.
 * Same contract as the reference's resizeFunctional lambdas (rasterize_points.cu:27-33). */
typedef void* (*dgs_alloc_fn)(size_t bytes, void* user);

typedef struct DgsRasterForwardArgs {
    /* ---- sizes ---- */
    int32_t P;        /* Gaussians per set                                            */
    int32_t D;        /* active SH degree (0..3)                                      */
    int32_t M;        /* SH coefficients stored per Gaussian ((deg_max+1)^2), 0 if no SH */
    int32_t width, height;
    int32_t V;        /* views rendered by this call (>= 1)                           */
    int32_t views_per_set; /* view v reads Gaussian set v / views_per_set; S = ceil(V / views_per_set) */
    /* ---- inputs, layouts as in the reference binding ---- */
    const float* background;     /* [3]                                                */
    const float* means3D;        /* [S,P,3]                                            */
    const float* shs;            /* [S,P,M,3] or NULL                                  */
    const float* colors_precomp; /* [S,P,3]   or NULL                                  */
    const float* opacities;      /* [S,P]                                              */
    const float* scales;         /* [S,P,3]   or NULL                                  */
    const float* rotations;      /* [S,P,4] (r,x,y,z) or NULL                          */
    const float* cov3D_precomp;  /* [S,P,6]   or NULL                                  */
    const float* viewmatrix;     /* [V,16] column-major W2C  (tensor = W2C^T)          */
    const float* projmatrix;     /* [V,16] column-major P*W2C                          */
    const float* campos;         /* [V,3]                                              */
    const float* tanfov;         /* [V,2] (tanfovx,tanfovy) on device, or NULL -> use the two scalars below for every view */
    float tanfovx, tanfovy;
    float scale_modifier;
    int32_t prefiltered;
    int32_t debug;               /* 1: synchronise + check after every stage           */
    /* DiffusionGS fusion (MI355X-first, off for the drop-in path): 1 = `scales`, `rotations`,
     * `opacities` hold RAW parameters and the kernel applies exp / normalize / sigmoid
     * (gs_core.py:330-334,544-570) itself.                                              */
    int32_t raw_activations;
    /* ---- outputs ---- */
    float* out_color;            /* [V,3,H,W], written for every pixel                 */
    int32_t* radii;              /* [V,P]                                              */
    /* ---- state buffers (opaque to the caller; consumed by dgs_raster_backward) ---- */
    dgs_alloc_fn geom_alloc;  void* geom_user;     /* size: dgs_raster_geom_bytes(P,V)          */
    dgs_alloc_fn img_alloc;   void* img_user;      /* size: dgs_raster_image_bytes(W,H,V)       */
    dgs_alloc_fn binning_alloc; void* binning_user;/* size: dgs_raster_binning_bytes(num_rendered) */
    /* async mode: if binning_capacity > 0, binning_alloc is called ONCE up front with
     * dgs_raster_binning_bytes(binning_capacity) and the call never synchronises, reads nothing back and may be captured in a
     * hipGraph (the reference's blocking read of num_rendered, rasterizer_impl.cu:281, is gone); more instances than the
     * capacity set status_dev[1] = DGS_ERR_BINNING_OVERFLOW and the image is filled with NaN (nothing plausible is rendered).
     * The backward of such a call takes binning_capacity as its `num_rendered`.                                              */
    int64_t binning_capacity;
    int32_t* num_rendered_dev;   /* device int32[4], written by the call in both modes when not NULL: [0] = num_rendered (low 32 bits),
                                    [1] = status (DgsStatus), [2] = longest tile list, [3] = 1 (written last: "the other three are valid") */
    int32_t* num_rendered_host;  /* DEVICE-ACCESSIBLE host int32[4] or NULL: the same four words, STORED BY A KERNEL of the call through this
                                    pointer (no memcpy: a copy node in a captured sequence proved unreliable on ROCm 7.2) -- so it must be
                                    page-locked memory that is mapped into the device's address space at this very address and coherent
                                    (hipHostMalloc with hipHostMallocMapped | hipHostMallocCoherent, or torch's pin_memory=True); plain
                                    pageable or un-mapped page-locked memory faults.  Valid once word [3] reads non-zero (clear it
                                    before the call; the kernel writes it last, behind a system-scope fence -- the call's second kernel,
                                    long before its blend), or once the stream has passed the call                                    */
    int64_t longest_hint;        /* async mode, IN: the longest tile list the caller expects (a previous call of this shape), 0 =
                                    unknown.  Sizes the LDS of the per-tile sort (form 3 named: that kernel is launched alone); a
                                    longer list is sorted in LDS-sized chunks merged by rank -- slower, never an error             */
    /* ---- result ---- */
    int64_t num_rendered;        /* host, OUT (sync mode); -1 in async mode             */
    int64_t longest_list;        /* host, OUT (sync mode); -1 in async mode             */
    int32_t binning_form;        /* per-tile ordering algorithm. 0: chosen from the instance statistics (on the host in the sync mode;
                                    in the async mode every form is launched and the device returns from the others at once);
                                    1 instance list + depth-rank bitmap sort, 2 per-tile scan of the depth-ordered Gaussians,
                                    3 instance list + per-tile sort in LDS (falls back to 1 when a list does not fit): that form
                                    only is launched -- what an async caller passes from dgs_raster_binning_form() of the
                                    previous call's statistics.  All forms produce the reference's lists bit for bit.  */
    int32_t exact_exp;           /* exponential of a (pixel, Gaussian) pair in the blend loop (the reference: CUDA's <= 2 ulp expf,
                                    forward.cu:332-358 / backward.cu:463-532; its build passes no fast-math flag).  0 (default):
                                    v_exp_f32 with the rounding error of its argument compensated (<= 1.3 ulp measured), and the
                                    oracle's sequence for a pair whose alpha falls within 1e-6 of the 1/255 cut-off, so both put
                                    every pair on the same side of it (csrc/dgs_device.h blend_exp; gradients within 1e-5 of the
                                    oracle's); 1: a fixed IEEE sequence the CPU oracle restates (oracle exp_mode 1), every float
                                    of the result bit-identical with the oracle.  Integer artefacts that do not depend on alpha
                                    (radii, tile lists, ranges, sort order) are identical in both; pass the same value to the
                                    backward.  */
} DgsRasterForwardArgs;

typedef struct DgsRasterBackwardArgs {
    int32_t P, D, M, width, height, V, views_per_set;
    int64_t num_rendered;        /* R, as returned by forward                           */
    const float* background;
    const float* means3D;
    const float* shs;
    const float* colors_precomp;
    const float* opacities;      /* [S,P] raw opacities; only read when raw_activations=1 */
    const float* scales;
    const float* rotations;
    const float* cov3D_precomp;
    const float* viewmatrix;
    const float* projmatrix;
    const float* campos;
    const float* tanfov;
    float tanfovx, tanfovy;
    float scale_modifier;
    int32_t debug;
    int32_t raw_activations;
    const int32_t* radii;        /* [V,P]                                               */
    const float* dL_dpix;        /* [V,3,H,W]                                           */
    const void* geom_buffer;
    const void* binning_buffer;
    const void* img_buffer;
    /* gradient outputs: every element is WRITTEN by the call (the caller need not pre-fill anything; the reference's torch::zeros,
     * rasterize_points.cu:148-156).  What the blend backward accumulates lives in the geometry buffer (a 64-byte record per
     * (view, Gaussian), zero-filled by the call); the per-view tensors below are written from it.
     * With V > 1 gradients of the views of one set are SUMMED into that set's slot, in view order.   */
    float* dL_dmeans2D;   /* [V,P,3]  (per view, like the reference's per-call tensor)   */
    float* dL_dconic;     /* [V,P,4]  scratch ([P,2,2] in the reference), never returned to Python */
    float* dL_dcolors;    /* SH given: [V,P,3] per-view scratch (consumed by the SH backward); colours precomputed: [S,P,3],
                             summed over the views of each set (this IS the gradient returned to Python)             */
    float* dL_dcov3D;     /* [V,P,6]  */
    float* dL_dopacity;   /* [S,P]    */
    float* dL_dmeans3D;   /* [S,P,3]  */
    float* dL_dsh;        /* [S,P,M,3] or NULL */
    float* dL_dscales;    /* [S,P,3]  or NULL */
    float* dL_drotations; /* [S,P,4]  or NULL */
    int32_t exact_exp;    /* as DgsRasterForwardArgs.exact_exp: must equal the forward's */
    /* Deterministic form (opt-in: dgs_amd/raster.py `deterministic`): `scratch` = dgs_raster_backward_scratch_bytes(...) bytes of device
     * memory (contents irrelevant, nothing is read before it is written).  Every (tile, Gaussian) instance then has a slot of its
     * own, a tile STORES its sums there and a gather adds a Gaussian's slots in a fixed order: no floating-point atomic anywhere,
     * the same bits on every run.  NULL: the sums of a Gaussian's tiles meet in fp32 atomics (the reference's way, backward.cu:
     * 10 atomicAdd per pair; here one instruction per (tile, Gaussian) into the record) -- run-to-run differences of the order of
     * 1e-7 relative.                                                                                                            */
    void* scratch;
    size_t scratch_bytes;
} DgsRasterBackwardArgs;

int dgs_abi_version(void);
const char* dgs_status_string(int status);

size_t dgs_raster_geom_bytes(int32_t P, int32_t V);
size_t dgs_raster_image_bytes(int32_t width, int32_t height, int32_t V);
size_t dgs_raster_binning_bytes(int64_t num_rendered);

int dgs_raster_forward(DgsRasterForwardArgs* args, dgs_stream_t stream);
/* The ordering form a call with these instance statistics picks (1..3, see DgsRasterForwardArgs.binning_form; `binning_form` 0 =
 * no preference): the host-side twin of the choice the kernels make in the async mode. */
int dgs_raster_binning_form(int32_t binning_form, int64_t num_rendered, int64_t longest_list, int32_t P, int32_t width, int32_t height,
                            int32_t V);
int dgs_raster_backward(const DgsRasterBackwardArgs* args, dgs_stream_t stream);
/* Bytes of DgsRasterBackwardArgs.scratch: 36 per instance slot (num_rendered as passed to the backward: the forward's count, or the
 * binning capacity of an asynchronous forward) + 20 per (view, Gaussian) + 8 per tile. */
size_t dgs_raster_backward_scratch_bytes(int32_t P, int32_t width, int32_t height, int32_t V, int64_t num_rendered);
int dgs_mark_visible(int32_t P, const float* means3D, const float* viewmatrix, const float* projmatrix,
                     uint8_t* present, dgs_stream_t stream);

/* All cameras of a step in one launch; replaces the reference's per-view `Camera` module
 * (diffusionGS/models/gsrenderer/gs_core.py:277-316).  c2w [n,16] row-major OpenCV camera-to-world, fxfycxcy [n,4]
 * pixels -> viewmatrix [n,16] (= W2C^T), projmatrix [n,16] (= W2C^T P^T), campos [n,3], tanfov [n,2]; all device. */
int dgs_cameras_from_c2w(int32_t n, const float* c2w, const float* fxfycxcy, int32_t height, int32_t width,
                         float znear, float zfar, float* viewmatrix, float* projmatrix, float* campos, float* tanfov,
                         dgs_stream_t stream);

/* Per-pixel rays of n cameras in one launch; replaces TransformInput (diffusionGS/systems/utils.py:621-684,751-757), the step
 * immediately before the denoiser.  c2w [n,16], fxfycxcy [n,4] -> ray_o, ray_d [n,3,H,W] (ray_d unit length). */
int dgs_rays_from_c2w(int32_t n, const float* c2w, const float* fxfycxcy, int32_t height, int32_t width, float* ray_o,
                      float* ray_d, dgs_stream_t stream);

/* Introspection for the parity tests: copies a named array of the forward state out of the
 * opaque buffers into `dst` (device pointer, `dst_bytes` capacity).  Names: "depths", "means2D",
 * "conic_opacity", "rgb", "tiles_touched", "clamped", "cov3D", "ranges", "n_contrib", "final_T",
 * "point_list", "list_len", "tile_work", "tile_scanned" (and "tile_stats", "tile_stats_bwd": zero in the product library).
 * "cov3D" is kept only by a forward with `debug` set (nothing on the product path reads it: the backward recomputes it).
 * Returns the number of bytes written or a negative DgsStatus.            */
int64_t dgs_raster_state_read(const char* name, int32_t P, int32_t width, int32_t height, int32_t V,
                              int64_t num_rendered, const void* geom_buffer, const void* binning_buffer,
                              const void* img_buffer, void* dst, int64_t dst_bytes, dgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DGS_RASTER_H_ */
