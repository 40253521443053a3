/* ============================================================================
 * dgs_dit.h -- C ABI of the MI355X-native DiffusionGS denoiser (DiT -> per-pixel Gaussians).
 *
 * The reference implements this half of the hot path in Python/PyTorch only
 * (paths relative to /root/reference/diffusionGS/):
 *
 *   dgs_dit_forward        <- DGSDenoiser.image_to_gaussians     models/denoiser/denoiser.py:306-416
 *                             (scene variant                      models/denoiser/denoiser_scene.py:292-420)
 *   dgs_dit_forward_train / dgs_dit_backward  <- the same function under torch autograd with per-block
 *                             torch.utils.checkpoint (denoiser.py:348-354,441-447); here nothing is recomputed
 *   dgs_dit_gemm           <- nn.Linear inside timm Attention/Mlp models/transformers/utils_transformer.py:254-265
 *                             (+ gated residual :286-289, GELU-tanh :259, tokenizer denoiser.py:216-221,
 *                              decoder head denoiser.py:148-164)
 *   dgs_dit_attention      <- F.scaled_dot_product_attention in timm==0.9.16 Attention.forward
 *   dgs_dit_layernorm      <- nn.LayerNorm + modulate()           utils_transformer.py:26-27,271-290; denoiser.py:21-22
 *   dgs_dit_layernorm_gemm <- modulate(norm1(x)) -> attn.qkv and modulate(norm2(x)) -> mlp.fc1 as DiTBlock.forward pairs them
 *                                                                 utils_transformer.py:271-290
 *   dgs_dit_rowlinear      <- adaLN_modulation / TimestepEmbedder / upsampler Linear on a handful of rows
 *                                                                 utils_transformer.py:266-269; denoiser.py:26-72,122-136
 *   dgs_dit_embed          <- ray/Plucker embedding + patchify    denoiser.py:312-334,210-215
 *   dgs_dit_gaussians      <- GaussiansUpsampler.to_gs + hard pixel alignment   denoiser.py:103-120,370-413
 *
 * Plain C: raw device pointers + sizes + a HIP stream.  bf16 tensors are passed as uint16_t*.
 * Internal token layout ("padded rows"): every sample owns `lpad` consecutive rows (lpad % 256 == 0,
 * lpad >= L); rows [0, L-n_g) are the image tokens in the reference's (v, hh, ww) order, rows
 * [L-n_g, L) are the n_g learned Gaussian tokens (the reference puts them FIRST; every operator of the
 * block is permutation-equivariant over tokens, so only the final gather restores the order), rows
 * [L, lpad) are padding that is computed but never observed (attention masks keys >= L).
 * Return value: DGS_OK or a negative DgsStatus (dgs_raster.h).
 * ==========================================================================*/
#ifndef DGS_DIT_H_
#define DGS_DIT_H_

#include <stddef.h>
#include <stdint.h>

#include "dgs_raster.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum DgsGemmEpilogue {
    DGS_EPI_BF16 = 0,          /* out_bf16[m,n] = acc + bias[n]                                       */
    DGS_EPI_GELU_BF16 = 1,     /* out_bf16[m,n] = gelu_tanh(acc + bias[n])                            */
    DGS_EPI_GATE_RESIDUAL = 2, /* out_f32[m,n] += gate[m / rows_per_batch, n] * (acc + bias[n])       */
    DGS_EPI_F32 = 3,           /* out_f32[m,n]  = acc + bias[n]                                       */
    DGS_EPI_QKV = 4,           /* n < 2N/3: out_bf16[m,n] (ldo = 2N/3);  n >= 2N/3: V^T: vt[(b*N/3 + n-2N/3)*lpad + t],
                                  b = m / lpad, t = m % lpad  (rows_per_batch = lpad)                 */
    DGS_EPI_DGELU_BF16 = 5     /* out_bf16[m,n] = (acc + bias[n]) * gelu_tanh'(aux_bf16[m,n])   (MLP backward) */
    /* BF16 / GELU_BF16 / DGELU_BF16 additionally write a transposed bf16 copy [batch, N, rows_per_batch] when `vt` is set;
     * GELU_BF16 / GATE_RESIDUAL additionally write the pre-activation / pre-gate value to `aux` (bf16 [M, ldo]) when set. */
} DgsGemmEpilogue;

typedef enum DgsGemmAlgo {
    DGS_GEMM_AUTO = 0,
    DGS_GEMM_SIMPLE128 = 1,    /* 128 x 128|64 tiles, two LDS stages, 2 workgroups / CU (AUTO: the N = 1024 GEMMs at 1 sample) */
    DGS_GEMM_SLICED = 4,       /* 256 x 256|128 x 32 tiles, 4-stage LDS-DMA ring, explicit MFMA / LDS issue slices (AUTO: QKV and
                                  fc1 at 1 sample, every eligible shape above 8192 rows)                               */
    DGS_GEMM_QUAD = 5,         /* the same with 4 waves of 128 x 128 (256 x 256 tiles only; else as SLICED)              */
    DGS_GEMM_SLICED128 = 6     /* the same ring and schedule on 128 x 128 tiles, 4 waves of 64 x 64 (AUTO: shapes with few tiles, e.g.
                                  the N = 1024 GEMMs at one sample: 256 tiles, one per CU)                                */
} DgsGemmAlgo;

typedef struct DgsDitGemmArgs {
    int32_t M, N, K;           /* M % 128 == 0, N % 128 == 0, K % 64 == 0                             */
    const uint16_t* A;         /* bf16 [M, lda]                                                       */
    int32_t lda;
    const uint16_t* W;         /* bf16 [N, ldw]  (nn.Linear weight layout: out_features x in_features) */
    int32_t ldw;
    const float* bias;         /* [N] or NULL                                                         */
    int32_t epilogue;          /* DgsGemmEpilogue                                                     */
    void* out;                 /* bf16 or f32 [M, ldo]                                                */
    int32_t ldo;
    const float* gate;         /* [batch, gate_stride] (DGS_EPI_GATE_RESIDUAL)                        */
    int32_t gate_stride;
    int32_t rows_per_batch;
    uint16_t* vt;              /* DGS_EPI_QKV: V^T bf16 [batch, N/3, rows_per_batch]                   */
    const float* resid;        /* DGS_EPI_GATE_RESIDUAL: residual input [M, ldo] f32; NULL -> `out` (in place)             */
    void* aux;                 /* bf16 [M, ldo]: see DgsGemmEpilogue                                                  */
    int32_t k_per_batch;       /* 0 -> K.  Otherwise the reduction runs over K / k_per_batch samples whose operand rows
                                  are a_batch_stride / w_batch_stride elements apart (weight gradients: reduction over the
                                  tokens of [batch, features, lpad] transposed activations)                          */
    int64_t a_batch_stride, w_batch_stride;
    int32_t algo;              /* DgsGemmAlgo; 0 = automatic                                                           */
    int32_t valid_rows;        /* 0 or rows_per_batch: every row is computed.  Otherwise rows [valid_rows, rows_per_batch)
                                  of every sample are padding: 32-row blocks made only of padding are neither computed
                                  nor stored (their output rows keep their previous contents).                       */
    float q_scale;             /* DGS_EPI_QKV: the q features (n < N/3) are multiplied by q_scale before the bf16 rounding
                                  (0 -> 1).  The denoiser passes scale * log2(e) so that the attention kernel gets its
                                  pre-scaled queries with a single rounding (DgsDitAttentionArgs.q_prescaled).        */
    float* splitk_ws;          /* optional f32 scratch of dgs_dit_gemm_splitk_bytes(M, N, K, k_per_batch) bytes.  When set,
                                  DGS_EPI_F32 without bias and that size is non-zero, the reduction is split over
                                  workgroups (256 x 256 tiles, one partial product per split, summed by a second kernel
                                  into `out`): weight gradients have few output tiles and a very long K.              */
} DgsDitGemmArgs;

/* Bytes of DgsDitGemmArgs.splitk_ws for this shape; 0 when the split-K path does not apply to it. */
size_t dgs_dit_gemm_splitk_bytes(int32_t M, int32_t N, int32_t K, int32_t k_per_batch);

typedef struct DgsDitAttentionArgs {
    int32_t B, heads, L, lpad; /* head dim is 64; L valid tokens per sample, lpad padded rows         */
    const uint16_t* qk;        /* bf16 [B*lpad, 2*heads*64]: q features then k features               */
    const uint16_t* vt;        /* bf16 [B, heads*64, lpad]                                            */
    uint16_t* out;             /* bf16 [B*lpad, heads*64]                                             */
    float scale;               /* 1/sqrt(64)                                                          */
    /* optional generalisations (training keeps q|k|v in ONE [B*lpad, 3W] tensor and its transposed copy):         */
    int32_t ld_qk;             /* row stride of `qk` in elements; 0 -> 2*heads*64                                  */
    int32_t k_offset;          /* elements from a row's q features to its k features; 0 -> heads*64                */
    int64_t vt_batch_stride;   /* elements between samples of `vt`; 0 -> heads*64*lpad                             */
    float* lse2;               /* optional out [B, heads, lpad]: log2-domain log-sum-exp per query (for backward)  */
    int32_t q_prescaled;       /* nonzero: the q features already carry the factor scale * log2(e) (DgsDitGemmArgs.q_scale);
                                  0: the kernel applies it to the bf16 queries itself (one extra bf16 rounding)       */
    void* tail_ws;             /* scratch of dgs_dit_attention_tail_bytes(B, heads, L) bytes, required when L % 32 != 0 (the
                                  DiT's two learned tokens): partial records of the L % 32 tail queries + arrival counters.
                                  Zero-filled ONCE by the caller (the kernel leaves the counters at zero); one per stream
                                  that launches concurrently.                                                          */
    size_t tail_ws_bytes;
    int32_t tail_mode;         /* reserved: 0 (anything else is DGS_ERR_INVALID_ARGUMENT).  The L % 32 tail queries run inside the main
                                  kernel: key-split records + a merge at its end.  (Rounds 4-5 ran them as a launch of their own behind
                                  this field; it lost twice -- profiles/r04_attention_tail_stream_ab.txt, profiles/r05_tail_chain_ab.txt --
                                  and lives on as tools/next/tail_chain_experiment.patch.)                                          */
} DgsDitAttentionArgs;

/* Bytes of DgsDitAttentionArgs.tail_ws for this shape (0 when L % 32 == 0). */
size_t dgs_dit_attention_tail_bytes(int32_t B, int32_t heads, int32_t L);

typedef struct DgsDitAttentionBackwardArgs {
    int32_t B, heads, L, lpad;
    const uint16_t* qkv;       /* bf16 [B*lpad, 3W] row-major: q | k | v                                           */
    const uint16_t* qkvT;      /* bf16 [B, 3W, lpad] token-contiguous copy                                         */
    const uint16_t* o;         /* bf16 [B*lpad, W] attention output of the forward                                 */
    const uint16_t* dO;        /* bf16 [B*lpad, W] its gradient                                                    */
    const uint16_t* dOT;       /* bf16 [B, W, lpad] token-contiguous copy of dO                                    */
    const float* lse2;         /* [B, heads, lpad] from the forward                                                */
    float* D;                  /* [B, heads, lpad] scratch: -rowsum(dO o O) (the initial value of the dP accumulators) */
    uint16_t* dqkv;            /* out bf16 [B*lpad, 3W]: dq | dk | dv                                              */
    float scale;
    uint16_t* dqkvT;           /* optional out bf16 [B, 3W, lpad]: the same gradients token-contiguous (what the qkv weight-gradient
                                  GEMM reads; padding tokens are not written: the caller's buffer holds zeros there) or NULL   */
    float* bias_part;          /* optional out [B * dgs_dit_attention_backward_slots(L)][3W]: per-workgroup column sums of dqkv over
                                  the workgroup's valid tokens (slot = sample * slots + block) -- the qkv bias gradient's partial
                                  rows, to be added in slot order -- or NULL                                                   */
} DgsDitAttentionBackwardArgs;
/* 256-token blocks (+ single tail tokens) per sample the backward's workgroups own: rows per sample of `bias_part`. */
int32_t dgs_dit_attention_backward_slots(int32_t L);

typedef struct DgsDitLayerNormArgs {
    int32_t rows, width;       /* width == 1024 (one wave per row, 16 elements per lane) or any multiple of 64 <= 2048 */
    const float* x;            /* f32 [rows, width]                                                   */
    const float* weight;       /* [width] or NULL                                                     */
    const float* shift;        /* [batch, mod_stride] or NULL -> no modulate                           */
    const float* scale;
    int32_t mod_stride;
    int32_t rows_per_batch;
    float eps;
    void* out;                 /* bf16 or f32 [rows, width]                                           */
    int32_t out_f32;
} DgsDitLayerNormArgs;

/* Backward of dgs_dit_layernorm: h = LN(x) * weight * (1 + scale) + shift.  dx_out = dx_in + dLN; column sums over the
 * rows of each sample are ADDED to dshift / dscale ([batch, mod_stride]) and dweight ([width]) with fp32 atomics. */
typedef struct DgsDitLayerNormBackwardArgs {
    int32_t rows, width;
    const float* x;            /* f32 [rows, width]: the forward's input                               */
    const void* dh;            /* bf16 (dh_f32 = 0) or f32 [rows, width]                                */
    int32_t dh_f32;
    const float* weight;       /* [width] or NULL                                                     */
    const float* scale;        /* [batch, mod_stride] or NULL                                         */
    int32_t mod_stride, rows_per_batch;
    float eps;
    const float* dx_in;        /* optional f32 [rows, width] added to the result (residual path)      */
    float* dx_out;             /* f32 [rows, width]                                                   */
    float *dshift, *dscale, *dweight;   /* optional column sums, WRITTEN: [batch, mod_stride] x 2, [width]  */
    void* scratch;             /* dgs_dit_layernorm_backward_scratch_bytes(rows, width, rows_per_batch) bytes of device memory when any
                                  column sum is requested: per-workgroup partial rows, summed in a fixed order (no atomics:
                                  the result is the same bits on every run)                              */
    size_t scratch_bytes;
} DgsDitLayerNormBackwardArgs;

/* Backward of dgs_dit_rowlinear (M <= 8): dW [N,K], db [N] and dx [M,K] are WRITTEN (dx through per-workgroup partial rows in
 * `scratch`, summed in a fixed order). */
typedef struct DgsDitRowLinearBackwardArgs {
    int32_t M, N, K;
    const float* x;            /* f32 [M, K] forward input (before the optional SiLU)                  */
    int32_t silu_input;
    const uint16_t* W;         /* bf16 [N, K]                                                         */
    const float* dy;           /* f32 [M, N]                                                          */
    float *dW, *db, *dx;       /* each optional                                                       */
    void* scratch;             /* dgs_dit_rowlinear_backward_scratch_bytes(M, N, K) bytes when dx is requested                         */
    size_t scratch_bytes;
} DgsDitRowLinearBackwardArgs;

/* dy = gate * dx (bf16 [B*rows, W] and its token-contiguous copy [B, W, rows]); dgate[b, n] = sum_t dx[t, n] y[t, n];
 * dbias[n] = sum_{b,t} dy[t, n] (the bias gradient of the Linear in front of the gate) -- both WRITTEN, order-deterministic. */
typedef struct DgsDitGateMulArgs {
    int32_t B, rows, width;
    const float* dx;           /* f32 [B*rows, width]                                                 */
    const uint16_t* y;         /* bf16 [B*rows, width]: pre-gate branch output saved by the forward    */
    const float* gate;         /* [B, gate_stride]                                                    */
    int32_t gate_stride;
    uint16_t *dy, *dyT;
    float* dgate;              /* [B, gate_stride]                                                    */
    float* dbias;              /* [width] or NULL                                                     */
    void* scratch;             /* dgs_dit_gate_mul_scratch_bytes(B, rows, width) bytes                */
    size_t scratch_bytes;
} DgsDitGateMulArgs;

typedef struct DgsDitRowLinearArgs {
    int32_t M, N, K;           /* M <= 16 rows; K % 512 == 0 or K == 256                              */
    const float* x;            /* f32 [M, K]                                                          */
    int32_t silu_input;        /* 1: apply SiLU to x first (adaLN_modulation = Sequential(SiLU, Linear)) */
    const uint16_t* W;         /* bf16 [N, K]                                                         */
    const float* bias;         /* [N] or NULL                                                         */
    int32_t silu_output;       /* 1: SiLU on the result (TimestepEmbedder mlp.1)                      */
    float* out;                /* f32 [M, N]                                                          */
} DgsDitRowLinearArgs;

typedef struct DgsDitLayerWeights {
    const uint16_t *qkv_w, *proj_w, *fc1_w, *fc2_w;   /* bf16 [3W,W] [W,W] [4W,W] [W,4W]              */
    const float *qkv_b, *proj_b, *fc1_b, *fc2_b;
} DgsDitLayerWeights;

typedef struct DgsDitModel {
    int32_t width, heads, layers, patch, in_channels, n_gaussians, gs_channels;
    int32_t scene;             /* 0: object model (denoiser.py), 1: scene model (denoiser_scene.py)   */
    int32_t relative_plk;      /* ray_pe_type: 1 = 'relative_plk', 0 = 'plk'                          */
    float range_near, range_far;
    const uint16_t* t_w0; const float* t_b0;    /* t_embedder.mlp.0  [W,256]                           */
    const uint16_t* t_w1; const float* t_b1;    /* t_embedder.mlp.2  [W,W]                             */
    const uint16_t* tok_w;                      /* image_tokenizer.1.weight [W, in_channels*patch^2]   */
    const float* pos_emb;                       /* gaussians_pos_embedding [n_gaussians, W]            */
    const float* in_ln_w;                       /* transformer_input_layernorm.weight                  */
    const DgsDitLayerWeights* layer;            /* HOST array [layers]                                 */
    const uint16_t* ada_w; const float* ada_b;  /* all adaLN_modulation.1 stacked: [layers*6W + 2W + 2W, W]:
                                                   block i rows [6W*i, 6W*(i+1)), then upsampler (2W), then image_token_decoder (2W) */
    const float* up_ln_w;  const uint16_t* up_w;   /* upsampler.layernorm.weight, upsampler.linear.weight [gs_channels, W] */
    const float* dec_ln_w; const uint16_t* dec_w;  /* image_token_decoder.*  linear [patch^2*gs_channels, W]             */
} DgsDitModel;

typedef struct DgsDitForwardArgs {
    int32_t B, V, H, W;
    const float* images;       /* [B,V,3,H,W] (first 3 channels of the reference's image tensor)      */
    const float* ray_o;        /* [B,V,3,H,W]                                                         */
    const float* ray_d;        /* [B,V,3,H,W]                                                         */
    const int64_t* t;          /* [B] diffusion timestep                                              */
    void* workspace;           /* dgs_dit_workspace_bytes(...) bytes, device; must have been zero-filled once after
                                  allocation and after every change of (B, V, H, W): padding rows are skipped and
                                  have to hold finite values                                            */
    size_t workspace_bytes;
    /* outputs in the reference's Gaussian order: P = n_gaussians + V*H*W, index 0..n_g-1 = learned tokens,
     * then (v, hh, ww, ph, pw) as denoiser.py:371-379 */
    float* xyz;                /* [B,P,3]   pixel-aligned                                             */
    float* features;           /* [B,P,1,3] (sh degree 0)                                             */
    float* scaling;            /* [B,P,3]   raw log-scale after (s-2.3).clamp(max=-1.2)               */
    float* rotation;           /* [B,P,4]   raw quaternion                                            */
    float* opacity;            /* [B,P,1]   raw (o - 2.0)                                             */
    float* aligned_xyz;        /* optional [B,V,3,H,W]                                                */
    float* tokens;             /* optional [B,L,W] f32: tokens after the last block, reference order  */
    /* optional measurement hook (bench.py roofline): hipEvent_t handles recorded on `stream` immediately before
     * and after every launch of ONE kernel class -- 1: attention, 2: QKV GEMM, 3: gate+residual GEMMs (proj, fc2),
     * 4: fc1 GEMM (+GELU), 5: LayerNorm+modulate, 6: the proj GEMM alone, 7: the fc2 GEMM alone.  prof_events holds 2 * prof_capacity handles, used in launch
     * order (before, after); *prof_count (host) receives the number of launches recorded.  NULL -> off.           */
    void** prof_events;
    int32_t prof_kind;
    int32_t prof_capacity;
    int32_t* prof_count;
    int32_t train_recompute;   /* dgs_dit_forward_train only. 0: save every activation the backward needs (nothing is recomputed);
                                  1: the reference's per-block checkpointing (torch.utils.checkpoint(run_layers(i, i+1)),
                                  denoiser.py:348-354): keep only each block's input, dgs_dit_backward re-runs the block       */
} DgsDitForwardArgs;

/* ---- training: forward that saves activations, and the backward ------------------------------------------------ */
typedef struct DgsDitLayerWeightsT {           /* K-contiguous copies for the input-gradient GEMMs: W^T */
    const uint16_t *qkv_wT, *proj_wT, *fc1_wT, *fc2_wT;   /* bf16 [W,3W] [W,W] [W,4W] [4W,W]            */
} DgsDitLayerWeightsT;

typedef struct DgsDitModelT {
    const DgsDitLayerWeightsT* layer;          /* HOST array [layers]                                    */
    const uint16_t* dec_wT;                    /* bf16 [W, patch^2*gs_channels]                          */
} DgsDitModelT;

typedef struct DgsDitLayerGrads {
    float *qkv_w, *proj_w, *fc1_w, *fc2_w, *qkv_b, *proj_b, *fc1_b, *fc2_b;
    float *ada_w, *ada_b;                      /* the block's adaLN_modulation.1: [6W, W], [6W] -- final together with the block's
                                                  other gradients (its rows of the stacked DgsDitModel.ada_w)              */
} DgsDitLayerGrads;

typedef struct DgsDitGrads {                   /* f32 gradient of every parameter, same shapes as the state dict; WRITTEN   */
    float *t_w0, *t_b0, *t_w1, *t_b1, *tok_w, *pos_emb, *in_ln_w;
    const DgsDitLayerGrads* layer;             /* HOST array [layers]                                    */
    float *head_ada_w, *head_ada_b;            /* the last 4W rows of the stacked DgsDitModel.ada_w / ada_b: upsampler (2W), then
                                                  image_token_decoder (2W); final with the heads (stage = layers)          */
    float *up_ln_w, *up_w, *dec_ln_w, *dec_w;
} DgsDitGrads;

typedef struct DgsDitBackwardArgs {
    int32_t B, V, H, W;
    const float* ray_d;                        /* [B,V,3,H,W]                                            */
    void* saved;        size_t saved_bytes;    /* arena filled by dgs_dit_forward_train                  */
    void* workspace;    size_t workspace_bytes;/* dgs_dit_backward_workspace_bytes, zero-filled once     */
    const float *dxyz, *dfeatures, *dscaling, *drotation, *dopacity;   /* gradients of the five outputs  */
    int32_t recompute;                         /* must equal the forward's train_recompute              */
    /* Optional host callback, called from inside dgs_dit_backward as soon as the kernels that complete a GROUP of
     * gradients have been enqueued on `stream` (they have not run yet: order follow-up work behind the stream):
     * stage = layers: the two heads (dec_w, dec_ln_w, up_w, up_ln_w, head_ada_w / _b); stage = layers-1 .. 0: that block's ten
     * tensors (incl. its adaLN Linear: a third of all parameters sits in those, none of it waits for the end of the backward);
     * stage = -1: the rest (embeddings, timestep MLP: ~2 M parameters).  This is where a data-parallel caller enqueues the
     * all-reduce of a finished gradient bucket so that it overlaps the remaining backward (Lightning DDP's overlap,
     * configs/diffusionGS_rel.yaml:80).                                                                            */
    void (*block_done)(void* user, int32_t stage);
    void* block_user;
} DgsDitBackwardArgs;

/* recompute: 0 / 1 as DgsDitForwardArgs.train_recompute */
size_t dgs_dit_saved_bytes(const DgsDitModel* m, int32_t B, int32_t V, int32_t H, int32_t W, int32_t recompute);
size_t dgs_dit_backward_workspace_bytes(const DgsDitModel* m, int32_t B, int32_t V, int32_t H, int32_t W);
/* same outputs as dgs_dit_forward (a->workspace is not used); B <= 4 */
int dgs_dit_forward_train(const DgsDitModel* m, const DgsDitForwardArgs* a, void* saved, size_t saved_bytes, dgs_stream_t stream);
int dgs_dit_backward(const DgsDitModel* m, const DgsDitModelT* mt, const DgsDitGrads* grads, const DgsDitBackwardArgs* a,
                     dgs_stream_t stream);

/* Padding contract of the token-row tensors (rows_per_batch > valid_rows): rows at and behind valid_rows are never computed; of the
 * 32-row block that holds the last valid rows only the valid rows are written.  Consumers (the attention kernel multiplies masked
 * probabilities 0 by the V^T padding columns) need FINITE values there: allocate `out`, `aux` and `vt` zero-filled (or any finite
 * fill) once; the library never writes a non-finite value into padding.                                                       */
int dgs_dit_gemm(const DgsDitGemmArgs* a, dgs_stream_t stream);
int32_t dgs_dit_gemm_sliced_tile(const DgsDitGemmArgs* a);   /* diagnostic: tile width (256 / 192 / 128) if this call runs on the 256-row
                                                                 ring kernel (dit_gemm_deep.hip), 0 otherwise; launches nothing */
int dgs_dit_attention(const DgsDitAttentionArgs* a, dgs_stream_t stream);
int dgs_dit_attention_backward(const DgsDitAttentionBackwardArgs* a, dgs_stream_t stream);
int dgs_dit_layernorm(const DgsDitLayerNormArgs* a, dgs_stream_t stream);
/* LayerNorm + modulate and the GEMM that consumes it (g->A == ln->out), as DiTBlock.forward runs them twice per block
 * (utils_transformer.py:271-290: norm1 -> attn.qkv, norm2 -> mlp.fc1): the same results as dgs_dit_layernorm followed by dgs_dit_gemm,
 * bit for bit.  When a sample ends one or two rows behind its last full 256-row tile (the DiT's learned tokens: L = 4096 v + 2) and
 * the GEMM runs on the 256-row kernel (QKV / GELU epilogues), those output rows are produced by the first workgroups of the LayerNorm
 * launch instead of by side jobs inside the GEMM launch (QKV -2.8 us, fc1 -2.4 us at one sample); every other shape is the two plain
 * launches.                                                                                                                       */
int dgs_dit_layernorm_gemm(const DgsDitLayerNormArgs* ln, const DgsDitGemmArgs* g, dgs_stream_t stream);
int32_t dgs_dit_layernorm_gemm_shares_rows(const DgsDitLayerNormArgs* ln, const DgsDitGemmArgs* g);   /* 1: the pair above takes that form */
int dgs_dit_layernorm_backward(const DgsDitLayerNormBackwardArgs* a, dgs_stream_t stream);
int dgs_dit_rowlinear_backward(const DgsDitRowLinearBackwardArgs* a, dgs_stream_t stream);
int dgs_dit_gate_mul(const DgsDitGateMulArgs* a, dgs_stream_t stream);
size_t dgs_dit_layernorm_backward_scratch_bytes(int32_t rows, int32_t width, int32_t rows_per_batch);
size_t dgs_dit_rowlinear_backward_scratch_bytes(int32_t M, int32_t N, int32_t K);
size_t dgs_dit_gate_mul_scratch_bytes(int32_t B, int32_t rows, int32_t width);
int dgs_dit_rowlinear(const DgsDitRowLinearArgs* a, dgs_stream_t stream);

/* DGSDenoiser.run_layers(first, last) (denoiser.py:441-447): DiT blocks [first, last) on a token tensor in the reference's
 * order.  The workspace is the one of dgs_dit_forward for a shape with the same token count (V views of H x W with
 * n_gaussians + V (H/patch) (W/patch) == L). */
typedef struct DgsDitRunBlocksArgs {
    int32_t B, L, V;           /* samples, tokens per sample (n_gaussians + image tokens), views (0: not given; else only used to
                                  check that L - n_gaussians is a multiple of it)                                              */
    int32_t first, last;       /* block range [first, last)                                                                  */
    const float* tokens_in;    /* f32 [B, L, W]: [gaussian tokens, image tokens]                                             */
    const float* cvec;         /* f32 [B, W]: the timestep embedding c = t_embedder(t)                                       */
    float* tokens_out;         /* f32 [B, L, W]                                                                              */
    void* workspace;           /* dgs_dit_workspace_bytes_for_tokens(m, B, L) bytes, zero-filled once                        */
    size_t workspace_bytes;
} DgsDitRunBlocksArgs;
size_t dgs_dit_workspace_bytes_for_tokens(const DgsDitModel* m, int32_t B, int32_t L);   /* the workspace depends on (B, L) only */
int dgs_dit_run_blocks(const DgsDitModel* m, const DgsDitRunBlocksArgs* a, dgs_stream_t stream);

/* Test hook: fills the LDS of every CU with NaN patterns (bf16 and f32).  The kernels above read operands that LDS-DMAs deliver
 * asynchronously; a read that overtakes its DMA would otherwise see the previous launch's -- usually identical -- data and pass
 * unnoticed.  tests/ call it in front of the kernels under test. */
int dgs_debug_poison_lds(dgs_stream_t stream);

/* Measurement hook (bench.py's `clock` object): 256 workgroups run a dependent fp32 chain for `microseconds` of the constant 100 MHz
 * clock (s_memrealtime); out (device int64[3]) receives {shader-clock cycles (s_memtime) workgroup 0 counted, 100 MHz ticks it
 * counted, 0}: effective shader clock = out[0] / out[1] x 100 MHz at the moment the stream reaches the probe. */
int dgs_debug_clock_probe(int64_t* out, int32_t microseconds, dgs_stream_t stream);

int32_t dgs_dit_lpad(int32_t L);   /* padded rows per sample */
size_t dgs_dit_workspace_bytes(const DgsDitModel* m, int32_t B, int32_t V, int32_t H, int32_t W);
int dgs_dit_forward(const DgsDitModel* m, const DgsDitForwardArgs* a, dgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DGS_DIT_H_ */
