/* dgs_optim.h -- C ABI of the optimizer step of the data-parallel training step (the training throughput
 * `north_star` names is "DiT fwd+bwd + raster fwd+bwd + gradient all-reduce + AdamW").
 *
 * Replaces, for the denoiser's parameters, what the reference's training loop does between two forward passes:
 *     torch.optim.AdamW(params, lr=1e-5, betas=[0.9, 0.99], eps=1e-8)        diffusionGS/configs/diffusionGS_rel.yaml:57-62,
 *                                                                             diffusionGS/utils/scheduler.py:34-53 (parse_optimizer)
 * plus what a bf16-MFMA engine has to do after it: bring its device-resident operand copies of the weights up to date (bf16 copies
 * [N, K] for the forward and input-gradient GEMMs, transposed bf16 copies [K, N] for the K-contiguous operand of the other one,
 * fp32 copies of the vectors).  As torch ops that is one multi-tensor AdamW launch per ~50 tensors + ~600 cast / transposed-copy /
 * copy launches per step; here it is ONE launch that reads p, g, m, v once and writes p, m, v and every copy.
 *
 * Arithmetic = torch's single-tensor AdamW (torch/optim/adamw.py, _single_tensor_adamw; fp32 throughout):
 *     p *= 1 - lr * weight_decay;  m += (g - m) * (1 - beta1);  v = v * beta2 + (1 - beta2) * g * g
 *     p -= (lr / bias_correction1) * m / (sqrt(v) / sqrt(bias_correction2) + eps),   bias_correction_i = 1 - beta_i ^ step
 * Device pointers, a HIP stream, no host synchronisation; returns DGS_OK or a negative DgsStatus.
 */
#ifndef DGS_OPTIM_H
#define DGS_OPTIM_H

#include <stdint.h>

#include "dgs_raster.h" /* DgsStatus, dgs_stream_t */

#ifdef __cplusplus
extern "C" {
#endif

#define DGS_OPTIM_COPY_NONE 0
#define DGS_OPTIM_COPY_BF16 1
#define DGS_OPTIM_COPY_F32 2

/* One parameter tensor.  `rows` x `cols` row-major (a vector: rows = 1).  A tensor with a transposed copy needs rows % 64 == 0 and
 * cols % 64 == 0 (every weight matrix of the DiT); `first_tile` is filled in by dgs_adamw_plan. */
typedef struct DgsAdamWTensor {
    float* p;          /* fp32 master parameter, updated in place                                   */
    const float* g;    /* fp32 gradient                                                             */
    float* m;          /* exp_avg                                                                   */
    float* v;          /* exp_avg_sq                                                                */
    void* copy;        /* optional row-major copy [rows, cols] of the NEW value: bf16 or f32         */
    void* copy_t;      /* optional transposed bf16 copy [cols, rows]                                */
    int64_t rows, cols;
    int32_t copy_kind; /* DGS_OPTIM_COPY_*                                                          */
    int32_t first_tile;
} DgsAdamWTensor;

typedef struct DgsAdamWArgs {
    const DgsAdamWTensor* tensors; /* DEVICE pointer to the planned table                           */
    int32_t n_tensors;
    int32_t n_tiles;               /* value returned by dgs_adamw_plan                              */
    float lr, beta1, beta2, eps, weight_decay;
    float bias_correction1;        /* 1 - beta1 ^ step  (computed by the caller, in double)          */
    float bias_correction2_sqrt;   /* sqrt(1 - beta2 ^ step)                                        */
    /* Global-norm gradient clip folded into the step (the reference trains with `gradient_clip_val: 0.5`,
     * diffusionGS/configs/diffusionGS_rel.yaml:76-77 = Lightning's norm clip = torch.nn.utils.clip_grad_norm_): every gradient is
     * read as g * min(1, max_grad_norm / (sqrt(*grad_sumsq) + 1e-6)); the gradient tensors themselves are not rewritten.
     * max_grad_norm <= 0: no clip (grad_sumsq is not read). */
    const float* grad_sumsq;       /* DEVICE float[1]: sum of squares of ALL gradients (dgs_sumsq_partials + dgs_sumsq_finish) */
    float max_grad_norm;
} DgsAdamWArgs;

/* HOST: fills `first_tile` of every entry of a host-side table and returns the launch's tile count (< 0: invalid table, e.g. a
 * transposed copy of a tensor whose sides are not multiples of 64).  Copy the table to the device afterwards. */
int32_t dgs_adamw_plan(DgsAdamWTensor* host_tensors, int32_t n_tensors);

/* One launch: AdamW step on every tensor of the table + its copies. */
int dgs_adamw_step(const DgsAdamWArgs* args, dgs_stream_t stream);

/* Sum of squares of a gradient buffer, in two deterministic stages: dgs_sumsq_partials WRITES one partial per 65,536 consecutive
 * elements of x[0, n) into partials[0, dgs_sumsq_count(n)) (x 16-byte aligned) -- callable per bucket of a larger buffer, as each
 * bucket becomes final, in any order; dgs_sumsq_finish adds `count` partials in index order into total[0].  Same bits every run. */
int32_t dgs_sumsq_count(int64_t n);
int dgs_sumsq_partials(const float* x, int64_t n, float* partials, dgs_stream_t stream);
int dgs_sumsq_finish(const float* partials, int32_t count, float* total, dgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
