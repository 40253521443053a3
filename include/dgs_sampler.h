/* dgs_sampler.h -- C ABI of the diffusion sampler step: the elementwise update that follows the
 * denoiser + rasterizer in every iteration of the sampling loop.
 *
 * Replaces, for the configuration the reference ships (create_diffusion: predict_xstart=True, learn_sigma=False;
 * diffusionGS/models/diffusion/__init__.py:15-51), the tensor-op chain of
 *     GaussianDiffusion.p_mean_variance   gaussian_diffusion.py:316-460  (x0 = clamp(model_output, -1, 1); posterior mean)
 *     GaussianDiffusion.q_posterior_mean_variance   :291-313
 *     GaussianDiffusion.p_sample          :479-518                        (x_{t-1} = mean + [t != 0] exp(0.5 log_var) noise)
 * which rebuilds numpy -> tensor coefficient tables four times per step (_extract_into_tensor, :853-866).  Here the tables
 * live on the device, the timestep index is read on the device (no host synchronisation), one launch per step.
 * Plain pointers and sizes only; device pointers; returns DGS_OK or a negative DgsStatus.
 */
#ifndef DGS_SAMPLER_H
#define DGS_SAMPLER_H

#include <stdint.h>

#include "dgs_raster.h" /* DgsStatus, dgs_stream_t */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct DgsSamplerStepArgs {
    int32_t B;                 /* samples                                                                             */
    int64_t per_sample;        /* elements per sample of x_t / noise / out ((V-1) * C * H * W)                        */
    int64_t model_stride;      /* elements between samples of model_output (V * C * H * W when it is the render)       */
    int64_t model_offset;      /* first predicted element inside a sample of model_output (C * H * W: view 0 is the
                                  conditioning view, render_imgs[:, 1:] is the prediction; gaussian_diffusion.py:351)   */
    const float* model_output; /* f32: the denoiser's x0 prediction (the rendered views)                              */
    const float* x_t;          /* f32 [B, per_sample]: current noisy views (input_batch["image_noisy"])               */
    const float* noise;        /* f32 [B, per_sample]: N(0, 1) draws (th.randn_like(x), :504); may be NULL iff every t is 0 */
    const int64_t* t;          /* [B] loop indices into the (respaced) tables                                         */
    const float* coef1;        /* [T] posterior_mean_coef1 (multiplies x0)                                            */
    const float* coef2;        /* [T] posterior_mean_coef2 (multiplies x_t)                                           */
    const float* sigma;        /* [T] exp(0.5 * model log-variance)                                                   */
    int32_t T;                 /* table length; t[b] outside [0, T) -> DGS_ERR_INVALID_ARGUMENT is NOT detectable on the
                                  device without a sync: such a sample is left untouched and flagged in *bad_t (if given) */
    int32_t clip_denoised;     /* clamp x0 to [-1, 1] (the reference's default)                                       */
    float* out;                /* f32 [B, per_sample]: x_{t-1}; may alias x_t                                         */
    float* pred_xstart;        /* optional f32 [B, per_sample]: the (clipped) x0                                      */
    int32_t* bad_t;            /* optional device int: set to 1 when some t[b] is out of range                        */
} DgsSamplerStepArgs;

int dgs_sampler_step(const DgsSamplerStepArgs* args, dgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
