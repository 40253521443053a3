/* dgs_loss.h -- C ABI of the image-space loss consumer: the op either side of the rasterizer
 * backward in a training step.
 *
 * Replaces the per-sample L2 / PSNR terms of diffusionGS/utils/losses.py:
 *     per_element_loss = F.mse_loss(rendering, target, reduction='none'); l2_loss = mean over (v, c, h, w)   :281-285
 *     psnr = -10 * log10(l2_loss)                                                                           :303
 *     compute_psnr: clamp both to [0, 1] first                                                               :399-402
 * and the autograd backward of the lambda_mse term (d/d rendering = 2 (rendering - target) * grad_scale / n).
 * LPIPS (a VGG network) and SSIM are out of scope.  One pass over the images, per-sample sums reduced in a fixed order
 * (deterministic), no host synchronisation.  Device pointers; returns DGS_OK or a negative DgsStatus.
 */
#ifndef DGS_LOSS_H
#define DGS_LOSS_H

#include <stdint.h>

#include "dgs_raster.h" /* DgsStatus, dgs_stream_t */

#ifdef __cplusplus
extern "C" {
#endif

#define DGS_LOSS_CHUNKS 64 /* partial sums per sample */

typedef struct DgsMseArgs {
    int32_t B;                 /* samples                                                                  */
    int64_t n;                 /* elements per sample (v * 3 * h * w), a multiple of 4                     */
    const float* rendering;    /* f32 [B, n]                                                               */
    const float* target;       /* f32 [B, n]                                                               */
    int32_t clamp01;           /* clamp both to [0, 1] before the difference (compute_psnr, :399-400)       */
    float* l2;                 /* out f32 [B]: mean squared error per sample                               */
    float* psnr;               /* optional out f32 [B]: -10 log10(l2)                                      */
    float* grad;               /* optional out f32 [B, n]: grad_scale * 2 (rendering - target) / n          */
    float grad_scale;          /* e.g. lambda_mse / B for the batch-mean loss                              */
    float* partial;            /* workspace f32 [B, DGS_LOSS_CHUNKS]                                        */
} DgsMseArgs;

int dgs_mse_psnr(const DgsMseArgs* args, dgs_stream_t stream);

/* The input path of the LPIPS term (losses.py:304-309): `F.interpolate(x, size=[256, 256], mode='bilinear') * 2.0 - 1.0`
 * for renderings and targets -- evaluated every step, also when lambda_lpips is 0.  One launch: bilinear resize
 * (align_corners = False, PyTorch's source-index rule) fused with the affine map; the backward accumulates
 * d src = mul * W^T d dst (dsrc is zero-filled by the call).  The LPIPS network itself is out of scope. */
typedef struct DgsResizeArgs {
    int32_t planes;            /* n * c image planes                                                       */
    int32_t in_h, in_w, out_h, out_w;
    const float* src;          /* f32 [planes, in_h, in_w]       (forward)                                 */
    float* dst;                /* f32 [planes, out_h, out_w]     (forward)                                 */
    float mul, add;            /* dst = resize(src) * mul + add                                            */
    const float* ddst;         /* f32 [planes, out_h, out_w]     (backward)                                */
    float* dsrc;               /* f32 [planes, in_h, in_w]       (backward, written)                       */
} DgsResizeArgs;

int dgs_resize_bilinear(const DgsResizeArgs* args, dgs_stream_t stream);
int dgs_resize_bilinear_backward(const DgsResizeArgs* args, dgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
