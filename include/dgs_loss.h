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

/* The two loss terms on the denoiser's pixel-aligned points `img_aligned_xyz` (losses.py:288-292, 325-364; `lambda_pointsdist: 1.0`
 * is the only active term of the first training steps of configs/diffusionGS_rel.yaml), forward values and the gradient with
 * respect to `aligned` in two passes over the tensors, every sum reduced in a fixed order:
 *   points-distribution loss, per sample b:
 *       dist = |aligned - ray_o|_2 per pixel; per (b, view): mean and UNBIASED std of dist over the view's pixels (detached);
 *       target = (dist - mean) / (std + 1e-8) * 0.5 + |ray_o|_2;   pointsdist[b] = mean over (v, h, w) of (dist - target)^2
 *   xyz loss, one scalar for the batch (optional: gt and masks given):
 *       xyz = sum((aligned * m - gt * m)^2) / sum(m),  m = masks [B, V, 1, H, W] broadcast over the three coordinates
 *   grad = w_pointsdist[b] * d pointsdist[b] / d aligned + w_xyz * d xyz / d aligned      (optional)                            */
typedef struct DgsPointsLossArgs {
    int32_t B, V, H, W;
    const float* aligned;      /* f32 [B, V, 3, H, W]                                                       */
    const float* ray_o;        /* f32 [B, V, 3, H, W]                                                       */
    const float* gt;           /* optional f32 [B, V, 3, H, W]                                              */
    const float* masks;        /* optional f32 [B, V, 1, H, W] (required with gt)                            */
    float* pointsdist;         /* out f32 [B]                                                               */
    float* xyz;                /* out f32 [1] (written when gt is given)                                    */
    const float* w_pointsdist; /* optional f32 [B]: upstream gradient of pointsdist[b] (NULL: 0)             */
    float w_xyz;               /* upstream gradient of xyz                                                  */
    float* grad;               /* optional out f32 [B, V, 3, H, W]                                          */
    float* workspace;          /* f32 [dgs_points_loss_workspace_floats(B, V)]                               */
} DgsPointsLossArgs;

int64_t dgs_points_loss_workspace_floats(int32_t B, int32_t V);
int dgs_points_loss(const DgsPointsLossArgs* args, dgs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
