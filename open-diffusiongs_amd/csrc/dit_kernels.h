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

int launch_layernorm(const DgsDitLayerNormArgs* a, hipStream_t st);
int launch_rowlinear(const DgsDitRowLinearArgs* a, hipStream_t st);
int launch_timestep(const int64_t* t, float* emb, int B, hipStream_t st);
int launch_embed(const EmbedParams& p, hipStream_t st);
int launch_pos_embed(const float* pe, float* x, int B, int lpad, int L, int ng, int width, hipStream_t st);
int launch_gather_tokens(const float* x, float* out, int B, int lpad, int L, int ng, int width, hipStream_t st);
int launch_gaussians(const GsParams& p, hipStream_t st);

}  // namespace dgs
