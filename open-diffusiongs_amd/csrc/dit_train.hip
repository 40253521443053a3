// dit_train.hip -- training-mode forward and the backward of DGSDenoiser.image_to_gaussians (denoiser.py:306-416) as two
// C calls, each one launch sequence on one stream with no host synchronisation.
//
// The reference trains with per-block activation checkpointing (torch.utils.checkpoint, denoiser.py:348-354): every block
// is recomputed in backward, 4x the forward FLOPs per step.  MI355X has 288 GB of HBM, so by default the training forward
// SAVES what the backward needs (54 W bytes per token and block, ~22 GB at batch 4 x 256^2) and nothing is recomputed: 3x.
// `recompute` (DgsDitForwardArgs.train_recompute / DgsDitBackwardArgs.recompute) is the reference's mode: only each
// block's INPUT is kept (4 W bytes per token and block) plus ONE block's worth of activations; the backward re-runs a
// block's forward (same kernels, same inputs: same bits) right before differentiating it.  That is what makes the 512^2
// configuration (L = 16,386; BASELINE configs[4]) trainable at the reference's batch sizes.
// Gradients come out as fp32 tensors in caller-provided buffers (one flat buffer in practice: dgs_amd/parallel.py), weight
// gradients are GEMMs whose reduction runs over the tokens of token-contiguous ("transposed") activation copies.
#include "dit_kernels.h"
#include "raster_state.h"

namespace dgs {

struct BlockSaved {
    float* x_in;     // [M, W]   block input
    bf16_t* h1;      // [M, W]   LN+modulate output (attention branch)
    bf16_t* qkv;     // [M, 3W]
    bf16_t* qkvT;    // [B, 3W, lpad]
    float* lse2;     // [B, heads, lpad]
    bf16_t* a;       // [M, W]   attention output
    bf16_t* y1;      // [M, W]   proj output before the gate
    float* x_mid;    // [M, W]
    bf16_t* h2;      // [M, W]
    bf16_t* u;       // [M, 4W]  fc1 output before GELU
    bf16_t* g;       // [M, 4W]  GELU(u)
    bf16_t* gT;      // [B, 4W, lpad]
    bf16_t* y2;      // [M, W]   fc2 output before the gate
};

struct DitSaved {
    BlockSaved blk[64];
    float* x_out;    // [M, W]   output of the last block
    float* x0_pre;   // [M, W]   tokenizer output + learned tokens, before the input LayerNorm
    bf16_t* emb;     // [M, kin]
    bf16_t* xn_dec;  // [M, W]
    float* dec;      // [M, pp*C]
    float *upn, *up, *temb, *c1, *cvec, *mod;
    void* attn_tail; size_t attn_tail_bytes;
    static DitSaved carve(void* buf, const DgsDitModel* m, size_t B, size_t lpad, int L, int recompute, size_t* bytes) {
        Carver c(buf);
        DitSaved s;
        const size_t M = B * lpad, W = (size_t)m->width, pp = (size_t)m->patch * m->patch;
        for (int i = 0; i < m->layers; ++i) {
            BlockSaved& k = s.blk[i];
            if (recompute && i > 0) {            // every block shares block 0's activation slots; only x_in is its own
                k = s.blk[0];
                k.x_in = c.take<float>(M * W);
                continue;
            }
            k.x_in = c.take<float>(M * W); k.h1 = c.take<bf16_t>(M * W); k.qkv = c.take<bf16_t>(M * 3 * W);
            k.qkvT = c.take<bf16_t>(M * 3 * W); k.lse2 = c.take<float>(B * m->heads * lpad); k.a = c.take<bf16_t>(M * W);
            k.y1 = c.take<bf16_t>(M * W); k.x_mid = c.take<float>(M * W); k.h2 = c.take<bf16_t>(M * W);
            k.u = c.take<bf16_t>(M * 4 * W); k.g = c.take<bf16_t>(M * 4 * W); k.gT = c.take<bf16_t>(M * 4 * W);
            k.y2 = c.take<bf16_t>(M * W);
        }
        s.x_out = c.take<float>(M * W); s.x0_pre = c.take<float>(M * W);
        s.emb = c.take<bf16_t>(M * pp * m->in_channels); s.xn_dec = c.take<bf16_t>(M * W);
        s.dec = c.take<float>(M * pp * m->gs_channels);
        s.upn = c.take<float>(B * m->n_gaussians * W); s.up = c.take<float>(B * m->n_gaussians * m->gs_channels);
        s.temb = c.take<float>(B * 256); s.c1 = c.take<float>(B * W); s.cvec = c.take<float>(B * W);
        s.mod = c.take<float>(B * (6 * (size_t)m->layers + 4) * W);
        s.attn_tail_bytes = dgs_dit_attention_tail_bytes((int)B, m->heads, L);
        s.attn_tail = c.take<char>(s.attn_tail_bytes);
        if (bytes) *bytes = c.bytes();
        return s;
    }
};

struct BwdScratch {
    float *dxa, *dxb;        // [M, W] residual-stream gradient ping-pong
    bf16_t *dy, *dyT;        // [M, W], [B, W, lpad]
    bf16_t *du, *duT;        // [M, 4W], [B, 4W, lpad]
    bf16_t* dh;              // [M, W]
    bf16_t *da, *daT;        // [M, W], [B, W, lpad]
    bf16_t *dqkv, *dqkvT;    // [M, 3W], [B, 3W, lpad]
    bf16_t* actT;            // [B, W, lpad]  transposed h1 / h2 / a / xn_dec
    bf16_t *ddec, *ddecT;    // [M, pp*C], [B, pp*C, lpad]
    bf16_t* embT;            // [B, kin, lpad]
    float* D;                // [B, heads, lpad]
    float *dmod, *dup, *dupn, *dcvec, *dc1, *ones;
    size_t wpart_bytes;
    float* wpart;            // split-K partial planes of the weight-gradient GEMMs (largest: fc1 / fc2)
    float* xre;              // [M, W] recompute mode: where a re-run block writes its (already known) output
    // slabs of per-workgroup partial column sums (dit_backward_elementwise.hip; summed by col_reduce in slot order)
    float* ln_part[2];       // [M / 32][3W]   LayerNorm backward: shift | scale | weight     ([0]: LN2 / heads / input LN, [1]: LN1)
    float* gate_part[2];     // [M / 64][2W]   gate_mul: gate gradient | bias gradient         ([0]: MLP branch, [1]: attention branch)
    float *fc1b_part, *qkvb_part;   // [M / 128][4W], [M / 128][3W]   bias gradients of fc1 / qkv
    float* rl_part;          // [nmod / 256][B * W]   dx of the adaLN Linear
    static BwdScratch carve(void* buf, const DgsDitModel* m, size_t B, size_t lpad, size_t* bytes) {
        Carver c(buf);
        BwdScratch s;
        const size_t M = B * lpad, W = (size_t)m->width, pp = (size_t)m->patch * m->patch;
        s.dxa = c.take<float>(M * W); s.dxb = c.take<float>(M * W);
        s.dy = c.take<bf16_t>(M * W); s.dyT = c.take<bf16_t>(M * W);
        s.du = c.take<bf16_t>(M * 4 * W); s.duT = c.take<bf16_t>(M * 4 * W);
        s.dh = c.take<bf16_t>(M * W);
        s.da = c.take<bf16_t>(M * W); s.daT = c.take<bf16_t>(M * W);
        s.dqkv = c.take<bf16_t>(M * 3 * W); s.dqkvT = c.take<bf16_t>(M * 3 * W);
        s.actT = c.take<bf16_t>(M * W);
        const size_t nd = (pp * m->gs_channels + 63) / 64 * 64;     // transposes work on multiples of 64 features
        s.ddec = c.take<bf16_t>(M * nd); s.ddecT = c.take<bf16_t>(M * nd);
        s.embT = c.take<bf16_t>(M * pp * m->in_channels);
        s.D = c.take<float>(B * m->heads * lpad);
        s.dmod = c.take<float>(B * (6 * (size_t)m->layers + 4) * W);
        s.dup = c.take<float>(B * m->n_gaussians * m->gs_channels); s.dupn = c.take<float>(B * m->n_gaussians * W);
        s.dcvec = c.take<float>(B * W); s.dc1 = c.take<float>(B * W); s.ones = c.take<float>(B * W);
        s.wpart_bytes = dgs_dit_gemm_splitk_bytes((int)(4 * W), (int)W, (int)M, (int)lpad);
        s.wpart = c.take<float>(s.wpart_bytes / sizeof(float));
        s.xre = c.take<float>(M * W);
        const size_t ln_blk = M / ln_backward_rows_per_block((int)lpad, (int)M, ln_backward_compute_units());
        for (int i = 0; i < 2; ++i) { s.ln_part[i] = c.take<float>(ln_blk * 3 * W); s.gate_part[i] = c.take<float>(M / 64 * 2 * W); }
        s.fc1b_part = c.take<float>((M + 127) / 128 * 4 * W); s.qkvb_part = c.take<float>((M + 127) / 128 * 3 * W);
        const size_t nmod_rows = (6 * (size_t)m->layers + 4) * W;
        s.rl_part = c.take<float>((nmod_rows + ROWLINEAR_BWD_ROWS - 1) / ROWLINEAR_BWD_ROWS * B * W);
        if (bytes) *bytes = c.bytes();
        return s;
    }
};

static int tokens_of(const DgsDitModel* m, int V, int H, int W) { return m->n_gaussians + V * (H / m->patch) * (W / m->patch); }

__global__ void fill_kernel(float* p, float v, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
// dpos[g, c] = sum_b dx[b, L - ng + g, c]
__global__ void pos_embed_backward_kernel(const float* dx, float* dpos, int B, int lpad, int L, int ng, int width) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ng * width) return;
    const int c = i % width, g = i / width;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dx[((size_t)b * lpad + (L - ng) + g) * width + c];
    dpos[i] = s;
}

}  // namespace dgs

using namespace dgs;

#define DGS_TRY(expr) do { const int rc_ = (expr); if (rc_ != DGS_OK) { fprintf(stderr, "[dgs] %s:%d: status %d\n", __FILE__, __LINE__, rc_); return rc_; } } while (0)
#define HIP_TRY(expr) do { if ((expr) != hipSuccess) return DGS_ERR_DEVICE; } while (0)

extern "C" size_t dgs_dit_saved_bytes(const DgsDitModel* m, int32_t B, int32_t V, int32_t H, int32_t W, int32_t recompute) {
    if (!m || B <= 0 || m->layers > 64) return 0;
    size_t b = 0;
    DitSaved::carve(nullptr, m, (size_t)B, (size_t)dgs_dit_lpad(tokens_of(m, V, H, W)), tokens_of(m, V, H, W), recompute, &b);
    return b;
}

namespace {
// One DiTBlock of the training forward (utils_transformer.py:271-290): reads k.x_in, fills k's activation slots, writes the
// block output to x_next.  Also what the recompute mode re-runs inside the backward.
int block_forward_train(const DgsDitModel* m, int i, const BlockSaved& k, float* x_next, const float* mod_all, void* attn_tail,
                        size_t attn_tail_bytes, int B, int lpad, int L, dgs_stream_t stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int W = m->width, M = B * lpad, nmod = (6 * m->layers + 4) * W;
    const DgsDitLayerWeights& lw = m->layer[i];
    const float* mod = mod_all + (size_t)i * 6 * W;
    DgsDitLayerNormArgs l1{};
    l1.rows = M; l1.width = W; l1.x = k.x_in; l1.shift = mod; l1.scale = mod + W; l1.mod_stride = nmod; l1.rows_per_batch = lpad;
    l1.eps = 1e-6f; l1.out = k.h1;
    DGS_TRY(launch_layernorm(&l1, st));
    DgsDitGemmArgs q{};
    q.M = M; q.N = 3 * W; q.K = W; q.A = k.h1; q.lda = W; q.W = lw.qkv_w; q.ldw = W; q.bias = lw.qkv_b; q.epilogue = DGS_EPI_BF16;
    q.out = k.qkv; q.ldo = 3 * W; q.vt = k.qkvT; q.rows_per_batch = lpad; q.valid_rows = L;
    DGS_TRY(dgs_dit_gemm(&q, stream));
    DgsDitAttentionArgs at{};
    at.B = B; at.heads = m->heads; at.L = L; at.lpad = lpad; at.qk = k.qkv; at.ld_qk = 3 * W; at.k_offset = W;
    at.vt = k.qkvT + (size_t)2 * W * lpad; at.vt_batch_stride = (int64_t)3 * W * lpad; at.out = k.a; at.scale = 0.125f; at.lse2 = k.lse2;
    at.tail_ws = attn_tail; at.tail_ws_bytes = attn_tail_bytes;
    DGS_TRY(dgs_dit_attention(&at, stream));
    DgsDitGemmArgs pr{};
    pr.M = M; pr.N = W; pr.K = W; pr.A = k.a; pr.lda = W; pr.W = lw.proj_w; pr.ldw = W; pr.bias = lw.proj_b;
    pr.epilogue = DGS_EPI_GATE_RESIDUAL; pr.resid = k.x_in; pr.out = k.x_mid; pr.aux = k.y1; pr.ldo = W; pr.gate = mod + 2 * W;
    pr.gate_stride = nmod; pr.rows_per_batch = lpad; pr.valid_rows = L;
    DGS_TRY(dgs_dit_gemm(&pr, stream));
    l1.x = k.x_mid; l1.shift = mod + 3 * W; l1.scale = mod + 4 * W; l1.out = k.h2;
    DGS_TRY(launch_layernorm(&l1, st));
    DgsDitGemmArgs f1{};
    f1.M = M; f1.N = 4 * W; f1.K = W; f1.A = k.h2; f1.lda = W; f1.W = lw.fc1_w; f1.ldw = W; f1.bias = lw.fc1_b;
    f1.epilogue = DGS_EPI_GELU_BF16; f1.out = k.g; f1.aux = k.u; f1.vt = k.gT; f1.ldo = 4 * W; f1.rows_per_batch = lpad; f1.valid_rows = L;
    DGS_TRY(dgs_dit_gemm(&f1, stream));
    DgsDitGemmArgs f2{};
    f2.M = M; f2.N = W; f2.K = 4 * W; f2.A = k.g; f2.lda = 4 * W; f2.W = lw.fc2_w; f2.ldw = 4 * W; f2.bias = lw.fc2_b;
    f2.epilogue = DGS_EPI_GATE_RESIDUAL; f2.resid = k.x_mid; f2.out = x_next; f2.aux = k.y2; f2.ldo = W; f2.gate = mod + 5 * W;
    f2.gate_stride = nmod; f2.rows_per_batch = lpad; f2.valid_rows = L;
    DGS_TRY(dgs_dit_gemm(&f2, stream));
    return DGS_OK;
}
}  // namespace

extern "C" size_t dgs_dit_backward_workspace_bytes(const DgsDitModel* m, int32_t B, int32_t V, int32_t H, int32_t W) {
    if (!m || B <= 0) return 0;
    size_t b = 0;
    BwdScratch::carve(nullptr, m, (size_t)B, (size_t)dgs_dit_lpad(tokens_of(m, V, H, W)), &b);
    return b;
}

// Training-mode forward: same kernels as dgs_dit_forward, but q|k|v stay in one tensor (+ its transposed copy), the
// residual stream is written out of place, and every tensor the backward needs lands in the `saved` arena.
extern "C" int dgs_dit_forward_train(const DgsDitModel* m, const DgsDitForwardArgs* a, void* saved, size_t saved_bytes, dgs_stream_t stream) {
    if (!m || !a || !saved || a->B <= 0 || a->B > 4 || m->layers > 64) return DGS_ERR_INVALID_ARGUMENT;   // B * n_gaussians <= 8 rows
    if (m->width % 256 || m->width != m->heads * 64 || a->H % m->patch || a->W % m->patch || m->gs_channels != 14 || m->in_channels != 9)
        return DGS_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int B = a->B, V = a->V, H = a->H, Wd = a->W, W = m->width, ng = m->n_gaussians, C = m->gs_channels;
    const int L = tokens_of(m, V, H, Wd), lpad = dgs_dit_lpad(L), M = B * lpad;
    const int pp = m->patch * m->patch, kin = m->in_channels * pp, nmod = (6 * m->layers + 4) * W;
    size_t need = 0;
    DitSaved sv = DitSaved::carve(saved, m, (size_t)B, (size_t)lpad, L, a->train_recompute, &need);
    if (saved_bytes < need) return DGS_ERR_ALLOC;

    DGS_TRY(launch_timestep(a->t, sv.temb, B, st));
    DgsDitRowLinearArgs r{};
    r.M = B; r.N = W; r.K = 256; r.x = sv.temb; r.W = m->t_w0; r.bias = m->t_b0; r.out = sv.c1;          // c1 = PRE-activation
    DGS_TRY(launch_rowlinear(&r, st));
    r.K = W; r.x = sv.c1; r.silu_input = 1; r.W = m->t_w1; r.bias = m->t_b1; r.out = sv.cvec;
    DGS_TRY(launch_rowlinear(&r, st));
    r.N = nmod; r.x = sv.cvec; r.W = m->ada_w; r.bias = m->ada_b; r.out = sv.mod;
    DGS_TRY(launch_rowlinear(&r, st));

    HIP_TRY(hipMemsetAsync(sv.emb, 0, (size_t)M * kin * sizeof(bf16_t), st));
    EmbedParams ep;
    ep.B = B; ep.V = V; ep.H = H; ep.W = Wd; ep.ps = m->patch; ep.lpad = lpad; ep.relative_plk = m->relative_plk;
    ep.images = a->images; ep.ray_o = a->ray_o; ep.ray_d = a->ray_d; ep.out = sv.emb;
    DGS_TRY(launch_embed(ep, st));
    DgsDitGemmArgs g{};
    g.M = M; g.N = W; g.K = kin; g.A = sv.emb; g.lda = kin; g.W = m->tok_w; g.ldw = kin; g.epilogue = DGS_EPI_F32; g.out = sv.x0_pre; g.ldo = W;
    g.rows_per_batch = lpad; g.valid_rows = L;
    DGS_TRY(dgs_dit_gemm(&g, stream));
    DGS_TRY(launch_pos_embed(m->pos_emb, sv.x0_pre, B, lpad, L, ng, W, st));
    DgsDitLayerNormArgs ln{};
    ln.rows = M; ln.width = W; ln.x = sv.x0_pre; ln.weight = m->in_ln_w; ln.eps = 1e-5f; ln.out = sv.blk[0].x_in; ln.out_f32 = 1; ln.rows_per_batch = lpad;
    DGS_TRY(launch_layernorm(&ln, st));

    for (int i = 0; i < m->layers; ++i)
        DGS_TRY(block_forward_train(m, i, sv.blk[i], (i + 1 < m->layers) ? sv.blk[i + 1].x_in : sv.x_out, sv.mod, sv.attn_tail,
                                    sv.attn_tail_bytes, B, lpad, L, stream));
    if (a->tokens) DGS_TRY(launch_gather_tokens(sv.x_out, a->tokens, B, lpad, L, ng, W, st));

    const float* mod_up = sv.mod + (size_t)m->layers * 6 * W;
    const float* mod_dec = mod_up + 2 * W;
    DgsDitLayerNormArgs ld{};
    ld.rows = M; ld.width = W; ld.x = sv.x_out; ld.weight = m->dec_ln_w; ld.shift = mod_dec; ld.scale = mod_dec + W; ld.mod_stride = nmod;
    ld.rows_per_batch = lpad; ld.eps = 1e-5f; ld.out = sv.xn_dec;
    DGS_TRY(launch_layernorm(&ld, st));
    DgsDitGemmArgs dg{};
    dg.M = M; dg.N = pp * C; dg.K = W; dg.A = sv.xn_dec; dg.lda = W; dg.W = m->dec_w; dg.ldw = W; dg.epilogue = DGS_EPI_F32; dg.out = sv.dec;
    dg.ldo = pp * C; dg.rows_per_batch = lpad; dg.valid_rows = L;
    DGS_TRY(dgs_dit_gemm(&dg, stream));
    for (int b = 0; b < B; ++b) {
        DgsDitLayerNormArgs lu{};
        lu.rows = ng; lu.width = W; lu.x = sv.x_out + ((size_t)b * lpad + (L - ng)) * W; lu.weight = m->up_ln_w;
        lu.shift = mod_up + (size_t)b * nmod; lu.scale = mod_up + W + (size_t)b * nmod; lu.mod_stride = nmod; lu.rows_per_batch = ng;
        lu.eps = 1e-5f; lu.out = sv.upn + (size_t)b * ng * W; lu.out_f32 = 1;
        DGS_TRY(launch_layernorm(&lu, st));
    }
    DgsDitRowLinearArgs ru{};
    ru.M = B * ng; ru.N = C; ru.K = W; ru.x = sv.upn; ru.W = m->up_w; ru.out = sv.up;
    DGS_TRY(launch_rowlinear(&ru, st));
    GsParams gp;
    gp.B = B; gp.V = V; gp.H = H; gp.W = Wd; gp.ps = m->patch; gp.lpad = lpad; gp.ng = ng; gp.C = C; gp.scene = m->scene;
    gp.relative_plk = m->relative_plk; gp.range_near = m->range_near; gp.range_far = m->range_far;
    gp.dec = sv.dec; gp.up = sv.up; gp.ray_o = a->ray_o; gp.ray_d = a->ray_d;
    gp.xyz = a->xyz; gp.features = a->features; gp.scaling = a->scaling; gp.rotation = a->rotation; gp.opacity = a->opacity;
    gp.aligned = a->aligned_xyz;
    DGS_TRY(launch_gaussians(gp, st));
    return DGS_OK;
}

namespace {

// C[N, K] (f32) = sum over samples and tokens of  dYT[b, n, t] * XT[b, k, t]
int wgrad(const bf16_t* dyT, int N, const bf16_t* xT, int K, float* dW, int B, int lpad, const BwdScratch& ws, dgs_stream_t stream) {
    DgsDitGemmArgs g{};
    g.M = N; g.N = K; g.K = B * lpad; g.A = dyT; g.lda = lpad; g.W = xT; g.ldw = lpad; g.epilogue = DGS_EPI_F32; g.out = dW; g.ldo = K;
    g.k_per_batch = lpad; g.a_batch_stride = (int64_t)N * lpad; g.w_batch_stride = (int64_t)K * lpad;
    const size_t need = dgs_dit_gemm_splitk_bytes(N, K, B * lpad, lpad);     // scratch is sized for the 4W x W gradients
    g.splitk_ws = need && need <= ws.wpart_bytes ? ws.wpart : nullptr;
    return dgs_dit_gemm(&g, stream);
}

// dX[M, K] (bf16) = dY[M, N] . W[N, K]  using the K-contiguous copy WT[K, N]
int dgrad(const bf16_t* dy, int N, const bf16_t* wT, int K, bf16_t* dx, int M, int lpad, int L, int epilogue, const void* aux, bf16_t* dxT,
          dgs_stream_t stream) {
    DgsDitGemmArgs g{};
    g.M = M; g.N = K; g.K = N; g.A = dy; g.lda = N; g.W = wT; g.ldw = N; g.epilogue = epilogue; g.out = dx; g.ldo = K;
    g.aux = const_cast<void*>(aux); g.vt = dxT; g.rows_per_batch = lpad; g.valid_rows = L;
    return dgs_dit_gemm(&g, stream);
}

}  // namespace

extern "C" int dgs_dit_backward(const DgsDitModel* m, const DgsDitModelT* mt, const DgsDitGrads* gr, const DgsDitBackwardArgs* a,
                                dgs_stream_t stream) {
    if (!m || !mt || !gr || !a || a->B <= 0 || a->B > 4 || !a->saved || !a->workspace) return DGS_ERR_INVALID_ARGUMENT;
    if (!a->dxyz || !a->dfeatures || !a->dscaling || !a->drotation || !a->dopacity || !a->ray_d) return DGS_ERR_INVALID_ARGUMENT;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int B = a->B, V = a->V, H = a->H, Wd = a->W, W = m->width, ng = m->n_gaussians, C = m->gs_channels;
    const int L = tokens_of(m, V, H, Wd), lpad = dgs_dit_lpad(L), M = B * lpad;
    const int pp = m->patch * m->patch, kin = m->in_channels * pp, nmod = (6 * m->layers + 4) * W, ND = pp * C;
    if (ND % 128 || kin % 64) return DGS_ERR_INVALID_ARGUMENT;
    size_t need = 0;
    DitSaved sv = DitSaved::carve(a->saved, m, (size_t)B, (size_t)lpad, L, a->recompute, &need);
    if (a->saved_bytes < need) return DGS_ERR_ALLOC;
    BwdScratch ws = BwdScratch::carve(a->workspace, m, (size_t)B, (size_t)lpad, &need);
    if (a->workspace_bytes < need) return DGS_ERR_ALLOC;

    // No accumulator is filled with atomics: every column sum goes through a slab of per-workgroup partial rows and col_reduce
    // (dit_backward_elementwise.hip), so two passes over the same inputs give the same bits.  What is zeroed here: the decoder
    // head's gradient rows nobody writes (learned-token / padding rows), and the two tensors the single-workgroup launches of
    // the upsampler head ADD to (stream-ordered plain adds).
    HIP_TRY(hipMemsetAsync(ws.dmod, 0, (size_t)B * nmod * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(ws.ddec, 0, (size_t)M * ND * sizeof(bf16_t), st));
    HIP_TRY(hipMemsetAsync(ws.dupn, 0, (size_t)B * ng * W * sizeof(float), st));
    HIP_TRY(hipMemsetAsync(gr->up_ln_w, 0, W * sizeof(float), st));
    hipLaunchKernelGGL(fill_kernel, dim3((B * W + 255) / 256), dim3(256), 0, st, ws.ones, 1.0f, (size_t)B * W);
    const int ln_slots = lpad / ln_backward_rows_per_block(lpad, M, ln_backward_compute_units()), gate_slots = lpad / 64;      // partial rows per sample

    // ---- to_gs + pixel alignment ----
    GsBwdParams gb;
    gb.B = B; gb.V = V; gb.H = H; gb.W = Wd; gb.ps = m->patch; gb.lpad = lpad; gb.ng = ng; gb.C = C; gb.scene = m->scene;
    gb.relative_plk = m->relative_plk; gb.range_near = m->range_near; gb.range_far = m->range_far;
    gb.dec = sv.dec; gb.up = sv.up; gb.ray_d = a->ray_d; gb.dxyz = a->dxyz; gb.dfeatures = a->dfeatures; gb.dscaling = a->dscaling;
    gb.drotation = a->drotation; gb.dopacity = a->dopacity; gb.ddec = ws.ddec; gb.dup = ws.dup;
    DGS_TRY(launch_gaussians_backward(gb, st));

    // ---- decoder head: Linear (no bias) <- LN(weight)+modulate ----
    const float* mod_up = sv.mod + (size_t)m->layers * 6 * W;
    const float* mod_dec = mod_up + 2 * W;
    float* dmod_up = ws.dmod + (size_t)m->layers * 6 * W;
    float* dmod_dec = dmod_up + 2 * W;
    DGS_TRY(launch_transpose(ws.ddec, ND, ws.ddecT, B, lpad, ND, st));
    DGS_TRY(launch_transpose(sv.xn_dec, W, ws.actT, B, lpad, W, st));
    DGS_TRY(wgrad(ws.ddecT, ND, ws.actT, W, gr->dec_w, B, lpad, ws, stream));
    DGS_TRY(dgrad(ws.ddec, ND, mt->dec_wT, W, ws.dh, M, lpad, L, DGS_EPI_BF16, nullptr, nullptr, stream));
    LnBwdParams lb{};
    lb.rows = M; lb.width = W; lb.mod_stride = nmod; lb.rows_per_batch = lpad; lb.eps = 1e-5f; lb.x = sv.x_out; lb.dh = ws.dh;
    lb.weight = m->dec_ln_w; lb.scale = mod_dec + W; lb.dx_in = nullptr; lb.dx_out = ws.dxa; lb.dshift = dmod_dec; lb.dscale = dmod_dec + W;
    lb.dweight = gr->dec_ln_w; lb.part = ws.ln_part[0]; lb.part_stride = 3 * W;
    DGS_TRY(launch_layernorm_backward(lb, st));
    {
        const ColReduceJob jobs[2] = {{ws.ln_part[0], dmod_dec, ln_slots, 3 * W, 2 * W, B, nmod},
                                      {ws.ln_part[0] + 2 * W, gr->dec_ln_w, B * ln_slots, 3 * W, W, 1, 0}};
        DGS_TRY(launch_col_reduce(jobs, 2, st));
    }
    // ---- upsampler head (the learned-token rows) ----
    RowLinBwdParams ub{};
    ub.M = B * ng; ub.N = C; ub.K = W; ub.x = sv.upn; ub.W = m->up_w; ub.dy = ws.dup; ub.dW = gr->up_w; ub.dx = ws.dupn;
    if (ub.M > 8) return DGS_ERR_INVALID_ARGUMENT;
    DGS_TRY(launch_rowlinear_backward(ub, st));
    for (int b = 0; b < B; ++b) {
        LnBwdParams lu{};
        float* rows = ws.dxa + ((size_t)b * lpad + (L - ng)) * W;
        lu.rows = ng; lu.width = W; lu.mod_stride = nmod; lu.rows_per_batch = ng; lu.eps = 1e-5f;
        lu.x = sv.x_out + ((size_t)b * lpad + (L - ng)) * W; lu.dh = ws.dupn + (size_t)b * ng * W; lu.dh_f32 = 1; lu.weight = m->up_ln_w;
        lu.scale = mod_up + W + (size_t)b * nmod; lu.dx_in = rows; lu.dx_out = rows;
        lu.dshift = dmod_up + (size_t)b * nmod; lu.dscale = dmod_up + W + (size_t)b * nmod; lu.dweight = gr->up_ln_w;
        DGS_TRY(launch_layernorm_backward(lu, st));
    }
    // adaLN Linear of a group of modulation rows [row0, row0 + rows) of the stacked weight: dW = dmod (x) silu(cvec) and db are
    // WRITTEN here, with the group (pure streaming stores, no weight read); d cvec -- the one quantity that needs every group -- is
    // one pass over the stacked weight at the end
    auto ada_backward = [&](int row0, int rows, float* dW, float* db) -> int {
        RowLinBwdParams ra{};
        ra.M = B; ra.N = rows; ra.K = W; ra.silu_in = 1; ra.x = sv.cvec; ra.W = m->ada_w + (size_t)row0 * W; ra.dy = ws.dmod + row0; ra.ldy = nmod;
        ra.dW = dW; ra.db = db; ra.rows_per_block = ROWLINEAR_DW_ROWS;
        return launch_rowlinear_backward(ra, st);
    };
    DGS_TRY(ada_backward(6 * W * m->layers, 4 * W, gr->head_ada_w, gr->head_ada_b));
    if (a->block_done) a->block_done(a->block_user, m->layers);     // dec_w, dec_ln_w, up_w, up_ln_w, head_ada_* are final (enqueued)

    // ---- 24 x DiTBlock, last to first ----
    float* dx = ws.dxa;      // gradient w.r.t. the block output
    float* dx_mid = ws.dxb;
    for (int i = m->layers - 1; i >= 0; --i) {
        const DgsDitLayerWeights& lw = m->layer[i];
        const DgsDitLayerWeightsT& lt = mt->layer[i];
        const DgsDitLayerGrads& lg = gr->layer[i];
        const BlockSaved& k = sv.blk[i];
        const float* mod = sv.mod + (size_t)i * 6 * W;
        float* dmod = ws.dmod + (size_t)i * 6 * W;
        (void)lw;
        // recompute mode: block i's activations are rebuilt from its saved input (the last block's are still in place)
        if (a->recompute && i + 1 < m->layers)
            DGS_TRY(block_forward_train(m, i, k, ws.xre, sv.mod, sv.attn_tail, sv.attn_tail_bytes, B, lpad, L, stream));
        // MLP branch
        DGS_TRY(launch_gate_mul(dx, k.y2, mod + 5 * W, nmod, ws.dy, ws.dyT, ws.gate_part[0], B, lpad, W, st));       // + fc2_b partials
        DGS_TRY(wgrad(ws.dyT, W, k.gT, 4 * W, lg.fc2_w, B, lpad, ws, stream));
        DGS_TRY(dgrad(ws.dy, W, lt.fc2_wT, 4 * W, ws.du, M, lpad, L, DGS_EPI_DGELU_BF16, k.u, ws.duT, stream));
        DGS_TRY(launch_colsum(ws.du, 4 * W, M, 4 * W, ws.fc1b_part, st));
        DGS_TRY(launch_transpose(k.h2, W, ws.actT, B, lpad, W, st));
        DGS_TRY(wgrad(ws.duT, 4 * W, ws.actT, W, lg.fc1_w, B, lpad, ws, stream));
        DGS_TRY(dgrad(ws.du, 4 * W, lt.fc1_wT, W, ws.dh, M, lpad, L, DGS_EPI_BF16, nullptr, nullptr, stream));
        LnBwdParams l2{};
        l2.rows = M; l2.width = W; l2.mod_stride = nmod; l2.rows_per_batch = lpad; l2.eps = 1e-6f; l2.x = k.x_mid; l2.dh = ws.dh;
        l2.scale = mod + 4 * W; l2.dx_in = dx; l2.dx_out = dx_mid; l2.dshift = dmod + 3 * W; l2.dscale = dmod + 4 * W;
        l2.part = ws.ln_part[0]; l2.part_stride = 3 * W;
        DGS_TRY(launch_layernorm_backward(l2, st));
        // attention branch
        DGS_TRY(launch_gate_mul(dx_mid, k.y1, mod + 2 * W, nmod, ws.dy, ws.dyT, ws.gate_part[1], B, lpad, W, st));   // + proj_b partials
        DGS_TRY(launch_transpose(k.a, W, ws.actT, B, lpad, W, st));
        DGS_TRY(wgrad(ws.dyT, W, ws.actT, W, lg.proj_w, B, lpad, ws, stream));
        DGS_TRY(dgrad(ws.dy, W, lt.proj_wT, W, ws.da, M, lpad, L, DGS_EPI_BF16, nullptr, ws.daT, stream));
        DgsDitAttentionBackwardArgs ab{};
        ab.B = B; ab.heads = m->heads; ab.L = L; ab.lpad = lpad; ab.qkv = k.qkv; ab.qkvT = k.qkvT; ab.o = k.a; ab.dO = ws.da; ab.dOT = ws.daT;
        ab.lse2 = k.lse2; ab.D = ws.D; ab.dqkv = ws.dqkv; ab.scale = 0.125f;
        // the token-contiguous copy the qkv weight gradient reads and the bias gradient's partial rows leave the attention backward's
        // own epilogues (rounds 3-5: a transpose and a column-sum kernel behind it, 37 + 23 us per block at 4 samples); the padding
        // tokens of dqkvT are never written: zero since the workspace was allocated
        // (DGS_ATTN_BWD_BYPRODUCTS=0, measurement aid: the two kernels again -- profiles/r06_attn_bwd_byproducts_ab.txt)
        static const bool byproducts = !(getenv("DGS_ATTN_BWD_BYPRODUCTS") && atoi(getenv("DGS_ATTN_BWD_BYPRODUCTS")) == 0);
        if (byproducts) { ab.dqkvT = ws.dqkvT; ab.bias_part = ws.qkvb_part; }
        DGS_TRY(dgs_dit_attention_backward(&ab, stream));
        if (!byproducts) {
            DGS_TRY(launch_colsum(ws.dqkv, 3 * W, M, 3 * W, ws.qkvb_part, st));
            DGS_TRY(launch_transpose(ws.dqkv, 3 * W, ws.dqkvT, B, lpad, 3 * W, st));
        }
        DGS_TRY(launch_transpose(k.h1, W, ws.actT, B, lpad, W, st));
        DGS_TRY(wgrad(ws.dqkvT, 3 * W, ws.actT, W, lg.qkv_w, B, lpad, ws, stream));
        DGS_TRY(dgrad(ws.dqkv, 3 * W, lt.qkv_wT, W, ws.dh, M, lpad, L, DGS_EPI_BF16, nullptr, nullptr, stream));
        LnBwdParams l1{};
        l1.rows = M; l1.width = W; l1.mod_stride = nmod; l1.rows_per_batch = lpad; l1.eps = 1e-6f; l1.x = k.x_in; l1.dh = ws.dh;
        l1.scale = mod + W; l1.dx_in = dx_mid; l1.dx_out = dx; l1.dshift = dmod; l1.dscale = dmod + W;
        l1.part = ws.ln_part[1]; l1.part_stride = 3 * W;
        DGS_TRY(launch_layernorm_backward(l1, st));
        {   // every column sum of the block in one launch: six modulation gradients per sample, four bias gradients
            const int bias_slots = colsum_slots(M, 4 * W, 4 * W);
            const ColReduceJob jobs[8] = {
                {ws.ln_part[1], dmod, ln_slots, 3 * W, 2 * W, B, nmod},                          // shift_msa | scale_msa
                {ws.gate_part[1], dmod + 2 * W, gate_slots, 2 * W, W, B, nmod},                  // gate_msa
                {ws.ln_part[0], dmod + 3 * W, ln_slots, 3 * W, 2 * W, B, nmod},                  // shift_mlp | scale_mlp
                {ws.gate_part[0], dmod + 5 * W, gate_slots, 2 * W, W, B, nmod},                  // gate_mlp
                {ws.gate_part[0] + W, lg.fc2_b, B * gate_slots, 2 * W, W, 1, 0},
                {ws.gate_part[1] + W, lg.proj_b, B * gate_slots, 2 * W, W, 1, 0},
                {ws.fc1b_part, lg.fc1_b, bias_slots, 4 * W, 4 * W, 1, 0},
                {ws.qkvb_part, lg.qkv_b, byproducts ? B * dgs_dit_attention_backward_slots(L) : colsum_slots(M, 3 * W, 3 * W), 3 * W, 3 * W, 1, 0}};
            DGS_TRY(launch_col_reduce(jobs, 8, st));
        }
        DGS_TRY(ada_backward(6 * W * i, 6 * W, lg.ada_w, lg.ada_b));       // the block's six modulation gradients are complete
        if (a->block_done) a->block_done(a->block_user, i);            // block i's eight gradients are final (enqueued)
    }

    // ---- input LayerNorm, learned tokens, tokenizer weight ----
    LnBwdParams li{};
    li.rows = M; li.width = W; li.rows_per_batch = lpad; li.eps = 1e-5f; li.x = sv.x0_pre; li.dh = dx; li.dh_f32 = 1; li.weight = m->in_ln_w;
    li.dx_out = dx_mid; li.dweight = gr->in_ln_w; li.part = ws.ln_part[0]; li.part_stride = 3 * W;
    DGS_TRY(launch_layernorm_backward(li, st));
    {
        const ColReduceJob job{ws.ln_part[0] + 2 * W, gr->in_ln_w, B * ln_slots, 3 * W, W, 1, 0};
        DGS_TRY(launch_col_reduce(&job, 1, st));
    }
    hipLaunchKernelGGL(pos_embed_backward_kernel, dim3((ng * W + 255) / 256), dim3(256), 0, st, dx_mid, gr->pos_emb, B, lpad, L, ng, W);
    // bf16 + token-contiguous copy of dx0 via gate_mul with a gate of ones (its column-sum by-products stay in the slab)
    DGS_TRY(launch_gate_mul(dx_mid, sv.xn_dec, ws.ones, W, ws.dy, ws.dyT, ws.gate_part[0], B, lpad, W, st));
    DGS_TRY(launch_transpose(sv.emb, kin, ws.embT, B, lpad, kin, st));
    DGS_TRY(wgrad(ws.dyT, W, ws.embT, kin, gr->tok_w, B, lpad, ws, stream));   // kin = 576: 64-column tiles

    // ---- d cvec: all modulation rows (blocks + heads) against the stacked adaLN weight; then the TimestepEmbedder ----
    {
        RowLinBwdParams ra{};
        ra.M = B; ra.N = nmod; ra.K = W; ra.silu_in = 1; ra.x = sv.cvec; ra.W = m->ada_w; ra.dy = ws.dmod; ra.dx = ws.dcvec; ra.part = ws.rl_part;
        DGS_TRY(launch_rowlinear_backward(ra, st));
        const ColReduceJob job{ws.rl_part, ws.dcvec, (nmod + ROWLINEAR_BWD_ROWS - 1) / ROWLINEAR_BWD_ROWS, B * W, B * W, 1, 0};
        DGS_TRY(launch_col_reduce(&job, 1, st));
    }
    RowLinBwdParams r1{};
    r1.M = B; r1.N = W; r1.K = W; r1.silu_in = 1; r1.x = sv.c1; r1.W = m->t_w1; r1.dy = ws.dcvec; r1.dW = gr->t_w1; r1.db = gr->t_b1; r1.dx = ws.dc1;
    r1.part = ws.rl_part;
    DGS_TRY(launch_rowlinear_backward(r1, st));
    {
        const ColReduceJob job{ws.rl_part, ws.dc1, (W + ROWLINEAR_BWD_ROWS - 1) / ROWLINEAR_BWD_ROWS, B * W, B * W, 1, 0};
        DGS_TRY(launch_col_reduce(&job, 1, st));
    }
    RowLinBwdParams r0{};
    r0.M = B; r0.N = W; r0.K = 256; r0.silu_in = 0; r0.x = sv.temb; r0.W = m->t_w0; r0.dy = ws.dc1; r0.dW = gr->t_w0; r0.db = gr->t_b0;
    DGS_TRY(launch_rowlinear_backward(r0, st));
    if (a->block_done) a->block_done(a->block_user, -1);               // everything else
    return hipGetLastError() == hipSuccess ? DGS_OK : DGS_ERR_DEVICE;
}
