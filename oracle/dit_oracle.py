"""CPU ORACLE for the denoiser half of the hot path: plain PyTorch fp32 restatement of
`DGSDenoiser.image_to_gaussians` and everything it calls.

TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).

Follows (paths relative to /root/reference/diffusionGS/):
  models/denoiser/denoiser.py:21-22 (modulate), :26-72 (TimestepEmbedder), :76-136 (GaussiansUpsampler),
      :139-164 (ImageTokenDecoder), :199-251 (parameters), :306-416 (image_to_gaussians)
  models/denoiser/denoiser_scene.py:232-263,407-418 (scene variant: [1,2,W] pos-embedding, depth range)
  models/transformers/utils_transformer.py:246-290 (DiTBlock)
  third party, not vendored: timm==0.9.16 (requirement.txt:27) `Attention` / `Mlp`
      (timm/models/vision_transformer.py, timm/layers/mlp.py) restated in `attention()` / `mlp()`:
      qkv Linear(bias) -> [B,N,3,h,d].permute(2,0,3,1,4) -> softmax(q k^T / sqrt(d)) v -> proj Linear(bias);
      fc1 -> GELU(tanh) -> fc2.
  systems/utils.py:621-684,751-757 (TransformInput rays)

PINNING: tests/golden/dit_golden_*.npz are produced by oracle/make_dit_golden.py, which imports the reference's
own denoiser.py / utils_transformer.py (third-party deps stubbed) -- tests/test_dit_oracle.py checks this
restatement against them.  State-dict keys are the reference's (SURVEY.md section 5 "checkpoint").
"""
import math

import torch
import torch.nn.functional as F


class Cfg:
    """Operator-surface config (denoiser.py:174-196 + denoiser_scene.py:202-204)."""

    def __init__(self, width=1024, in_channels=9, patch_size=8, n_gaussians=2, dim_heads=64, num_layers=24,
                 ray_pe_type="relative_plk", gaussians_sh_degree=0, scene=False, range_near=0.0, range_far=500.0):
        self.width, self.in_channels, self.patch_size, self.n_gaussians = width, in_channels, patch_size, n_gaussians
        self.dim_heads, self.num_layers, self.ray_pe_type = dim_heads, num_layers, ray_pe_type
        self.gaussians_sh_degree, self.scene, self.range_near, self.range_far = gaussians_sh_degree, scene, range_near, range_far

    @property
    def gs_channels(self):
        return 3 + (self.gaussians_sh_degree + 1) ** 2 * 3 + 3 + 4 + 1


def init_state_dict(cfg, seed=0, dtype=torch.float32):
    """Random weights with the reference initialisers (utils_transformer.py:30-36 applied at denoiser.py:223,246-251;
    t_embedder :205-206; pos-embedding trunc_normal :231; LayerNorm weight 1; other biases 0; t_embedder biases keep
    nn.Linear's default uniform init)."""
    g = torch.Generator().manual_seed(seed)
    W = cfg.width
    n = lambda *s: torch.randn(*s, generator=g, dtype=dtype) * 0.02
    sd = {}
    for i, (o, k) in enumerate([(W, 256), (W, W)]):
        sd[f"t_embedder.mlp.{2 * i}.weight"] = n(o, k)
        b = 1.0 / math.sqrt(k)
        sd[f"t_embedder.mlp.{2 * i}.bias"] = (torch.rand(o, generator=g, dtype=dtype) * 2 - 1) * b
    sd["image_tokenizer.1.weight"] = n(W, cfg.in_channels * cfg.patch_size ** 2)
    pe = torch.nn.init.trunc_normal_(torch.empty(cfg.n_gaussians, W, dtype=dtype), std=0.02, generator=g)
    sd["gaussians_pos_embedding"] = pe[None] if cfg.scene else pe
    sd["transformer_input_layernorm.weight"] = torch.ones(W, dtype=dtype)
    for i in range(cfg.num_layers):
        p = f"transformer.{i}."
        sd[p + "attn.qkv.weight"] = n(3 * W, W); sd[p + "attn.qkv.bias"] = torch.zeros(3 * W, dtype=dtype)
        sd[p + "attn.proj.weight"] = n(W, W); sd[p + "attn.proj.bias"] = torch.zeros(W, dtype=dtype)
        sd[p + "mlp.fc1.weight"] = n(4 * W, W); sd[p + "mlp.fc1.bias"] = torch.zeros(4 * W, dtype=dtype)
        sd[p + "mlp.fc2.weight"] = n(W, 4 * W); sd[p + "mlp.fc2.bias"] = torch.zeros(W, dtype=dtype)
        sd[p + "adaLN_modulation.1.weight"] = n(6 * W, W); sd[p + "adaLN_modulation.1.bias"] = torch.zeros(6 * W, dtype=dtype)
    for head, out in (("upsampler", cfg.gs_channels), ("image_token_decoder", cfg.patch_size ** 2 * cfg.gs_channels)):
        sd[head + ".layernorm.weight"] = torch.ones(W, dtype=dtype)
        sd[head + ".linear.weight"] = n(out, W)
        sd[head + ".adaLN_modulation.1.weight"] = n(2 * W, W)
        sd[head + ".adaLN_modulation.1.bias"] = torch.zeros(2 * W, dtype=dtype)
    return sd


def parity_state_dict(cfg, seed=0):
    """Seeded weights for parity tests: reference initialisers, then every bias / LayerNorm weight perturbed (the
    reference initialises biases to 0, which would hide bias-path bugs) and every GEMM weight rounded to a
    bf16-representable value (the HIP path stores GEMM weights in bf16; tolerances then only cover activation rounding).
    Deterministic given (cfg, seed) on the same torch build -- golden fixtures store the seed, not the weights."""
    sd = init_state_dict(cfg, seed)
    g = torch.Generator().manual_seed(seed + 1000)
    for k in sorted(sd):
        if k.endswith("bias") or k.endswith("layernorm.weight"):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
        elif k.endswith("weight") and sd[k].dim() == 2:
            sd[k] = sd[k].to(torch.bfloat16).float()
    return sd


def timestep_embedding(t, dim=256, max_period=10000):
    """denoiser.py:46-67"""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half).to(t.device)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def t_embed(sd, t):
    """TimestepEmbedder.forward, denoiser.py:69-72"""
    h = F.linear(timestep_embedding(t), sd["t_embedder.mlp.0.weight"], sd["t_embedder.mlp.0.bias"])
    return F.linear(F.silu(h), sd["t_embedder.mlp.2.weight"], sd["t_embedder.mlp.2.bias"])


def modulate(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


def attention(x, sd, p, num_heads):
    """timm 0.9.16 Attention.forward (qk_norm off, dropout 0)."""
    B, N, C = x.shape
    d = C // num_heads
    qkv = F.linear(x, sd[p + "qkv.weight"], sd[p + "qkv.bias"]).reshape(B, N, 3, num_heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv.unbind(0)
    attn = (q * d ** -0.5) @ k.transpose(-2, -1)
    attn = attn.softmax(dim=-1)
    o = (attn @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[p + "proj.weight"], sd[p + "proj.bias"])


def mlp(x, sd, p):
    """timm 0.9.16 Mlp.forward with act = GELU(approximate='tanh') (utils_transformer.py:259-265)."""
    h = F.gelu(F.linear(x, sd[p + "fc1.weight"], sd[p + "fc1.bias"]), approximate="tanh")
    return F.linear(h, sd[p + "fc2.weight"], sd[p + "fc2.bias"])


def dit_block(x, c, sd, p, num_heads):
    """DiTBlock.forward, utils_transformer.py:271-290"""
    W = x.shape[-1]
    mod = F.linear(F.silu(c), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"])
    sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
    x = x + g_a.unsqueeze(1) * attention(modulate(F.layer_norm(x, (W,), eps=1e-6), sh_a, sc_a), sd, p + "attn.", num_heads)
    x = x + g_m.unsqueeze(1) * mlp(modulate(F.layer_norm(x, (W,), eps=1e-6), sh_m, sc_m), sd, p + "mlp.")
    return x


def head(tokens, c, sd, p):
    """GaussiansUpsampler.forward / ImageTokenDecoder.forward, denoiser.py:122-136,155-164"""
    W = tokens.shape[-1]
    shift, scale = F.linear(F.silu(c), sd[p + "adaLN_modulation.1.weight"], sd[p + "adaLN_modulation.1.bias"]).chunk(2, dim=1)
    h = modulate(F.layer_norm(tokens, (W,), sd[p + "layernorm.weight"], None, 1e-5), shift, scale)
    return F.linear(h, sd[p + "linear.weight"])


def patchify(posed, ps):
    """Rearrange 'b v c (hh ph) (ww pw) -> (b v) (hh ww) (ph pw c)', denoiser.py:211-215"""
    b, v, c, H, W = posed.shape
    x = posed.reshape(b, v, c, H // ps, ps, W // ps, ps).permute(0, 1, 3, 5, 4, 6, 2)
    return x.reshape(b * v, (H // ps) * (W // ps), ps * ps * c)


def image_to_gaussians(sd, cfg, images, ray_o, ray_d, t, return_tokens=False, checkpoint_blocks=False):
    """denoiser.py:306-416 (obj) / denoiser_scene.py mirror.  Returns (dict(xyz, features, scaling, rotation, opacity),
    img_aligned_xyz[b,v,3,H,W]).  checkpoint_blocks: every block under torch.utils.checkpoint, as the reference itself runs them
    (denoiser.py:348-354) -- same values and gradients, one block's activations alive at a time."""
    if cfg.ray_pe_type == "relative_plk":
        o_dot_d = torch.sum(-ray_o * ray_d, dim=2, keepdim=True)
        nearest = ray_o + o_dot_d * ray_d
        posed = torch.cat([images[:, :, :3] * 2.0 - 1.0, ray_d, nearest], dim=2)
    else:
        posed = torch.cat([images[:, :, :3] * 2.0 - 1.0, torch.cross(ray_o, ray_d, dim=2), ray_d], dim=2)
    b, v, c, h, w = posed.shape
    ps = cfg.patch_size
    tok = F.linear(patchify(posed, ps), sd["image_tokenizer.1.weight"])
    cvec = t_embed(sd, t)
    n_patches = tok.shape[1]
    tok = tok.reshape(b, v * n_patches, cfg.width)
    pe = sd["gaussians_pos_embedding"]
    pe = pe.expand(b, -1, -1) if pe.dim() == 3 else pe[None].expand(b, -1, -1)
    x = torch.cat((pe, tok), dim=1)
    x = F.layer_norm(x, (cfg.width,), sd["transformer_input_layernorm.weight"], None, 1e-5)
    heads = cfg.width // cfg.dim_heads
    for i in range(cfg.num_layers):
        if checkpoint_blocks and torch.is_grad_enabled():
            from torch.utils.checkpoint import checkpoint
            x = checkpoint(dit_block, x, cvec, sd, f"transformer.{i}.", heads, use_reentrant=False)
        else:
            x = dit_block(x, cvec, sd, f"transformer.{i}.", heads)
    g_tok, i_tok = x.split([cfg.n_gaussians, v * n_patches], dim=1)
    gaussians = head(g_tok, cvec, sd, "upsampler.")
    img_g = head(i_tok, cvec, sd, "image_token_decoder.").reshape(b, -1, cfg.gs_channels)
    n_img = img_g.shape[1]
    allg = torch.cat((gaussians, img_g), dim=1)
    # to_gs, denoiser.py:103-120
    M = (cfg.gaussians_sh_degree + 1) ** 2
    xyz, features, scaling, rotation, opacity = allg.split([3, M * 3, 3, 4, 1], dim=2)
    features = features.reshape(b, -1, M, 3)
    scaling = (scaling - 2.3).clamp(max=-1.20)
    opacity = opacity - 2.0
    # pixel alignment, denoiser.py:370-406 : '(v h w ph pw) c -> v c (h ph) (w pw)'
    ix = xyz[:, -n_img:, :].reshape(b, v, h // ps, w // ps, ps, ps, 3).permute(0, 1, 6, 2, 4, 3, 5).reshape(b, v, 3, h, w)
    depth = ix.mean(dim=2, keepdim=True)
    if cfg.scene:
        depth = torch.sigmoid(depth) * (cfg.range_far - cfg.range_near) + cfg.range_near   # denoiser_scene.py:262
    else:
        depth = torch.sigmoid(depth)
        if cfg.ray_pe_type == "relative_plk":
            depth = (2.0 * depth - 1.0) * 1.8 + o_dot_d
    aligned = ray_o + depth * ray_d
    al = aligned.reshape(b, v, 3, h // ps, ps, w // ps, ps).permute(0, 1, 3, 5, 4, 6, 2).reshape(b, -1, 3)
    xyz = torch.cat((xyz[:, :-n_img, :], al), dim=1)
    out = dict(xyz=xyz, features=features, scaling=scaling, rotation=rotation, opacity=opacity)
    if return_tokens:
        out["tokens"] = x
    return out, aligned


def transform_input_rays(c2w, fxfycxcy, h, w):
    """TransformInput, systems/utils.py:621-684,751-757: c2w [b,v,4,4], fxfycxcy [b,v,4] -> ray_o, ray_d [b,v,3,h,w]."""
    b, v = c2w.shape[:2]
    c2w = c2w.reshape(b * v, 4, 4)
    k = fxfycxcy.reshape(b * v, 4)
    y, x = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    x = x[None].expand(b * v, -1, -1).reshape(b * v, -1).to(c2w)
    y = y[None].expand(b * v, -1, -1).reshape(b * v, -1).to(c2w)
    x = (x + 0.5 - k[:, 2:3]) / k[:, 0:1]
    y = (y + 0.5 - k[:, 3:4]) / k[:, 1:2]
    d = torch.stack([x, y, torch.ones_like(x)], dim=2)
    d = torch.bmm(d, c2w[:, :3, :3].transpose(1, 2))
    d = d / torch.norm(d, dim=2, keepdim=True)
    o = c2w[:, :3, 3][:, None, :].expand_as(d)
    rs = lambda a: a.reshape(b, v, h, w, 3).permute(0, 1, 4, 2, 3)
    return rs(o), rs(d)


def camera_matrices(c2w, fxfycxcy, h, w, znear=0.01, zfar=100.0):
    """Camera, models/gsrenderer/gs_core.py:277-316, batched over leading dims.
    Returns viewmatrix [...,4,4] (= W2C^T), projmatrix [...,4,4] (= W2C^T P^T), campos [...,3], tanfov [...,2]."""
    c2w = c2w.float()
    w2c = torch.linalg.inv(c2w)
    fx, fy, cx, cy = fxfycxcy.unbind(-1)
    P = torch.zeros(c2w.shape[:-2] + (4, 4), dtype=torch.float32, device=c2w.device)
    P[..., 0, 0] = 2 * fx / w
    P[..., 1, 1] = 2 * fy / h
    P[..., 0, 2] = 2 * (cx / w) - 1
    P[..., 1, 2] = 2 * (cy / h) - 1
    P[..., 2, 2] = -(zfar + znear) / (zfar - znear)
    P[..., 3, 2] = 1.0
    P[..., 2, 3] = -(2 * zfar * znear) / (zfar - znear)
    view = w2c.transpose(-1, -2)
    proj = view @ P.transpose(-1, -2)
    tanfov = torch.stack([w / (2 * fx), h / (2 * fy)], dim=-1)
    return view.contiguous(), proj.contiguous(), c2w[..., :3, 3].contiguous(), tanfov.contiguous()
