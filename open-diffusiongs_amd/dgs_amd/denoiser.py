"""Operator surface of the reference's denoiser models on top of the gfx950 kernels.

Mirrors (names, Config fields, method signatures, state-dict keys) of
  /root/reference/diffusionGS/models/denoiser/denoiser.py:166-447        DGSDenoiser  "diffusion-gs-model"
  /root/reference/diffusionGS/models/denoiser/denoiser_scene.py:171-457  DGSDenoiser  "diffusion-gs-model-scene"
  /root/reference/diffusionGS/models/gsrenderer/renderer.py:20-92        Renderer
  /root/reference/diffusionGS/models/gsrenderer/gs_core.py:321-373,544-570  GaussianModel (set_data / get_* only)
so the callers (systems/diffusion_gs_system.py:90-92, gaussian_diffusion.py:350,359, pipline_obj.py:298-308) can sit on
top unchanged.  The parameters live in ordinary nn.Parameters under the reference's key names (checkpoints load with
load_state_dict); compute is ONE C call into libdgs_hip.so for the DiT (dgs_amd.dit.DitEngine) and one batched C call
for all (sample, view) rasterizations (dgs_amd.raster).  There is no PyTorch fallback.
"""
import copy
import math
from dataclasses import dataclass, fields

import torch
import torch.nn as nn

from .dit import DitEngine
from .raster import default_backend

_REGISTRY = {}


def register(name):
    def deco(cls):
        _REGISTRY[name] = cls
        return cls
    return deco


def find(name):
    """diffusionGS.find (diffusionGS/__init__.py:6-31)."""
    return _REGISTRY[name]


class AttrDict(dict):
    """easydict.EasyDict stand-in (the reference returns edict(xyz=..., ...), denoiser.py:414)."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__ = self


class GaussianModel:
    """Container + activations of gs_core.py:321-373,544-570 (exp / normalize / sigmoid)."""

    def __init__(self, sh_degree, scaling_modifier=None):
        self.sh_degree, self.scaling_modifier = sh_degree, scaling_modifier
        self.empty()

    def empty(self):
        self._xyz = self._features_dc = self._scaling = self._rotation = self._opacity = torch.empty(0)
        self._features_rest = torch.empty(0) if self.sh_degree > 0 else None

    def set_data(self, xyz, features, scaling, rotation, opacity):
        self._xyz = xyz
        self._features_dc = features[:, :1, :].contiguous()
        self._features_rest = features[:, 1:, :].contiguous() if self.sh_degree > 0 else None
        self._scaling, self._rotation, self._opacity = scaling, rotation, opacity
        return self

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        s = torch.exp(self._scaling)
        return s * self.scaling_modifier if self.scaling_modifier is not None else s

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return self._features_dc if self._features_rest is None else torch.cat((self._features_dc, self._features_rest), dim=1)

    # -- on-disk format, gs_core.py:578-760 (dgs_amd/consumers.py) --
    def construct_dtypes(self, use_fp16=False, enable_gs_viewer=True):
        from . import consumers
        assert not use_fp16, "PLY has no 16-bit float property type"
        return consumers.construct_dtypes(self, enable_gs_viewer)

    def save_ply(self, path, use_fp16=False, enable_gs_viewer=True, color_code=False, filter_mask=None):
        from . import consumers
        assert not use_fp16 and not color_code
        consumers.save_ply(self, path, enable_gs_viewer, filter_mask)

    def load_ply(self, path, device="cpu"):
        from . import consumers
        return consumers.load_ply(self, path, device)


class Renderer(nn.Module):
    """renderer.py:20-92.  forward(...) -> [b, v, 3, H, W] float32; all b*v views in one launch sequence."""

    def __init__(self, config, backend=None):
        super().__init__()
        self.config = config
        self._backend = backend          # None -> the product library (libdgs_hip.so); tests inject the CPU emulation build
        self.scaling_modifier = None
        self.gaussians_model = GaussianModel(config.gaussians_sh_degree, self.scaling_modifier)

    def backend(self):
        return self._backend if self._backend is not None else default_backend()

    def forward(self, xyz, features, scaling, rotation, opacity, height, width, C2W, fxfycxcy, deferred=True):
        f = lambda t: t.float()      # custom_fwd(cast_inputs=float32), renderer.py:34
        backend = self.backend()
        if torch.is_grad_enabled() and any(t.requires_grad for t in (xyz, features, scaling, rotation, opacity)):
            from .raster import render_views_autograd      # training: DeferredGaussianRender's role, gs_core.py:949-1064
            return render_views_autograd(backend, f(xyz), f(features), f(scaling), f(rotation), f(opacity), height, width,
                                         f(C2W), f(fxfycxcy))
        return backend.render_views(f(xyz), f(features), f(scaling), f(rotation), f(opacity), height, width,
                                              f(C2W), f(fxfycxcy))


SceneRenderer = Renderer


class _Linear(nn.Module):
    """Parameter holder with nn.Linear's state-dict keys (weight/bias); never called -- the GEMM runs in HIP."""

    def __init__(self, i, o, bias=True):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.zeros(o)) if bias else None


class _LNWeight(nn.Module):
    def __init__(self, w):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(w))


def _seq(*mods):
    """nn.Sequential keeps the reference's numeric sub-keys ('mlp.0', 'adaLN_modulation.1', 'image_tokenizer.1')."""
    return nn.Sequential(*mods)


class _Block(nn.Module):
    def __init__(self, w):
        super().__init__()
        self.attn = nn.Module(); self.attn.qkv = _Linear(w, 3 * w); self.attn.proj = _Linear(w, w)
        self.mlp = nn.Module(); self.mlp.fc1 = _Linear(w, 4 * w); self.mlp.fc2 = _Linear(4 * w, w)
        self.adaLN_modulation = _seq(nn.Identity(), _Linear(w, 6 * w))


class _Head(nn.Module):
    def __init__(self, w, out):
        super().__init__()
        self.layernorm = _LNWeight(w)
        self.linear = _Linear(w, out, bias=False)
        self.adaLN_modulation = _seq(nn.Identity(), _Linear(w, 2 * w))


def _aligned_to_points(d_aligned, ps):
    """Inverse of the reference's "b (v h w ph pw) c -> b v c (h ph) (w pw)" rearrange (denoiser.py:371-379,401-406)."""
    B, V, C, H, W = d_aligned.shape
    x = d_aligned.reshape(B, V, C, H // ps, ps, W // ps, ps).permute(0, 1, 3, 5, 4, 6, 2)
    return x.reshape(B, V * H * W, C)


class _ArenaGuard:
    """Holds one of the engine's activation arenas between a training forward and its backward; dies with the autograd graph."""

    def __init__(self, eng, arena):
        self.eng, self.arena = eng, arena

    def release(self):
        if self.arena is not None:
            self.eng.release_arena(self.arena)
            self.arena = None

    __del__ = release


class _DitFunction(torch.autograd.Function):
    """image_to_gaussians under torch autograd: forward = dgs_dit_forward_train (activations saved in one of the engine's arenas;
    per-block recompute when the module's checkpoint policy says so, denoiser.py:348-354), backward = dgs_dit_backward.
    Several forwards may be pending (each holds its own arena until its backward has run or its graph is dropped).
    The parameter gradients land in the engine's flat fp32 buffer (dgs_amd.parallel.FlatGrads); autograd receives COPIES
    of its slices in named_parameters() order (so .grad accumulation over several backward passes is correct), unless a
    trainer owns the buffer (module._grads_in_place: the .grad tensors ARE views of it and nothing is copied)."""

    @staticmethod
    def forward(ctx, module, names, images, ray_o, ray_d, t, *params):
        eng = module.engine()
        B, V, _, H, W = images.shape
        out, aligned = eng.forward_train(images, ray_o, ray_d, t, recompute=module.recompute_policy(B, V, H, W), keep_busy=True)
        ctx.guard = _ArenaGuard(eng, eng._train["current"])     # released by backward, or when the graph (and with it ctx) is dropped unused
        ctx.engine, ctx.module, ctx.names, ctx.param_shapes = eng, module, names, [tuple(p.shape) for p in params]
        return out["xyz"], out["features"], out["scaling"], out["rotation"], out["opacity"], aligned

    @staticmethod
    def backward(ctx, dxyz, dfeat, dscal, drot, dopa, daligned):
        eng, module = ctx.engine, ctx.module
        z = lambda g, like: g if g is not None else torch.zeros(like, device=eng.device)
        arena = ctx.guard.arena
        if arena is None:
            raise RuntimeError("DGSDenoiser: backward through the same forward twice (its activations were released by the first)")
        B, V, H, W = arena["shape"]
        P = eng.ng + V * H * W
        dxyz = z(dxyz, (B, P, 3))
        if daligned is not None:      # img_aligned_xyz is a rearranged view of xyz[:, n_gaussians:] (denoiser.py:401-409)
            dxyz = dxyz.clone()
            dxyz[:, eng.ng:] += _aligned_to_points(daligned.to(dxyz.dtype), eng.patch)
        try:
            eng.backward(dxyz, z(dfeat, (B, P, 1, 3)), z(dscal, (B, P, 3)), z(drot, (B, P, 4)), z(dopa, (B, P, 1)),
                         block_hook=module._block_hook, arena=arena)
        finally:
            ctx.guard.release()
        if module._grads_in_place:
            return (None,) * (6 + len(ctx.names))
        views = eng.grad_views()
        grads = tuple(views[n].reshape(shape).clone() for n, shape in zip(ctx.names, ctx.param_shapes))
        return (None, None, None, None, None, None) + grads


@register("diffusion-gs-model")
class DGSDenoiser(nn.Module):
    SCENE = False

    @dataclass
    class Config:   # denoiser.py:174-196 (+ denoiser_scene.py:202-204)
        pretrained_model_name_or_path: str = ""
        use_downsample: bool = False
        num_latents: int = 256
        width: int = 1024
        in_channels: int = 3
        patch_size: int = 16
        n_gaussians: int = 2
        dim_heads: int = 64
        num_layers: int = 24
        ray_pe_type: str = "relative_plk"
        hard_pixelalign: bool = True
        clip_xyz: bool = True
        gaussians_sh_degree: int = 0
        use_gssplat: bool = False
        prior_distribution: str = "gaussian"
        use_flash: bool = False
        use_checkpoint: bool = True
        grad_checkpoint_every: int = 1
        range_setting_type: str = "sigmoid"
        range_setting_near: float = 0.0
        range_setting_far: float = 500.0

    def __init__(self, cfg=None, device="cuda", lib=None):
        super().__init__()
        self._lib = lib                  # None -> the product library; tests inject the CPU emulation build
        known = {f.name for f in fields(self.Config)}
        self.cfg = self.Config(**{k: v for k, v in dict(cfg or {}).items() if k in known})
        c, w = self.cfg, self.cfg.width
        if not c.hard_pixelalign:
            raise NotImplementedError("only hard_pixelalign=True (every shipped config) is implemented")
        self.device = torch.device(device)
        self.t_embedder = nn.Module()
        self.t_embedder.mlp = _seq(_Linear(256, w), nn.Identity(), _Linear(w, w))
        self.image_tokenizer = _seq(nn.Identity(), _Linear(c.in_channels * c.patch_size ** 2, w, bias=False))
        self.gaussians_pos_embedding = nn.Parameter(torch.empty((1, c.n_gaussians, w) if self.SCENE else (c.n_gaussians, w)))
        self.transformer_input_layernorm = _LNWeight(w)
        self.transformer = nn.ModuleList([_Block(w) for _ in range(c.num_layers)])
        gs_ch = 3 + (c.gaussians_sh_degree + 1) ** 2 * 3 + 3 + 4 + 1
        self.upsampler = _Head(w, gs_ch)
        self.image_token_decoder = _Head(w, c.patch_size ** 2 * gs_ch)
        from .raster import RasterBackend
        self.gs_renderer = Renderer(c, backend=RasterBackend(lib) if lib is not None else None)
        self.reset_parameters()
        self._engine, self._engine_version = None, None
        self._block_hook = None          # set by a data-parallel trainer: called per finished gradient group during backward
        self._grads_in_place = False     # set by a trainer that made the .grad tensors views of the flat gradient buffer
        self.activation_budget_bytes = None   # None: 60 % of the device's free memory when the first training forward runs
        self._graphs = {}                # shape key -> dgs_amd.graph.GraphedForward
        if c.pretrained_model_name_or_path:
            self._load_pretrained(c.pretrained_model_name_or_path)

    # -- initialisers of the reference (utils_transformer.py:30-36; denoiser.py:205-206,223,231,246-251) --------
    def reset_parameters(self, seed=None):
        g = torch.Generator().manual_seed(seed) if seed is not None else None
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("layernorm.weight"):
                    p.fill_(1.0)
                elif name == "gaussians_pos_embedding":
                    nn.init.trunc_normal_(p, std=0.02, generator=g)
                elif name.endswith(".weight"):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
                elif name.startswith("t_embedder") and name.endswith(".bias"):
                    bound = 1.0 / math.sqrt(256 if ".0." in name else self.cfg.width)   # nn.Linear default bias init
                    p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
                else:
                    p.zero_()

    def _load_pretrained(self, path):   # denoiser.py:256-282
        """The checkpoint layouts the reference reads: {'model': {'denoiser.<key>': ...}} (its live branch, :259-267), a Lightning
        checkpoint {'state_dict': {'shape_model.<key>': ...}} (what its training writes and pipline_obj.py:66-70 loads at the system
        level; the branch is commented out at :269-280) and a flat state dict (optionally 'shape_model.'-prefixed); always strict."""
        ckpt = torch.load(path, map_location="cpu") if not isinstance(path, dict) else path
        if "model" in ckpt:
            ckpt = {k.replace("denoiser.", ""): v for k, v in ckpt["model"].items()
                    if k.startswith("denoiser.") and not k.startswith("denoiser.loss_computer")}
        elif "state_dict" in ckpt:
            ckpt = {k[len("shape_model."):]: v for k, v in ckpt["state_dict"].items() if k.startswith("shape_model.")}
        elif any(k.startswith("shape_model.") for k in ckpt):
            ckpt = {k[len("shape_model."):]: v for k, v in ckpt.items() if k.startswith("shape_model.")}
        self.load_state_dict(ckpt, strict=True)

    # -- engine -------------------------------------------------------------------------------------------
    def engine(self):
        """The device-resident bf16 copy of the parameters + workspaces.  Built once; when parameter versions change
        (optimizer.step(), load_state_dict) the new values are copied INTO the existing buffers (no reallocation of the
        weights, the activation arenas or the flat gradient buffer).  Updates that bypass version counters (`p.data = ...`,
        an EMA swap) need an explicit `refresh_engine_weights()`."""
        version = tuple(p._version for p in self.parameters())
        if self._engine is None:
            c = self.cfg
            self._engine = DitEngine(self.state_dict(), width=c.width, patch_size=c.patch_size, n_gaussians=c.n_gaussians,
                                     dim_heads=c.dim_heads, num_layers=c.num_layers, in_channels=c.in_channels,
                                     ray_pe_type=c.ray_pe_type, gaussians_sh_degree=c.gaussians_sh_degree, scene=self.SCENE,
                                     range_near=c.range_setting_near, range_far=c.range_setting_far, device=self.device,
                                     lib=self._lib)
        elif version != self._engine_version:
            self._engine.refresh_weights(self.state_dict())
        self._engine_version = version
        return self._engine

    def refresh_engine_weights(self):
        if self._engine is not None:
            self._engine.refresh_weights(self.state_dict())
            self._engine_version = tuple(p._version for p in self.parameters())

    def recompute_policy(self, B, V, H, W):
        """Per-block activation recompute (the reference's torch.utils.checkpoint(run_layers(i, i+1)), denoiser.py:348-354,
        `use_checkpoint` / `grad_checkpoint_every: 1`) costs a fourth forward per step; MI355X has 288 GB, so it is only
        used when saving every activation would not fit: use_checkpoint=False never recomputes; otherwise recompute iff the
        save-all arena exceeds `activation_budget_bytes`."""
        if not self.cfg.use_checkpoint:
            return False
        # grad_checkpoint_every = k groups k blocks per torch.utils.checkpoint call in the reference (denoiser.py:343-354): the gradients
        # are those of the plain graph for every k, only the memory differs.  The engine's recompute mode keeps EVERY block's input
        # (4 W bytes per token and block) and re-runs one block at a time: for k > 1 that is (k - 1) / k of the block inputs more than
        # the reference keeps and the same arithmetic, so every k >= 1 maps onto it
        if int(self.cfg.grad_checkpoint_every) < 1:
            raise ValueError("grad_checkpoint_every must be >= 1")
        eng = self.engine()
        need = eng.saved_bytes(B, V, H, W, recompute=False)
        budget = self.activation_budget_bytes
        if budget is None:
            if self.device.type == "cuda":
                tr = eng._train or {}
                have = tr["saved"].numel() if tr.get("saved") is not None and not tr.get("recompute") else 0
                budget = int(0.6 * (torch.cuda.mem_get_info(self.device)[0] + have))
            else:
                budget = 1 << 62
        return need > budget

    # -- reference surface ------------------------------------------------------------------------------------
    def forward(self, input_batch, timesteps):   # denoiser.py:284-287
        params, _ = self.image_to_gaussians(input_batch["image"], input_batch["ray_o"], input_batch["ray_d"], timesteps)
        rendered = self.render_gaussians(params, input_batch["c2w"], input_batch["fxfycxcy"], input_batch["image"].shape[3],
                                         input_batch["image"].shape[4])
        return rendered, self.prepare_to_save(params)

    def graphed(self, input_batch, timesteps):
        """`forward` at these shapes as one captured hipGraph (dgs_amd/graph.py): built on first use per shape (two eager warm-up
        calls + the capture), then `graphed(batch, t)(batch, t)` copies the inputs into the graph's tensors and replays.  Inference
        only; outputs are the graph's own tensors (overwritten by the next replay)."""
        from .graph import GraphedForward
        key = GraphedForward.shape_key(input_batch, timesteps)
        g = self._graphs.get(key)
        if g is None:
            self.engine()                    # weights up to date before anything is captured
            g = self._graphs[key] = GraphedForward(self, input_batch, timesteps)
        else:
            self.engine()                    # parameter versions moved (optimizer step, load_state_dict): copies refreshed in place
        return g

    def drop_graphs(self):
        self._graphs = {}

    def prepare_to_save(self, gaussians_parameters):   # denoiser.py:290-304
        out = []
        for b in range(gaussians_parameters.xyz.size(0)):
            self.gs_renderer.gaussians_model.empty()
            gm = copy.deepcopy(self.gs_renderer.gaussians_model)
            out.append(gm.set_data(*(gaussians_parameters[k][b].detach().float()
                                     for k in ("xyz", "features", "scaling", "rotation", "opacity"))))
        return out

    MAX_DIFFERENTIABLE_BATCH = 4      # samples per dgs_dit_forward_train / dgs_dit_backward call (include/dgs_dit.h)

    def image_to_gaussians(self, images, ray_o, ray_d, t, training=False):   # denoiser.py:306-416
        ng = self.cfg.n_gaussians
        differentiable = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())   # in train AND eval mode
        if differentiable:
            B, mb = images.shape[0], self.MAX_DIFFERENTIABLE_BATCH
            if B > mb and self._grads_in_place:
                # a trainer made the .grad tensors views of the flat buffer, which every backward call OVERWRITES: it splits its
                # batch itself (DataParallelTrainer: micro-batches of <= 4 inside one optimizer step)
                raise RuntimeError(f"DGSDenoiser: under a trainer with in-place gradients one call takes at most {mb} samples "
                                   "(DataParallelTrainer.step splits larger per-rank batches into micro-batches itself)")
            names = [n for n, _ in self.named_parameters()]
            plist = [p for _, p in self.named_parameters()]
            parts = []
            for b0 in range(0, B, mb):      # a call takes <= 4 samples: larger batches run as chunks, each with its own activation arena,
                sl = slice(b0, b0 + mb)     # and autograd sums their parameter gradients
                parts.append(_DitFunction.apply(self, names, images[sl], ray_o[sl], ray_d[sl], t[sl], *plist))
            outs = parts[0] if len(parts) == 1 else tuple(torch.cat([p[i] for p in parts], dim=0) for i in range(6))
            xyz, aligned = outs[0], outs[5]
            if self.cfg.clip_xyz and training:        # denoiser.py:397-398 (no shipped caller passes training=True)
                xyz = torch.cat((xyz[:, :ng], xyz[:, ng:].clamp(-1.0, 1.0)), dim=1)
                aligned = aligned.clamp(-1.0, 1.0)
            out = dict(zip(("xyz", "features", "scaling", "rotation", "opacity"), (xyz,) + tuple(outs[1:5])))
            return AttrDict(out), aligned
        out, aligned = self.engine().image_to_gaussians(images, ray_o, ray_d, t)
        if self.cfg.clip_xyz and training:
            out["xyz"][:, ng:].clamp_(-1.0, 1.0)
            aligned.clamp_(-1.0, 1.0)
        return AttrDict(out), aligned

    def render_gaussians(self, gaussian_params, c2w, fxfycxcy, height, width):   # denoiser.py:420-434
        return self.gs_renderer(gaussian_params.xyz, gaussian_params.features, gaussian_params.scaling,
                                gaussian_params.rotation, gaussian_params.opacity, height, width, C2W=c2w, fxfycxcy=fxfycxcy)

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    def run_layers(self, start, end, views=None):   # denoiser.py:441-447 (same signature; `views` only adds a consistency check)
        """-> custom_forward(concat_nerf_img_tokens [b, L, d], t = t_embedder(timesteps) [b, d]) running blocks [start, end).
        The reference hands this closure to torch.utils.checkpoint; here checkpointing lives inside the training path
        (`recompute_policy`), and the closure is an inference-mode utility on the same kernels."""
        def custom_forward(concat_nerf_img_tokens, t):
            with torch.no_grad():
                return self.engine().run_blocks(concat_nerf_img_tokens, t, start, min(end, self.cfg.num_layers), views=views)
        return custom_forward


@register("diffusion-gs-model-scene")
class DGSDenoiserScene(DGSDenoiser):
    SCENE = True

    @dataclass
    class Config(DGSDenoiser.Config):   # denoiser_scene.py:179-204: the same fields; the depth-range triple is declared only here
        range_setting_type: str = "linear_depth"   # never read by the reference: range_func is sigmoid(t) * (far - near) + near (:263)
