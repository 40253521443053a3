"""Generates tests/golden/dit_golden_{obj,scene}.npz by running THE REFERENCE'S OWN Python denoiser.

Runs only in the build container (needs /root/reference); the fixtures it writes are committed so nothing at test /
bench time touches /root/reference.  The reference modules imported verbatim (by file path) are
    diffusionGS/models/transformers/utils_transformer.py   (DiTBlock, _init_weights)
    diffusionGS/models/denoiser/denoiser.py                (DGSDenoiser and its heads)
    diffusionGS/models/denoiser/denoiser_scene.py          (scene variant)
    diffusionGS/systems/utils.py::TransformInput           (ray generation; extracted by source slice, the module
                                                            itself imports cv2/kiui which are not installed)
Their third-party imports that are not installed here are stubbed:
    timm==0.9.16 Attention / Mlp  -> restated below from the published timm source (vision_transformer.py, layers/mlp.py)
    xformers (import-only), easydict (attribute dict), torchvision.utils.save_image (unused), and the diffusionGS
    package plumbing (register, BaseModule config parsing, Renderer) which is not part of the denoiser math.
"""
import dataclasses
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ---- timm 0.9.16 restatements (timm/models/vision_transformer.py Attention, timm/layers/mlp.py Mlp) ----
class Attention(nn.Module):
    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False, attn_drop=0.0, proj_drop=0.0, norm_layer=nn.LayerNorm):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = nn.Identity()
        self.k_norm = nn.Identity()
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        q, k = self.q_norm(q), self.k_norm(k)
        x = F.scaled_dot_product_attention(q, k, v)
        x = x.transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, norm_layer=None, bias=True, drop=0.0):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class EasyDict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__ = self


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def install_stubs():
    _mod("xformers"); _mod("xformers.ops")
    _mod("timm"); _mod("timm.models")
    _mod("timm.models.vision_transformer", PatchEmbed=object, Attention=Attention, Mlp=Mlp)
    _mod("easydict", EasyDict=EasyDict)
    _mod("torchvision"); _mod("torchvision.utils", save_image=lambda *a, **k: None)

    def register(name):
        return lambda cls: cls

    class BaseModule(nn.Module):
        @dataclasses.dataclass
        class Config:
            weights: object = None

        def __init__(self, cfg=None):
            super().__init__()
            fields = {f.name for f in dataclasses.fields(self.Config)}
            self.cfg = self.Config(**{k: v for k, v in (cfg or {}).items() if k in fields})
            self.configure()

        def configure(self):
            pass

    class Renderer(nn.Module):
        def __init__(self, cfg):
            super().__init__()

    _mod("diffusionGS", register=register)
    _mod("diffusionGS.utils")
    _mod("diffusionGS.utils.checkpoint", checkpoint=None)
    _mod("diffusionGS.utils.base", BaseModule=BaseModule)
    _mod("diffusionGS.utils.typing", __all__=[])
    _mod("diffusionGS.utils.ops", generate_dense_grid_points=None)
    _mod("diffusionGS.models"); _mod("diffusionGS.models.transformers"); _mod("diffusionGS.models.gsrenderer")
    _mod("diffusionGS.models.gsrenderer.renderer", Renderer=Renderer, SceneRenderer=Renderer)
    _load("diffusionGS.models.transformers.utils_transformer", f"{REF}/diffusionGS/models/transformers/utils_transformer.py")
    obj = _load("ref_denoiser", f"{REF}/diffusionGS/models/denoiser/denoiser.py")
    scene = _load("ref_denoiser_scene", f"{REF}/diffusionGS/models/denoiser/denoiser_scene.py")
    return obj, scene


def load_transform_input():
    src = open(f"{REF}/diffusionGS/systems/utils.py").read()
    start = src.index("def TransformInput(")
    end = src.index("\n# ", src.index("return ray_o, ray_d", start))
    ns = {"torch": torch, "F": F}
    exec(compile(src[start:end], "ref_TransformInput", "exec"), ns)
    return ns["TransformInput"]


def ring_c2w(n, radius=3.0, elev=11.0, phase=0.0):
    sys.path.insert(0, os.path.join(os.path.dirname(OUT), "..", "open-diffusiongs_amd"))
    from dgs_amd import cameras
    return cameras.ring_cameras(n, radius, elev, phase)


def main():
    os.makedirs(OUT, exist_ok=True)
    obj, scene = install_stubs()
    TransformInput = load_transform_input()
    for tag, modcls, extra in (("obj", obj.DGSDenoiser, dict(ray_pe_type="relative_plk")),
                               ("scene", scene.DGSDenoiser, dict(ray_pe_type="plk", range_setting_near=0.0, range_setting_far=500.0))):
        torch.manual_seed(0)
        cfg = dict(width=64, in_channels=9, patch_size=8, n_gaussians=2, dim_heads=32, num_layers=2, gaussians_sh_degree=0,
                   hard_pixelalign=True, **extra)
        model = modcls(cfg).float().eval()
        # adaLN gates are zero-ish only through N(0,.02) init (reference initialises biases to 0): make the test
        # sharper by giving every bias a non-zero value
        with torch.no_grad():
            for n_, p_ in model.named_parameters():
                if n_.endswith(".bias"):
                    p_.copy_(torch.randn_like(p_) * 0.05)
        b, v, res = 2, 2, 32
        g = torch.Generator().manual_seed(1)
        images = torch.rand(b, v, 3, res, res, generator=g)
        c2w = torch.tensor(np.stack([ring_c2w(v, phase=20.0 * i) for i in range(b)]))
        f = 1422.222 / 1024 * res
        fxfycxcy = torch.tensor([f, f, res / 2, res / 2], dtype=torch.float32).expand(b, v, 4).contiguous()
        ray_o, ray_d = TransformInput(images, c2w, fxfycxcy)
        t = torch.tensor([17, 801])
        with torch.no_grad():
            params, aligned = model.image_to_gaussians(images, ray_o, ray_d, t)
        out = {"cfg_" + k: np.array(v_) for k, v_ in cfg.items() if not isinstance(v_, str)}
        out["cfg_ray_pe_type"] = np.array(cfg["ray_pe_type"])
        for k, v_ in model.state_dict().items():
            out["sd_" + k] = v_.numpy()
        out.update(in_images=images.numpy(), in_c2w=c2w.numpy(), in_fxfycxcy=fxfycxcy.numpy(), in_t=t.numpy(),
                   ray_o=ray_o.contiguous().numpy(), ray_d=ray_d.contiguous().numpy(), out_aligned=aligned.numpy())
        for k in ("xyz", "features", "scaling", "rotation", "opacity"):
            out["out_" + k] = params[k].numpy()
        path = os.path.join(OUT, f"dit_golden_{tag}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, {k: tuple(params[k].shape) for k in params})


def main_hip(tag="hip256", width=256, layers=2, res=16, b=2, v=2, seed=5, kinds=("obj", "scene"), sh_degree=0):
    """Fixture the gfx950 kernels can run (head_dim 64, width % 256 == 0): weights are NOT stored -- they are
    dit_oracle.parity_state_dict(cfg, seed), loaded into the reference modules with load_state_dict."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import dit_oracle as D
    obj, scene = install_stubs()
    TransformInput = load_transform_input()
    out = {}
    for kind, modcls, extra in (("obj", obj.DGSDenoiser, dict(ray_pe_type="relative_plk")),
                                ("scene", scene.DGSDenoiser, dict(ray_pe_type="plk", range_setting_near=0.0, range_setting_far=50.0))):
        if kind not in kinds:
            continue
        cfg = dict(width=width, in_channels=9, patch_size=8, n_gaussians=2, dim_heads=64, num_layers=layers,
                   gaussians_sh_degree=sh_degree, hard_pixelalign=True, **extra)
        model = modcls(cfg).float().eval()
        ocfg = D.Cfg(width=width, num_layers=layers, ray_pe_type=extra["ray_pe_type"], scene=(kind == "scene"), range_far=50.0,
                     gaussians_sh_degree=sh_degree)
        sd = D.parity_state_dict(ocfg, seed)
        missing, unexpected = model.load_state_dict(sd, strict=True), None
        g = torch.Generator().manual_seed(seed + 1)
        images = torch.rand(b, v, 3, res, res, generator=g)
        c2w = torch.tensor(np.stack([ring_c2w(v, phase=17.0 * i) for i in range(b)]))
        f = 1422.222 / 1024 * res
        fxfycxcy = torch.tensor([f, f, res / 2, res / 2], dtype=torch.float32).expand(b, v, 4).contiguous()
        ray_o, ray_d = TransformInput(images, c2w, fxfycxcy)
        t = torch.tensor([17, 801])[:b]
        with torch.no_grad():
            params, aligned = model.image_to_gaussians(images, ray_o, ray_d, t)
        pre = kind + "_"
        out.update({pre + "in_images": images.numpy(), pre + "in_c2w": c2w.numpy(), pre + "in_fxfycxcy": fxfycxcy.numpy(),
                    pre + "in_t": t.numpy(), pre + "out_aligned": aligned.numpy()})
        for k in ("xyz", "features", "scaling", "rotation", "opacity"):
            out[pre + "out_" + k] = params[k].numpy()
    out.update(width=np.array(width), layers=np.array(layers), res=np.array(res), seed=np.array(seed), sh_degree=np.array(sh_degree))
    path = os.path.join(OUT, f"dit_golden_{tag}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if "--hip" in sys.argv:
        main_hip()
        # 258 tokens (res 64, 4 views): the attention kernel's multi-tile / ring / tail-merge paths, from the reference's code
        main_hip(tag="hip256_l258", res=64, b=1, v=4, seed=9, kinds=("obj",))
        # gaussians_sh_degree 1 (23 Gaussian channels): the heads' and to_gs's general split (denoiser.py:96,109-117)
        main_hip(tag="hip256_sh1", seed=11, kinds=("obj",), sh_degree=1)
    else:
        main()
