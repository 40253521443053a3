"""Generates tests/golden/sampler_golden.npz by running THE REFERENCE'S OWN diffusion code.

Runs only in the build container (needs /root/reference); the fixture it writes is committed, so nothing at test / bench
time touches /root/reference.  The reference package diffusionGS/models/diffusion (gaussian_diffusion.py, respace.py,
diffusion_utils.py, __init__.py::create_diffusion) is imported verbatim as a stand-alone package (by path: importing
`diffusionGS` itself pulls pytorch_lightning, which is not installed).  Golden content:
  * schedule tables of create_diffusion(timestep_respacing=R) for R in {"30", "ddim25", "1000"}: spaced betas, timestep_map,
    posterior_mean_coef1/2, the FIXED_LARGE (default) and FIXED_SMALL log-variances;
  * full p_sample() steps (gaussian_diffusion.py:479-518) at several loop indices incl. t == 0, with a stand-in model that
    returns a stored "render" (the sampler only consumes render_imgs[:, 1:]); the noise torch.randn_like drew is stored.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/diffusionGS/models/diffusion"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sampler_golden.npz")


def load_reference_package():
    pkg = types.ModuleType("refdiffusion")
    pkg.__path__ = [REF]
    sys.modules["refdiffusion"] = pkg
    for name in ("diffusion_utils", "gaussian_diffusion", "respace"):
        spec = importlib.util.spec_from_file_location(f"refdiffusion.{name}", os.path.join(REF, f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"refdiffusion.{name}"] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
    spec = importlib.util.spec_from_file_location("refdiffusion", os.path.join(REF, "__init__.py"), submodule_search_locations=[REF])
    init = importlib.util.module_from_spec(spec)
    sys.modules["refdiffusion"] = init
    spec.loader.exec_module(init)
    return init


def main():
    ref = load_reference_package()
    out = {}
    for tag, respacing, kw in (("r30", "30", {}), ("ddim25", "ddim25", {}), ("full", "", {}), ("r30_small", "30", {"sigma_small": True})):
        d = ref.create_diffusion(timestep_respacing=respacing, **kw)
        out[f"{tag}_betas"] = d.betas
        out[f"{tag}_timestep_map"] = np.array(d.timestep_map, dtype=np.int64)
        out[f"{tag}_coef1"] = d.posterior_mean_coef1
        out[f"{tag}_coef2"] = d.posterior_mean_coef2
        gd = sys.modules["refdiffusion.gaussian_diffusion"]
        if d.model_var_type == gd.ModelVarType.FIXED_LARGE:
            out[f"{tag}_log_variance"] = np.log(np.append(d.posterior_variance[1], d.betas[1:]))
        else:
            out[f"{tag}_log_variance"] = d.posterior_log_variance_clipped
    # ---- p_sample steps through the reference code path ----
    d = ref.create_diffusion(timestep_respacing="30")
    g = torch.Generator().manual_seed(1234)
    B, V, C, H, W = 2, 4, 3, 8, 8
    seen_t = []

    class Model:                                   # what DGSDenoiser.forward returns: (render_imgs [B, V, 3, H, W], gaussians)
        def __init__(self, render):
            self.render = render

        def __call__(self, batch, t):
            seen_t.append(t.clone())
            return self.render, None

    for k, i in enumerate((29, 17, 1, 0)):
        render = torch.rand(B, V, C, H, W, generator=g) * 2.4 - 1.2          # exceeds [-1, 1]: exercises the clip
        x_t = torch.randn(B, V - 1, C, H, W, generator=g)
        batch = {"image": torch.rand(B, V, C, H, W, generator=g), "image_noisy": x_t.clone()}
        t = torch.tensor([i] * B)
        torch.manual_seed(100 + k)                                           # p_sample draws th.randn_like(x)
        res = d.p_sample(Model(render), batch, t, clip_denoised=True)
        torch.manual_seed(100 + k)
        noise = torch.randn_like(x_t)
        out[f"step{k}_index"] = np.int64(i)
        out[f"step{k}_model_t"] = seen_t[-1].numpy()                         # the remapped timestep the model was called with
        out[f"step{k}_render"] = render.numpy()
        out[f"step{k}_x_t"] = x_t.numpy()
        out[f"step{k}_noise"] = noise.numpy()
        out[f"step{k}_sample"] = res["sample"].numpy()
        out[f"step{k}_pred_xstart"] = res["pred_xstart"].numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: np.asarray(v).shape for k, v in out.items() if k.startswith("r30_") or k.startswith("step0")})


if __name__ == "__main__":
    main()
