"""Diffusion sampler on the device: the respaced schedule tables of the reference's `create_diffusion` and the per-step update
of its sampling loop, behind the same names.

Mirrors diffusionGS/models/diffusion:
    create_diffusion(timestep_respacing, noise_schedule, sigma_small, diffusion_steps)     __init__.py:15-51
    SpacedDiffusion (spaced betas, timestep_map, _WrappedModel timestep remap)             respace.py:69-137
    GaussianDiffusion.p_mean_variance / p_sample / p_sample_loop(_progressive)             gaussian_diffusion.py:316-603
for the configuration the repository ships (predict_xstart=True, learn_sigma=False).  The tables are computed once in
float64 numpy exactly like the reference's constructors (host set-up code, not the hot path) and uploaded as float32; every
step is ONE launch of csrc/sampler.hip through the C ABI (include/dgs_sampler.h) with the timestep index read on the device --
the reference rebuilds four numpy->tensor tables per step and deep-copies Gaussian models in prepare_to_save.
There is no CPU fallback: without the HIP library `step` raises.
"""
import ctypes
import math
import os

import numpy as np
import torch

from . import _native


def _named_beta_schedule(name, n):                      # gaussian_diffusion.py:122-165
    if name == "linear":
        scale = 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if name == "squaredcos_cap_v2":
        ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - ab((i + 1) / n) / ab(i / n), 0.999) for i in range(n)])
    raise NotImplementedError(f"unknown beta schedule: {name}")


def space_timesteps(num_timesteps, section_counts):     # respace.py:16-66
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == desired:
                    return set(range(0, num_timesteps, i))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start_idx, all_steps = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        stride = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            all_steps.append(start_idx + round(cur))
            cur += stride
        start_idx += size
    return set(all_steps)


class SpacedDiffusion:
    """The sampling half of the reference's SpacedDiffusion (training_losses stays with the training system)."""

    def __init__(self, use_timesteps, betas, sigma_small=False, device="cuda", lib=None):
        base_acp = np.cumprod(1.0 - np.asarray(betas, dtype=np.float64))
        use = set(use_timesteps)
        last, new_betas, self.timestep_map = 1.0, [], []
        for i, a in enumerate(base_acp):               # respace.py:82-90
            if i in use:
                new_betas.append(1 - a / last)
                last = a
                self.timestep_map.append(i)
        self.original_num_steps = len(base_acp)
        self.betas = b = np.array(new_betas, dtype=np.float64)
        assert b.ndim == 1 and (b > 0).all() and (b <= 1).all()
        self.num_timesteps = int(b.shape[0])
        alphas = 1.0 - b                               # gaussian_diffusion.py:204-246
        acp = np.cumprod(alphas)
        acp_prev = np.append(1.0, acp[:-1])
        self.posterior_variance = b * (1.0 - acp_prev) / (1.0 - acp)
        self.posterior_mean_coef1 = b * np.sqrt(acp_prev) / (1.0 - acp)
        self.posterior_mean_coef2 = (1.0 - acp_prev) * np.sqrt(alphas) / (1.0 - acp)
        if self.num_timesteps > 1:
            large = np.append(self.posterior_variance[1], b[1:])                      # FIXED_LARGE (:374-378)
            small = np.append(self.posterior_variance[1], self.posterior_variance[1:])  # FIXED_SMALL: the clipped posterior (:222-226)
        else:
            large = small = b
        self.model_log_variance = np.log(small if sigma_small else large)
        self.device = torch.device(device)
        self.lib = lib if lib is not None else _native.lib()
        f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).to(torch.float32).to(self.device)
        self._coef1, self._coef2 = f32(self.posterior_mean_coef1), f32(self.posterior_mean_coef2)
        self._sigma = torch.exp(0.5 * f32(self.model_log_variance))
        self._map = torch.tensor(self.timestep_map, dtype=torch.int64, device=self.device)

    # -- _WrappedModel.__call__ (respace.py:131-137): the denoiser sees the ORIGINAL timestep --
    def model_timesteps(self, t):
        return self._map[t]

    def step(self, model_output, x_t, t, noise=None, clip_denoised=True, out=None, pred_xstart=None, predicted_views_offset=1):
        """One p_sample (gaussian_diffusion.py:479-518).  model_output: the denoiser's render [B, V, C, H, W] (views
        `predicted_views_offset`.. are the x0 prediction, :351) or already-sliced [B, V-1, C, H, W] with offset 0;
        x_t / noise [B, V-1, C, H, W]; t int64 [B] loop indices.  Returns x_{t-1} (written into `out`, which may be x_t)."""
        assert model_output.dtype == torch.float32 and x_t.dtype == torch.float32 and t.dtype == torch.int64
        model_output, x_t = model_output.contiguous(), x_t.contiguous()
        B = x_t.shape[0]
        per = x_t[0].numel()
        view = model_output[0, 0].numel()
        if noise is None:
            noise = torch.randn_like(x_t)              # th.randn_like(x), :504
        noise = noise.contiguous()
        if out is None:
            out = torch.empty_like(x_t)
        a = _native.DgsSamplerStepArgs()
        a.B, a.per_sample = B, per
        a.model_stride, a.model_offset = model_output[0].numel(), predicted_views_offset * view
        if a.model_stride - a.model_offset != per:
            raise ValueError("model_output does not hold exactly the predicted views")
        ptr = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
        a.model_output, a.x_t, a.noise, a.t = ptr(model_output), ptr(x_t), ptr(noise), ptr(t)
        a.coef1, a.coef2, a.sigma, a.T = ptr(self._coef1), ptr(self._coef2), ptr(self._sigma), self.num_timesteps
        a.clip_denoised, a.out, a.pred_xstart = int(clip_denoised), ptr(out), ptr(pred_xstart)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x_t.device).cuda_stream) if x_t.is_cuda else None
        rc = self.lib.dgs_sampler_step(ctypes.byref(a), stream)
        if rc != 0:
            raise RuntimeError(f"dgs_sampler_step failed: {rc}")
        return out

    @torch.no_grad()
    def p_sample_loop(self, model, input_batch, clip_denoised=True, keep_last_only=True, use_graph=None):
        """p_sample_loop / p_sample_loop_progressive (gaussian_diffusion.py:520-603) with the model protocol of
        p_mean_variance (:348-352): batch["image"] = cat(image[:, :1], image_noisy); render, gaussians = model(batch, t).
        Returns the last step's dict {"sample", "pred_xstart", "input_batch", "denoiser_output_dict"}.
        use_graph (default: on a GPU, when the model offers `graphed`): the 30 forwards of the loop are replays of one captured
        hipGraph (dgs_amd/graph.py) -- same kernels, same results; the outputs of the last step are cloned out of the graph's tensors."""
        x = input_batch["image_noisy"]
        B = x.shape[0]
        final = None
        if use_graph is None:        # DGS_GRAPH=0: the eager loop everywhere (a debugging switch)
            use_graph = x.is_cuda and hasattr(model, "graphed") and keep_last_only and os.environ.get("DGS_GRAPH", "1") != "0"
        for i in reversed(range(self.num_timesteps)):
            t = torch.full((B,), i, dtype=torch.int64, device=x.device)
            input_batch["image"] = torch.cat([input_batch["image"][:, 0:1], input_batch["image_noisy"]], dim=1)
            if use_graph:
                mt = self.model_timesteps(t)
                render, gaussians = model.graphed(input_batch, mt)(input_batch, mt)
                if i == 0:
                    model.graphed(input_batch, mt).check(wait=True)       # a deferred device-side failure of the replays surfaces here
                    render = render.clone()
                    # NEW containers over copies: the graph's own GaussianModel objects keep pointing at its output tensors
                    import copy
                    gaussians = [copy.copy(gm).set_data(gm._xyz.clone(), gm.get_features.clone(), gm._scaling.clone(), gm._rotation.clone(),
                                                        gm._opacity.clone()) for gm in gaussians]
            else:
                render, gaussians = model(input_batch, self.model_timesteps(t))
            pred = torch.empty_like(x) if (i == 0 or not keep_last_only) else None
            x = self.step(render.float(), input_batch["image_noisy"].float(), t, clip_denoised=clip_denoised, pred_xstart=pred)
            input_batch["image_noisy"] = x
            final = {"sample": x, "pred_xstart": pred, "input_batch": input_batch,
                     "denoiser_output_dict": {"render_images": render, "pred_gaussians": gaussians}}
        return final


def create_diffusion(timestep_respacing, noise_schedule="squaredcos_cap_v2", sigma_small=False, predict_xstart=True, learn_sigma=False,
                     diffusion_steps=1000, device="cuda", lib=None):
    """diffusionGS/models/diffusion/__init__.py:15-51 (sampling side)."""
    if not predict_xstart or learn_sigma:
        raise NotImplementedError("the reference configuration is predict_xstart=True, learn_sigma=False")
    betas = _named_beta_schedule(noise_schedule, diffusion_steps)
    if timestep_respacing is None or timestep_respacing == "":
        timestep_respacing = [diffusion_steps]
    return SpacedDiffusion(space_timesteps(diffusion_steps, timestep_respacing), betas, sigma_small=sigma_small, device=device, lib=lib)
