"""CPU restatement (numpy, float64 tables / float32 step) of the reference's diffusion sampler step -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(dgs_amd/sampler.py + csrc/sampler.hip) never does.  Pinned by tests/golden/sampler_golden.npz, which
oracle/make_sampler_golden.py produced by running the reference's own diffusionGS/models/diffusion package.

Follows (reference file:line):
  get_named_beta_schedule / betas_for_alpha_bar      gaussian_diffusion.py:122-165
  space_timesteps                                    respace.py:16-66
  SpacedDiffusion.__init__ (spaced betas, timestep_map)   respace.py:77-91
  GaussianDiffusion.__init__ (posterior tables)      gaussian_diffusion.py:183-246
  p_mean_variance: START_X + clip, FIXED_LARGE / FIXED_SMALL variance, q_posterior mean   :316-412, :291-313
  p_sample: x_{t-1} = mean + [t != 0] exp(0.5 log_var) noise                               :479-518
  _WrappedModel: the model sees timestep_map[t]      respace.py:121-137
"""
import math

import numpy as np


def named_beta_schedule(name, n):
    if name == "linear":
        scale = 1000 / n
        return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)
    if name == "squaredcos_cap_v2":
        ab = lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1 - ab((i + 1) / n) / ab(i / n), 0.999) for i in range(n)])
    raise NotImplementedError(name)


def space_timesteps(num_timesteps, section_counts):
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            want = int(section_counts[4:])
            for i in range(1, num_timesteps):
                if len(range(0, num_timesteps, i)) == want:
                    return set(range(0, num_timesteps, i))
            raise ValueError("no integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per, extra = divmod(num_timesteps, len(section_counts))
    start, steps = 0, []
    for i, cnt in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < cnt:
            raise ValueError("section too small")
        stride = 1 if cnt <= 1 else (size - 1) / (cnt - 1)
        cur = 0.0
        for _ in range(cnt):
            steps.append(start + round(cur))
            cur += stride
        start += size
    return set(steps)


class Tables:
    """create_diffusion(timestep_respacing, noise_schedule, sigma_small, diffusion_steps) with predict_xstart=True, learn_sigma=False."""

    def __init__(self, timestep_respacing="", noise_schedule="squaredcos_cap_v2", sigma_small=False, diffusion_steps=1000):
        base = named_beta_schedule(noise_schedule, diffusion_steps)
        if timestep_respacing is None or timestep_respacing == "":
            timestep_respacing = [diffusion_steps]
        use = space_timesteps(diffusion_steps, timestep_respacing)
        acp_base = np.cumprod(1.0 - base)
        last, betas, self.timestep_map = 1.0, [], []
        for i, a in enumerate(acp_base):
            if i in use:
                betas.append(1 - a / last)
                last = a
                self.timestep_map.append(i)
        self.betas = betas = np.array(betas, dtype=np.float64)
        alphas = 1.0 - betas
        acp = np.cumprod(alphas)
        acp_prev = np.append(1.0, acp[:-1])
        post_var = betas * (1.0 - acp_prev) / (1.0 - acp)
        self.coef1 = betas * np.sqrt(acp_prev) / (1.0 - acp)
        self.coef2 = (1.0 - acp_prev) * np.sqrt(alphas) / (1.0 - acp)
        if sigma_small:
            self.log_variance = np.log(np.append(post_var[1], post_var[1:]))
        else:
            self.log_variance = np.log(np.append(post_var[1], betas[1:]))
        self.num_timesteps = len(betas)


def p_sample(tab, render, x_t, index, noise, clip_denoised=True):
    """render [B, V, C, H, W] (model output: views 1.. are the prediction), x_t / noise [B, V-1, C, H, W], index: loop index
    (same for the whole batch).  float32 arithmetic in the reference's order.  -> (sample, pred_xstart)."""
    mo = render[:, 1:].astype(np.float32)
    x0 = np.clip(mo, -1.0, 1.0) if clip_denoised else mo
    c1, c2 = np.float32(tab.coef1[index]), np.float32(tab.coef2[index])
    mean = c1 * x0 + c2 * x_t.astype(np.float32)
    sigma = np.exp(np.float32(0.5) * np.float32(tab.log_variance[index]))
    nz = np.float32(0.0 if index == 0 else 1.0)
    return (mean + nz * sigma * noise.astype(np.float32)).astype(np.float32), x0
