"""Diffusion sampler step: the oracle against golden vectors produced by the reference's own code, the
host tables against the oracle, and the HIP kernel (CPU emulation build here, the real library under -m gpu) against both."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import sampler_oracle as so  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "sampler_golden.npz"))
CASES = {"r30": ("30", False), "ddim25": ("ddim25", False), "full": ("", False), "r30_small": ("30", True)}


@pytest.mark.parametrize("tag", sorted(CASES))
def test_oracle_tables_match_reference(tag):
    resp, small = CASES[tag]
    tab = so.Tables(resp, sigma_small=small)
    assert list(tab.timestep_map) == list(GOLD[f"{tag}_timestep_map"])
    for name, mine in (("betas", tab.betas), ("coef1", tab.coef1), ("coef2", tab.coef2), ("log_variance", tab.log_variance)):
        np.testing.assert_allclose(mine, GOLD[f"{tag}_{name}"], rtol=1e-12, atol=0, err_msg=name)


def test_oracle_step_matches_reference():
    tab = so.Tables("30")
    for k in range(4):
        i = int(GOLD[f"step{k}_index"])
        sample, x0 = so.p_sample(tab, GOLD[f"step{k}_render"], GOLD[f"step{k}_x_t"], i, GOLD[f"step{k}_noise"])
        np.testing.assert_allclose(x0, GOLD[f"step{k}_pred_xstart"], rtol=0, atol=0)
        np.testing.assert_allclose(sample, GOLD[f"step{k}_sample"], rtol=2e-6, atol=2e-6)
        assert list(GOLD[f"step{k}_model_t"]) == [tab.timestep_map[i]] * 2      # _WrappedModel remap


def _check_device(lib, device):
    from dgs_amd import sampler
    for tag, (resp, small) in CASES.items():
        d = sampler.create_diffusion(resp, sigma_small=small, device=device, lib=lib)
        tab = so.Tables(resp, sigma_small=small)
        assert d.timestep_map == tab.timestep_map
        np.testing.assert_allclose(d.posterior_mean_coef1, tab.coef1, rtol=1e-12)
        np.testing.assert_allclose(d.posterior_mean_coef2, tab.coef2, rtol=1e-12)
        np.testing.assert_allclose(d.model_log_variance, tab.log_variance, rtol=1e-12)
    d = sampler.create_diffusion("30", device=device, lib=lib)
    for k in range(4):
        i = int(GOLD[f"step{k}_index"])
        render = torch.from_numpy(GOLD[f"step{k}_render"]).to(device)
        x_t = torch.from_numpy(GOLD[f"step{k}_x_t"]).to(device)
        noise = torch.from_numpy(GOLD[f"step{k}_noise"]).to(device)
        t = torch.full((render.shape[0],), i, dtype=torch.int64, device=device)
        pred = torch.empty_like(x_t)
        out = d.step(render, x_t, t, noise=noise, pred_xstart=pred)
        np.testing.assert_allclose(pred.cpu().numpy(), GOLD[f"step{k}_pred_xstart"], rtol=0, atol=0)
        np.testing.assert_allclose(out.cpu().numpy(), GOLD[f"step{k}_sample"], rtol=2e-6, atol=2e-6)
        assert d.model_timesteps(t).tolist() == list(GOLD[f"step{k}_model_t"])
    # per-sample timesteps, in-place update, t == 0 without noise
    render = torch.from_numpy(GOLD["step0_render"]).to(device)
    x_t = torch.from_numpy(GOLD["step0_x_t"]).to(device)
    noise = torch.from_numpy(GOLD["step0_noise"]).to(device)
    t = torch.tensor([0, 11], dtype=torch.int64, device=device)
    want = [so.p_sample(so.Tables("30"), GOLD["step0_render"][b:b + 1], GOLD["step0_x_t"][b:b + 1], int(t[b]), GOLD["step0_noise"][b:b + 1])[0]
            for b in range(2)]
    buf = x_t.clone()
    d.step(render, buf, t, noise=noise, out=buf)
    np.testing.assert_allclose(buf.cpu().numpy(), np.concatenate(want), rtol=2e-6, atol=2e-6)


def test_kernel_on_emulator():
    from emu_util import emu_lib
    _check_device(emu_lib(), "cpu")


@pytest.mark.gpu
def test_kernel_on_gpu():
    _check_device(None, "cuda:0")


@pytest.mark.gpu
def test_sampling_loop_runs_end_to_end():
    """30-step loop with the real denoiser at 64^2 (random weights): finite, in range after the last clip."""
    from dgs_amd import sampler, synth
    from dgs_amd.denoiser import DGSDenoiser
    torch.manual_seed(0)
    model = DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="relative_plk"), device="cuda:0").eval()   # configs/diffusionGS_rel.yaml
    batch = synth.make_batch(1, 64, device="cuda:0")
    batch["image_noisy"] = torch.randn_like(batch["image"][:, 1:])
    d = sampler.create_diffusion("30", device="cuda:0")
    res = d.p_sample_loop(model, batch)
    assert torch.isfinite(res["sample"]).all() and float(res["pred_xstart"].abs().max()) <= 1.0
