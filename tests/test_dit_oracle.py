"""Pins oracle/dit_oracle.py (the denoiser restatement) against golden vectors produced by the REFERENCE'S OWN
Python code (oracle/make_dit_golden.py imports /root/reference/diffusionGS/models/denoiser/denoiser{,_scene}.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import dit_oracle as D

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(tag):
    z = np.load(os.path.join(GOLD, f"dit_golden_{tag}.npz"))
    sd = {k[3:]: torch.tensor(z[k]) for k in z.files if k.startswith("sd_")}
    cfg = D.Cfg(width=int(z["cfg_width"]), in_channels=int(z["cfg_in_channels"]), patch_size=int(z["cfg_patch_size"]),
                n_gaussians=int(z["cfg_n_gaussians"]), dim_heads=int(z["cfg_dim_heads"]), num_layers=int(z["cfg_num_layers"]),
                ray_pe_type=str(z["cfg_ray_pe_type"]), scene=(tag == "scene"),
                range_near=float(z["cfg_range_setting_near"]) if tag == "scene" else 0.0,
                range_far=float(z["cfg_range_setting_far"]) if tag == "scene" else 500.0)
    return z, sd, cfg


@pytest.mark.parametrize("tag", ["obj", "scene"])
def test_restatement_matches_reference_code(tag):
    z, sd, cfg = _load(tag)
    t = lambda k: torch.tensor(z[k])
    # rays: TransformInput restatement vs the reference's function
    ray_o, ray_d = D.transform_input_rays(t("in_c2w"), t("in_fxfycxcy"), 32, 32)
    np.testing.assert_allclose(ray_o.numpy(), z["ray_o"], atol=1e-6)
    np.testing.assert_allclose(ray_d.numpy(), z["ray_d"], atol=1e-6)
    out, aligned = D.image_to_gaussians(sd, cfg, t("in_images"), t("ray_o"), t("ray_d"), t("in_t"))
    for k in ("xyz", "features", "scaling", "rotation", "opacity"):
        ref = z["out_" + k]
        assert out[k].shape == ref.shape
        np.testing.assert_allclose(out[k].numpy(), ref, rtol=2e-5, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    np.testing.assert_allclose(aligned.numpy(), z["out_aligned"], rtol=2e-5, atol=1e-3 if tag == "scene" else 2e-5)


def test_state_dict_keys_match_reference_layout():
    z, sd, cfg = _load("obj")
    mine = D.init_state_dict(cfg)
    assert set(mine.keys()) == set(sd.keys())
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
    zs, sds, cfgs = _load("scene")
    assert tuple(D.init_state_dict(cfgs)["gaussians_pos_embedding"].shape) == tuple(sds["gaussians_pos_embedding"].shape) == (1, 2, 64)


def test_camera_matrices_match_numpy_helper():
    from dgs_amd import cameras
    c2w = torch.tensor(cameras.ring_cameras(3, phase_deg=7.0))
    k = torch.tensor(cameras.default_fxfycxcy(48, 32)).expand(3, 4)
    view, proj, campos, tanfov = D.camera_matrices(c2w, k, 32, 48)
    for i in range(3):
        ref = cameras.camera_from_c2w(c2w[i].numpy(), k[i].numpy(), 32, 48)
        np.testing.assert_allclose(view[i].numpy(), ref["viewmatrix"], atol=1e-6)
        np.testing.assert_allclose(proj[i].numpy(), ref["projmatrix"], atol=1e-5)
        np.testing.assert_allclose(tanfov[i].numpy(), [ref["tanfovx"], ref["tanfovy"]], rtol=1e-6)
