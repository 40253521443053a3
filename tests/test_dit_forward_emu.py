"""Whole image_to_gaussians on the CPU emulator vs the fp32 oracle (tiny config: the emulator runs every lane as a fiber)."""
import pytest
import torch

from dgs_amd.dit import DitEngine
from dit_util import bf16_round_state_dict, rel_l2, synth_inputs
from emu_util import emu_lib
from oracle import dit_oracle as D


@pytest.mark.parametrize("scene,pe", [(False, "relative_plk"), (True, "plk")])
def test_forward_matches_oracle(scene, pe):
    cfg = D.Cfg(width=256, num_layers=2, ray_pe_type=pe, scene=scene, range_far=50.0)
    sd = bf16_round_state_dict(D.init_state_dict(cfg, seed=3))
    # non-trivial LayerNorm weights / biases so every parameter is exercised
    g = torch.Generator().manual_seed(9)
    for k in sd:
        if k.endswith("layernorm.weight") or k.endswith("bias"):
            sd[k] = sd[k] + 0.1 * torch.randn(sd[k].shape, generator=g)
    B, V, res = 2, 2, 16
    images, ray_o, ray_d, t, _, _ = synth_inputs(cfg, B, V, res, seed=1)
    ref, ref_aligned = D.image_to_gaussians(sd, cfg, images, ray_o, ray_d, t, return_tokens=True)
    eng = DitEngine(sd, width=cfg.width, num_layers=cfg.num_layers, ray_pe_type=pe, scene=scene, range_near=cfg.range_near,
                    range_far=cfg.range_far, device="cpu", lib=emu_lib())
    out, aligned = eng.image_to_gaussians(images, ray_o, ray_d, t, return_tokens=True)
    assert rel_l2(out["tokens"], ref["tokens"]) < 1e-2
    for k in ("xyz", "features", "scaling", "rotation", "opacity"):
        assert out[k].shape == ref[k].shape, k
        assert rel_l2(out[k], ref[k]) < 2e-2, (k, rel_l2(out[k], ref[k]))
    assert rel_l2(aligned, ref_aligned) < 2e-2
