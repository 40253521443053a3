"""Whole-model backward (training forward that saves activations + dgs_dit_backward) on the CPU emulator vs torch autograd
through the fp32 oracle, tiny configuration."""
import pytest
import torch

from dgs_amd.dit import DitEngine
from dit_util import rel_l2, synth_inputs
from emu_util import emu_lib
from oracle import dit_oracle as D

FIELDS = ("xyz", "features", "scaling", "rotation", "opacity")


@pytest.mark.parametrize("scene,pe", [(False, "relative_plk"), (True, "plk")])
def test_parameter_gradients_match_autograd(scene, pe):
    cfg = D.Cfg(width=256, num_layers=2, ray_pe_type=pe, scene=scene, range_far=50.0)
    sd = D.parity_state_dict(cfg, seed=7)
    B, V, res = 2, 2, 16
    images, ray_o, ray_d, t, _, _ = synth_inputs(cfg, B, V, res, seed=4)
    g = torch.Generator().manual_seed(5)
    # oracle with autograd
    leaf = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref, _ = D.image_to_gaussians(leaf, cfg, images, ray_o, ray_d, t)
    wts = {k: torch.randn(ref[k].shape, generator=g) for k in FIELDS}
    sum((ref[k] * wts[k]).sum() for k in FIELDS).backward()
    # HIP kernels (emulated)
    eng = DitEngine(sd, width=cfg.width, num_layers=cfg.num_layers, ray_pe_type=pe, scene=scene, range_near=cfg.range_near,
                    range_far=cfg.range_far, device="cpu", lib=emu_lib())
    out, _ = eng.forward_train(images, ray_o, ray_d, t)
    for k in FIELDS:
        assert rel_l2(out[k], ref[k].detach()) < 2e-2, k
    eng.backward(*(wts[k] for k in FIELDS))
    grads = eng.grad_views()
    assert set(grads.keys()) == set(sd.keys())
    worst = {}
    for k, gv in grads.items():
        r = leaf[k].grad
        assert r is not None, k
        err = rel_l2(gv.reshape(r.shape), r)
        worst[k] = err
        assert err < 4e-2, (k, err)
    assert max(worst.values()) < 4e-2
