"""tests/golden/dit_golden_hip256.npz was produced by the REFERENCE'S OWN denoiser code (oracle/make_dit_golden.py --hip)
at a configuration the gfx950 kernels can run (width 256, head_dim 64).  Here: (1) the fp32 oracle reproduces it,
(2) the HIP kernels -- on the CPU emulator -- reproduce it within bf16 tolerance.  The GPU run is tests/test_dit_gpu.py."""
import pytest

from dit_util import golden_case, rel_l2
from oracle import dit_oracle as D

FIELDS = ("xyz", "features", "scaling", "rotation", "opacity")


@pytest.mark.parametrize("kind", ["obj", "scene"])
def test_oracle_matches_reference_golden(kind):
    cfg, sd, inp, ref = golden_case(kind)
    out, aligned = D.image_to_gaussians(sd, cfg, inp["images"], inp["ray_o"], inp["ray_d"], inp["t"])
    for k in FIELDS:
        assert rel_l2(out[k], ref[k]) < 1e-5, k
    assert rel_l2(aligned, ref["aligned"]) < 1e-5


@pytest.mark.parametrize("kind", ["obj", "scene"])
def test_emulated_hip_matches_reference_golden(kind):
    from dgs_amd.dit import DitEngine
    from emu_util import emu_lib
    cfg, sd, inp, ref = golden_case(kind)
    eng = DitEngine(sd, width=cfg.width, num_layers=cfg.num_layers, ray_pe_type=cfg.ray_pe_type, scene=cfg.scene,
                    range_near=cfg.range_near, range_far=cfg.range_far, device="cpu", lib=emu_lib())
    out, aligned = eng.image_to_gaussians(inp["images"], inp["ray_o"], inp["ray_d"], inp["t"])
    for k in FIELDS:
        assert rel_l2(out[k], ref[k]) < 2e-2, (k, rel_l2(out[k], ref[k]))
    assert rel_l2(aligned, ref["aligned"]) < 2e-2


def test_oracle_matches_reference_golden_258_tokens():
    cfg, sd, inp, ref = golden_case("obj", tag="hip256_l258")
    out, aligned = D.image_to_gaussians(sd, cfg, inp["images"], inp["ray_o"], inp["ray_d"], inp["t"])
    for k in FIELDS:
        assert rel_l2(out[k], ref[k]) < 1e-5, k
    assert rel_l2(aligned, ref["aligned"]) < 1e-5
