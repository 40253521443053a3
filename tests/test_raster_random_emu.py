"""Randomised shapes through the emulated rasterizer against the oracle: image sizes that are not tile multiples, list lengths from
a few entries to several 256-entry batches per tile, runs of equal depth keys of random length (below and above the per-tile sort's
bucket limit), 1-2 views, SH degree 0-2; forward (integer artefacts bit-exact) and, for the smaller scenes, backward in both forms.
The fixed seeds make it a regression test, the spread of shapes is what the hand-picked cases of the other files do not have."""
import numpy as np
import pytest
import torch

from emu_util import emu_backend
from parity_util import assert_forward_parity
from raster_bwd_util import assert_backward_parity
from util_scene import small_scene

CPU = torch.device("cpu")


@pytest.mark.parametrize("form", ["1", "2", "3"], ids=["sort", "scan", "bitonic"])
@pytest.mark.parametrize("seed", [7, 11])
def test_random_scenes(form, seed, monkeypatch):
    monkeypatch.setenv("DGS_RASTER_BIN", form)
    be = emu_backend()
    rng = np.random.default_rng(seed)
    old = be.deterministic
    try:
        for it in range(5):
            H, W = int(rng.choice([16, 24, 48, 64])), int(rng.choice([16, 40, 48, 64]))
            P = int(rng.choice([300, 1500, 5000]))
            V, deg = int(rng.integers(1, 3)), int(rng.integers(0, 3))
            sc, cams = small_scene(P, W, H, seed=int(rng.integers(1, 10 ** 6)), sh_degree=deg, n_views=V, log_scale=float(rng.choice([-1.2, -2.0, -3.5])))
            if rng.random() < 0.6:
                k = int(rng.integers(5, 60))
                a = int(rng.integers(0, P - k))
                sc["xyz"][a:a + k] = sc["xyz"][a]                      # a run of equal depth keys
            assert_forward_parity(be, sc, cams, H, W, CPU, sh_degree=deg)
            if P <= 1500:
                be.deterministic = bool(it & 1)
                assert_backward_parity(be, sc, cams, H, W, CPU, sh_degree=deg, seed=it)
    finally:
        be.deterministic = old
