"""CPU-side checks of bench.py's bookkeeping (the timed path itself needs an MI355X): the algorithmic FLOP formulas of SURVEY.md 8d,
and the rule that a PMC traffic figure is only reported for the kernel sources it was measured on (per kernel family)."""
import json
import os

import bench


def test_algorithmic_flops_match_survey_8d():
    assert abs(bench.dit_flops(258) / 1e12 - 0.163) < 1e-3
    assert abs(bench.dit_flops(4098) / 1e12 - 4.139) < 1e-3
    assert abs(bench.dit_flops(16386) / 1e12 - 36.341) < 2e-3
    assert bench.kernel_flops("attention", 4098, 1) == 4.0 * 4098 * 4098 * 1024
    assert bench.kernel_flops("gemm_qkv", 4098, 2) == 2.0 * 4098 * 3 * 1024 * 1024 * 2


def test_families_partition_the_measured_kernels():
    d = os.path.join(bench.ROOT, "open-diffusiongs_amd", "csrc")
    files = [f for f in os.listdir(d) if f.endswith((".hip", ".h"))]
    dit = {f for f in files if bench.FAMILIES["dit"](f)}
    raster = {f for f in files if bench.FAMILIES["raster"](f)}
    assert "dit_attention.hip" in dit and "dit_common.h" in dit and "raster_forward.hip" not in dit
    assert {"raster_forward.hip", "raster_backward.hip", "raster_common.h", "raster_state.h", "camera.hip"} <= raster
    assert not any(f.startswith("dit_") for f in raster)
    # a family hash moves with a file of the family, and only with one
    read = lambda f: open(os.path.join(d, f), "rb").read()
    poked = lambda f: read(f) + (b"\n// x" if f == "dit_attention.hip" else b"")
    assert bench.kernel_source_sha("dit", read=poked) != bench.kernel_source_sha("dit")
    assert bench.kernel_source_sha("raster", read=poked) == bench.kernel_source_sha("raster")
    assert bench.kernel_source_sha(read=poked) != bench.kernel_source_sha()


def test_traffic_is_reported_only_for_the_sources_it_was_measured_on():
    j = json.load(open(os.path.join(bench.ROOT, "profiles", "pmc_traffic.json")))
    for fam in bench.FAMILIES:
        valid = j["kernel_source_sha"] == bench.kernel_source_sha() or j["family_sha"][fam] == bench.kernel_source_sha(fam)
        assert (bench.pmc_traffic(fam) is not None) == valid
