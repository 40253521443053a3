"""Where the waves of the rasterizer's kernels spend their cycles: one `rocprofv3 --kernel-trace --pmc` pass per regime with the SQ
counters of MI355X_MICROARCH.md's table (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, quad-cycles): parked on
s_waitcnt / a barrier, stalled at issue, issuing -- and the VALU / LDS share of the issuing part.  Forward + backward of 4 views at
256^2 (tools/raster_microbench.py).  Output: gpurun_out/raster_sq_pmc.txt (copy to profiles/)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_traffic import collect_with_durations  # noqa: E402

COUNTERS = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
            "SQ_INSTS_VALU", "SQ_WAVES"]


def main():
    raw = os.path.join(ROOT, "gpurun_out", "raster_sq_pmc_raw")
    lines = []
    for regime in ("trained", "init"):
        cmd = [sys.executable, os.path.join(ROOT, "tools", "raster_microbench.py"), "--res", "256", "--views", "4", "--regime", regime, "--iters", "5"]
        rows = collect_with_durations(COUNTERS, os.path.join(raw, regime), cmd)
        lines.append(f"== {regime}-like regime, 4 views at 256^2; per launch, quad-cycles summed over the launch's waves")
        lines.append(f"{'kernel':58s} {'calls':>5s} {'us':>8s} {'waves':>7s} {'wave cyc':>11s} {'parked':>7s} {'stalled':>8s} {'issuing':>8s} {'valu':>6s} {'lds':>6s} {'VALU insts':>11s}")
        for name, d in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0.0)):
            n = d["calls"]
            wc = d.get("SQ_WAVE_CYCLES", 0.0)
            if "dgs::" not in name or not n or wc <= 0:
                continue
            f = lambda c: d.get(c, 0.0) / wc
            lines.append(f"{name.replace('dgs::', '')[:58]:58s} {n:5d} {d['dur_ns'] / n / 1e3:8.1f} {d.get('SQ_WAVES', 0.0) / n:7.0f} {wc / n:11.0f} "
                         f"{f('SQ_WAIT_ANY'):7.2f} {f('SQ_WAIT_INST_ANY'):8.2f} {f('SQ_ACTIVE_INST_ANY'):8.2f} {f('SQ_ACTIVE_INST_VALU'):6.2f} "
                         f"{f('SQ_ACTIVE_INST_LDS'):6.2f} {d.get('SQ_INSTS_VALU', 0.0) / n:11.0f}")
    out = os.path.join(ROOT, "gpurun_out", "raster_sq_pmc.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
