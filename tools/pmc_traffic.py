"""HBM traffic per launch from the PMC counters, as MI355X_MICROARCH.md's HBM / rocprofv3 sections prescribe: FETCH_SIZE and
WRITE_SIZE in SEPARATE `rocprofv3 --kernel-trace --pmc` passes (never combined with other tracing), FETCH_SIZE doubled (gfx950
tallies a 128-B request as 64 B), units KiB.  Passes run over bench.py itself (DiT kernels) and over this file's
`--target` mode (the rasterizer, per regime, forward and forward+backward).

    python tools/pmc_traffic.py            -> profiles/pmc_traffic.json (+ gpurun_out/pmc/ raw csv)

The JSON carries the SHA-256 of the kernel sources it was measured on; bench.py reports `traffic` only when that matches
the sources it runs.
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))


def target(regime, what, iters):
    import numpy as np
    import torch
    from dgs_amd import cameras, synth
    from dgs_amd.raster import default_backend, render_views_autograd
    dev, res, V = torch.device("cuda:0"), 256, 4
    be = default_backend()
    tt = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
    sc = synth.gaussian_scene(res, regime=regime, seed=0, activated=False)
    leaves = [tt(sc[k])[None].requires_grad_(what != "forward") for k in ("xyz", "shs", "scales", "rotations", "opacities")]
    c2w = tt(cameras.ring_cameras(V, phase_deg=10))[None]
    k = tt(cameras.default_fxfycxcy(res)).expand(1, V, 4).contiguous()
    w = torch.randn(1, V, 3, res, res, device=dev) / (3 * res * res)
    for _ in range(iters):
        if what == "forward":
            be.render_views(*leaves, res, res, c2w, k)
        else:
            for x in leaves:
                x.grad = None
            render_views_autograd(be, *leaves, res, res, c2w, k).backward(w)
    torch.cuda.synchronize()


def collect(counter, out_dir, cmd):
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out_dir, "--"] + cmd,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    rows = {}
    for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            d = rows.setdefault(r.get("Kernel_Name", "?"), [0.0, 0])
            d[0] += float(r.get("Counter_Value", 0) or 0)
            d[1] += 1
    return rows


# the four block GEMMs of the sampling step at one sample (L = 4098 valid rows, width 1024): kernel-name fragment -> (family,
# algorithmic bytes per launch = A read + W read + output written (+ residual read), 2-byte operands)
_L, _W = 4098, 1024
DIT_FAMILIES = {
    "gemm_sliced_kernel<4, 192, 8, 0, 256>": ("gemm_qkv", _L * _W * 2 + 3 * _W * _W * 2 + _L * 3 * _W * 2),
    "gemm_sliced_kernel<1, 256, 8, 0, 256>": ("gemm_fc1_gelu", _L * _W * 2 + 4 * _W * _W * 2 + _L * 4 * _W * 2),
    # proj and fc2 are one kernel since the K groups (128 x 128 tiles, two launches per block): the per-launch figures are the MEAN of the two
    "gemm_sliced_kernel<2, 128, 8, 0, 128>": ("gemm_proj_fc2_gate_residual_mean",
                                               (_L * 4 * _W * 2 + 4 * _W * _W * 2 + 2 * _L * _W * 4 + _L * _W * 2 + _W * _W * 2 + 2 * _L * _W * 4) // 2),
}


def collect_with_durations(counters, out_dir, cmd):
    """One PMC pass with several SQ counters + the kernel trace of the same run: per kernel {counter: sum, calls, dur_ns: sum}."""
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *counters, "--output-format", "csv", "-d", out_dir, "--"] + cmd,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
    rows = {}
    for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d = rows.setdefault(r.get("Kernel_Name", "?"), {"calls": 0, "dur_ns": 0.0})
            c = r.get("Counter_Name")
            d[c] = d.get(c, 0.0) + float(r.get("Counter_Value", 0) or 0)
            if c == counters[0]:
                d["calls"] += 1
                try:
                    d["dur_ns"] += float(r.get("End_Timestamp", 0)) - float(r.get("Start_Timestamp", 0))
                except (TypeError, ValueError):
                    pass
    return rows


def mfma_busy(raw, bench, out):
    """Matrix-pipe occupancy of the DiT kernels (`north_star`: "rocprof-evidenced MFMA utilisation for the DiT path"):
    SQ_VALU_MFMA_BUSY_CYCLES per dispatch / (1,024 SIMDs x the dispatch's duration x the shader clock).  The counter is summed over
    the chip's SIMDs... the clock is not in the trace, so the table gives busy cycles per SIMD and per microsecond (= the fraction of
    a 1 GHz clock; divide by the clock in GHz that bench.py's `shader_clock_mhz` reports for the fraction of peak issue slots)."""
    rows = collect_with_durations(["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_BUSY_CYCLES"], os.path.join(raw, "dit_mfma"), bench)
    table = {}
    for name, d in rows.items():
        if "dgs::" not in name or not d["calls"]:
            continue
        busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / d["calls"]
        dur_us = d["dur_ns"] / d["calls"] / 1e3 if d["dur_ns"] else None
        table[name[:90]] = {"launches": d["calls"], "mfma_busy_cycles_per_launch": busy, "mfma_insts_per_launch": d.get("SQ_INSTS_MFMA", 0.0) / d["calls"],
                            "avg_us_under_pmc": dur_us,
                            "busy_cycles_per_simd_per_us": (busy / 1024.0 / dur_us) if dur_us else None}
    out["dit_mfma"] = table
    lines = [f"{'kernel':72s} {'launches':>8s} {'MFMA busy cyc/launch':>22s} {'MFMA insts':>12s} {'avg us (PMC run)':>17s} {'busy cyc / SIMD / us':>21s}"]
    for k, v in sorted(table.items(), key=lambda kv: -(kv[1]["mfma_busy_cycles_per_launch"] or 0)):
        lines.append(f"{k[:72]:72s} {v['launches']:8d} {v['mfma_busy_cycles_per_launch']:22.0f} {v['mfma_insts_per_launch']:12.0f} "
                     f"{(v['avg_us_under_pmc'] or 0):17.2f} {(v['busy_cycles_per_simd_per_us'] or 0):21.1f}")
    open(os.path.join(ROOT, "gpurun_out", "dit_pmc.txt"), "w").write("\n".join(lines) + "\n")


def calibrate(raw):
    """tools/ubench/fetch_calib_bench under --pmc FETCH_SIZE: what the counter reports for patterns of known byte counts (streaming,
    one 16-byte piece per 128- / 64-byte line, the blend's record gather).  `reported_bytes` is the RAW counter (KiB x 1024, no
    doubling): 64 B per TCC_EA0_RDREQ.  If a sparse request moved 128 B, `line128` would run at `gbps_if_128B_per_request`."""
    exe = os.path.join(ROOT, "tools", "ubench", "fetch_calib_bench")
    if not os.path.exists(exe):
        src = exe + ".hip"
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", exe, src], check=True, stderr=subprocess.DEVNULL)
    plain = subprocess.run([exe], capture_output=True, text=True).stdout.split("\n")
    rows = {l.split()[0]: l.split()[1:] for l in plain[1:] if len(l.split()) == 5}
    f = collect("FETCH_SIZE", os.path.join(raw, "calib_fetch"), [exe])
    key = {"stream": "stream_kernel", "line128": "line_kernel<128>", "line64": "line_kernel<64>", "records": "records_kernel"}
    out = {}
    for pat, (lanes, useful, ms, _) in rows.items():
        hit = [v for k, v in f.items() if key[pat] in k]
        if not hit:
            continue
        reported = 1024.0 * hit[0][0] / max(hit[0][1], 1)
        lanes, useful, ms = int(lanes), int(useful), float(ms)
        req = reported / 64.0
        out[pat] = {"lanes": lanes, "useful_bytes": useful, "ms": ms, "reported_bytes": int(reported),
                    "reported_over_useful": round(reported / useful, 4), "requests_per_lane": round(req / lanes, 4),
                    "gbps_if_64B_per_request": round(req * 64 / ms / 1e6, 1), "gbps_if_128B_per_request": round(req * 128 / ms / 1e6, 1)}
    return out


def main():
    from bench import FAMILIES, kernel_source_sha
    raw = os.path.join(ROOT, "gpurun_out", "pmc")
    os.makedirs(raw, exist_ok=True)
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    out = {"_source": "tools/pmc_traffic.py: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, separate passes; FETCH_SIZE x 2 "
                      "(MI355X_MICROARCH.md, HBM); KiB -> bytes; averages per dispatch (dit) / per call of 4 views at 256^2 (raster)",
           "kernel_source_sha": kernel_source_sha(), "family_sha": {f: kernel_source_sha(f) for f in FAMILIES},
           "git_head": head or None, "dit": {}, "raster": {}, "kernels": {}}
    py = sys.executable
    bench = [py, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extras", "--graph", "0", "--preheat-s", "0"]
    f = collect("FETCH_SIZE", os.path.join(raw, "dit_fetch"), bench)
    w = collect("WRITE_SIZE", os.path.join(raw, "dit_write"), bench)
    for name in sorted(set(f) | set(w)):
        if "dgs::" not in name:
            continue
        fb = 2.0 * 1024 * f[name][0] / max(f[name][1], 1) if name in f else None
        wb = 1024.0 * w[name][0] / max(w[name][1], 1) if name in w else None
        out["kernels"][name[:90]] = {"fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "launches": f.get(name, w.get(name))[1]}
        if "attention_fwd_kernel" in name and fb is not None and wb is not None:
            out["dit"]["attention"] = {"fetch_bytes_per_launch": int(fb), "write_bytes_per_launch": int(wb),
                                       "traffic_bytes_per_launch": int(fb + wb), "algorithmic_bytes_per_launch": 4098 * 1024 * 2 * 4}
        for frag, (fam, algo) in DIT_FAMILIES.items():
            if frag in name and fb is not None and wb is not None:
                out["dit"][fam] = {"fetch_bytes_per_launch": int(fb), "write_bytes_per_launch": int(wb), "traffic_bytes_per_launch": int(fb + wb),
                                   "algorithmic_bytes_per_launch": int(algo), "traffic_over_algorithmic": round((fb + wb) / algo, 3)}
    try:
        mfma_busy(raw, bench, out)
    except Exception as e:                      # noqa: BLE001 -- an annex
        out["dit_mfma"] = {"error": f"{type(e).__name__}: {e}"}
    iters = 3
    for regime in ("init", "trained"):
        out["raster"][regime] = {}
        for what in ("forward", "forward_backward"):
            cmd = [py, os.path.abspath(__file__), "--target", regime, what, str(iters)]
            f = collect("FETCH_SIZE", os.path.join(raw, f"raster_{regime}_{what}_fetch"), cmd)
            w = collect("WRITE_SIZE", os.path.join(raw, f"raster_{regime}_{what}_write"), cmd)
            fb = sum(2.0 * 1024 * v[0] for k, v in f.items() if "dgs::" in k) / iters
            wb = sum(1024.0 * v[0] for k, v in w.items() if "dgs::" in k) / iters
            out["raster"][regime][what] = int(fb + wb)
            out["raster"][regime][what + "_detail"] = {"fetch_bytes": int(fb), "write_bytes": int(wb), "calls": iters,
                                                       "per_kernel_fetch_bytes": {k[:60]: int(2048 * v[0] / iters) for k, v in f.items() if "dgs::" in k}}
    try:
        out["fetch_calibration"] = calibrate(raw)
    except Exception as e:                      # noqa: BLE001 -- the calibration is an annex, the traffic figures stand without it
        out["fetch_calibration"] = {"error": f"{type(e).__name__}: {e}"}
    path = os.path.join(ROOT, "gpurun_out", "pmc_traffic.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("dit", "dit_mfma", "raster", "fetch_calibration")}, indent=1))
    print("wrote", path, "-> copy to profiles/pmc_traffic.json")


def annotate(path):
    """Adds `family_sha` to a JSON written before the per-family hashes existed: computed from the csrc/ files of the JSON's own
    `git_head` (git show), after checking that the whole-tree hash of that revision IS the JSON's `kernel_source_sha`."""
    from bench import FAMILIES, kernel_source_sha
    d = json.load(open(path))
    rev = d["git_head"]
    ls = subprocess.run(["git", "ls-tree", "--name-only", f"{rev}:open-diffusiongs_amd/csrc"], capture_output=True, text=True, cwd=ROOT, check=True)
    names = ls.stdout.split()
    read = lambda f: subprocess.run(["git", "show", f"{rev}:open-diffusiongs_amd/csrc/{f}"], capture_output=True, cwd=ROOT, check=True).stdout
    assert kernel_source_sha(read=read, names=names) == d["kernel_source_sha"], "the JSON was not measured on the sources of its git_head"
    d["family_sha"] = {f: kernel_source_sha(f, read=read, names=names) for f in FAMILIES}
    d["_family_sha_note"] = (f"family_sha added after the measurement by `tools/pmc_traffic.py --annotate`: hashes of the csrc/ files of git_head {rev} "
                             "per kernel family (bench.FAMILIES); the whole-tree hash of that revision was checked against kernel_source_sha; no measured value was touched")
    json.dump(d, open(path, "w"), indent=1)
    print(json.dumps(d["family_sha"], indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--annotate":
        annotate(sys.argv[2])
    elif len(sys.argv) > 1 and sys.argv[1] == "--target":
        target(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    else:
        main()
