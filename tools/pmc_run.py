"""Collect rocprofv3 PMC counters for a command in separate passes (one counter group per run, kernel-trace only --
never combined with sys/hip/hsa tracing) and print per-kernel sums.  Usage:
    python tools/pmc_run.py OUT_PREFIX -- python tools/attn_bench.py ...
Writes OUT_PREFIX_pmc.txt (table) next to the raw rocprofv3 output directories under gpurun_out/."""
import csv
import glob
import os
import subprocess
import sys

GROUPS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES",
     "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_LDS", "SQ_INSTS_MFMA",
     "SQ_WAVES", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]


def available():
    try:
        txt = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return None
    return txt


def main():
    prefix = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    avail = available()
    rows = {}
    for gi, group in enumerate(GROUPS):
        g = [c for c in group if avail is None or c in avail]
        if not g:
            continue
        out = f"{prefix}_pmc{gi}"
        subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", *g, "--output-format", "csv", "-d", out, "--"] + cmd,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r.get("Kernel_Name", "?")[:60]
                d = rows.setdefault(k, {"calls": {}})
                c = r.get("Counter_Name")
                d[c] = d.get(c, 0.0) + float(r.get("Counter_Value", 0) or 0)
                d["calls"][c] = d["calls"].get(c, 0) + 1
    lines = []
    for k, d in sorted(rows.items()):
        lines.append(k)
        for c, v in sorted(d.items()):
            if c == "calls":
                continue
            n = d["calls"][c]
            lines.append(f"    {c:28s} total {v:16.0f}   per-dispatch {v / max(n, 1):16.1f}   dispatches {n}")
    text = "\n".join(lines)
    print(text)
    open(prefix + "_pmc.txt", "w").write(text + "\n")


if __name__ == "__main__":
    main()
