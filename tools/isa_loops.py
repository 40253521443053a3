"""Per kernel of one csrc file: the instruction mix of its hot loops (every loop with at least half as many MFMAs as the largest), read off `hipcc -S` with the
flags of dgs_amd/build.py.  What to look at (tools/ubench/issue_bench, profiles/r02_issue_scalar_microbench.txt): beside an MFMA +
VALU stream an s_add costs the wave 2.75 cycles, an s_waitcnt 1.5-2.75, a compare + branch 16 (untaken) / 27 (taken); scratch
traffic inside a loop that also counts vmcnt for its DMA ring drains the ring.  Works without a GPU.
    python tools/isa_loops.py dit_attention.hip [substring of a mangled kernel name ...]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
from dgs_amd import build  # noqa: E402


def disassemble(src):
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + build.COMMON + build.FLAGS.get(src, []) + \
          ["--cuda-device-only", "-S", os.path.join(build.CSRC, src), "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def loops_of(body):
    """Instructions per loop, keyed by the header block, from LLVM's loop-membership comments on the block labels."""
    loops, cur = collections.defaultdict(list), None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):\s*;?(.*)$", l)
        if m or l.startswith("; %bb."):
            com = m.group(2) if m else l
            h = re.search(r"Header=BB(\d+_\d+)", com)
            cur = m.group(1)[2:] if m and "Loop Header" in com else (h.group(1) if h else None)
            continue
        t = l.strip()
        if cur and t and not t.startswith(";") and not t.startswith("."):
            loops[cur].append(t)
    return loops


def main():
    src, want = sys.argv[1], sys.argv[2:]
    L = disassemble(src)
    for start, name in [(i, l.split(":")[0]) for i, l in enumerate(L) if l.startswith("_Z") and ":" in l]:
        if want and not any(w in name for w in want):
            continue
        end = next((i for i in range(start, len(L)) if L[i].startswith(".Lfunc_end")), None)
        if end is None or not any("s_endpgm" in l for l in L[start:end]):
            continue                                    # a device variable, not a kernel
        rows = []
        for h, seg in loops_of(L[start:end]).items():
            c = collections.Counter(x.split()[0].replace("_e32", "").replace("_e64", "") for x in seg)
            mf = sum(v for k, v in c.items() if k.startswith("v_mfma"))
            if mf:
                rows.append((mf, h, c, len(seg)))
        if not rows:
            continue
        print(name)
        top = max(r[0] for r in rows)
        for mf, h, c, n in rows:
            if 2 * mf < top:
                continue                                # prologue / remainder loops
            salu = sum(v for k, v in c.items() if k.startswith("s_") and "branch" not in k and k not in ("s_waitcnt", "s_barrier", "s_nop"))
            valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
            print(f"    loop {h}: {n} instructions | MFMA {mf}  VALU {valu}  SALU {salu}  branches {sum(v for k, v in c.items() if 'branch' in k)}  "
                  f"s_waitcnt {c['s_waitcnt']}  s_nop {c['s_nop']}  s_barrier {c['s_barrier']}  ds_read_b128 {c['ds_read_b128']}  "
                  f"LDS-DMA {c['global_load_lds_dwordx4']}  scratch {sum(v for k, v in c.items() if k.startswith('scratch'))}")


if __name__ == "__main__":
    main()
