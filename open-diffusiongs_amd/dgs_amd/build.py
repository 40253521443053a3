"""Build recipe for the gfx950 HIP library (in-tree, explicit hipcc; no JIT cache).

`python -m dgs_amd.build` or __graft_entry__.build() produces open-diffusiongs_amd/lib/libdgs_hip.so
from csrc/*.hip.  hipcc cross-compiles for gfx950 without a GPU present.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(HERE)                      # open-diffusiongs_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_DIR = os.path.join(PKG_ROOT, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libdgs_hip.so")
INCLUDE = os.path.join(REPO_ROOT, "include")

COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC,
          "-Wno-unused-value", "-Wno-unused-result"]
# Rasterizer: un-fused IEEE arithmetic is part of the parity contract.
STRICT_FP = ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"]
# DiT attention: fmaxf on MFMA outputs must not be preceded by canonicalising v_max (cdna_hip_programming.md appendix B);
# masked scores use -inf, so infinities stay honoured.
# raster_backward.hip: the SLP vectorizer pairs loads of neighbouring struct fields of the staged entry and leaves it in scratch memory
# (blend_backward_pair_kernel: 3 dwords stored and reloaded per step); the two-pixel arithmetic is written on explicit float pairs
FLAGS = {"raster_forward.hip": STRICT_FP, "raster_backward.hip": STRICT_FP + ["-fno-slp-vectorize"], "sampler.hip": STRICT_FP, "loss.hip": STRICT_FP, "dit_attention.hip": ["-fno-honor-nans", "-fno-slp-vectorize"], "dit_attention_backward.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force=False, verbose=False, instrument=None):
    """instrument (default: env DGS_INSTRUMENT=1): the tools' build with -DDGS_INSTRUMENT -- cycle stamps and measurement variants of
    the GEMM / attention kernels (DGS_GEMM_DBG, DGS_ATTN_DBG, DGS_GEMM_EXP) -- as lib/libdgs_hip_instr.so (objects under lib/instr/);
    load it with DGS_AMD_LIBRARY=<path>.  The product library has none of it compiled in."""
    if instrument is None:
        instrument = os.environ.get("DGS_INSTRUMENT", "0") not in ("", "0")
    if instrument:
        return _build(force, verbose, os.path.join(LIB_DIR, "instr"), os.path.join(LIB_DIR, "libdgs_hip_instr.so"), ["-DDGS_INSTRUMENT"])
    return _build(force, verbose, LIB_DIR, LIB_PATH, [])


def _build(force, verbose, obj_dir, lib_path, extra):
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    objs, todo = [], []
    for src in sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            todo.append([hipcc] + COMMON + extra + FLAGS.get(src, []) + ["-c", s, "-o", o])
        objs.append(o)
    rebuilt = bool(todo)
    if todo:                                              # one hipcc per source, side by side (a header change rebuilds all 15: 4 min serial)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                errs = [ln for ln in r.stderr.splitlines() if "error" in ln]
                sys.stderr.write("\n".join(errs[:20] or r.stderr.splitlines()[-20:]) + "\n")
                raise subprocess.CalledProcessError(r.returncode, cmd)

        with ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as pool:
            list(pool.map(run, todo))
    if rebuilt or not os.path.exists(lib_path):
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
    return lib_path


if __name__ == "__main__":
    # both builds, always: the tools' library has to follow every ABI change of the product's (bench.py counts with it)
    if os.environ.get("DGS_INSTRUMENT", "0") in ("", "0"):
        print(build_hip(force="--force" in sys.argv, verbose=True, instrument=False))
    print(build_hip(force="--force" in sys.argv, verbose=True, instrument=True))
