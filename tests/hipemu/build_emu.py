"""Builds the CPU-emulation variant of the kernel library (TEST INFRASTRUCTURE ONLY).

Same sources as the product library (open-diffusiongs_amd/csrc/*.hip) compiled as host C++ against
tests/hipemu/hip/hip_runtime.h.  Result: tests/hipemu/build/libdgs_cpuemu.so with the same C ABI.
Never loaded by the product path (dgs_amd/_native.py only ever opens lib/libdgs_hip.so).
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "open-diffusiongs_amd", "csrc")
INCLUDE = os.path.join(REPO, "include")
OUT_DIR = os.path.join(HERE, "build")
LIB = os.path.join(OUT_DIR, "libdgs_cpuemu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "hipemu_runtime.cpp")]
    deps += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE)]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    objs = []
    common = ["-std=c++17", "-O2", "-g", "-fPIC", "-ffp-contract=off", "-mfma", "-I" + HERE, "-I" + INCLUDE, "-I" + CSRC,
              "-Wno-unused-value", "-Wno-unknown-attributes", "-Wno-unknown-pragmas", "-Wno-pass-failed", "-Wno-psabi"]
    from concurrent.futures import ThreadPoolExecutor          # one clang per source, all host cores: ~2.5 min serial -> well under a minute

    def compile_one(s):
        o = os.path.join(OUT_DIR, os.path.basename(s) + ".o")
        flags = list(common)
        if os.path.basename(s) == "dit_gemm_deep.hip":       # ~30 kernel instantiations: 130 s of clang with -g and the vectorizers, 59 s without
            flags = [f for f in flags if f != "-g"] + ["-fno-vectorize", "-fno-slp-vectorize"]
        subprocess.check_call([CLANG, "-x", "c++"] + flags + ["-c", s, "-o", o])
        return o

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as pool:
        objs = list(pool.map(compile_one, srcs + [os.path.join(HERE, "hipemu_runtime.cpp")]))
    subprocess.check_call([CLANG, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
