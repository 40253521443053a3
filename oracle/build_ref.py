"""Recipe: build the REFERENCE rasterizer itself for MI355X as a checker (oracle/_ref/).

TEST INFRASTRUCTURE ONLY.  The product (open-diffusiongs_amd/) is written from scratch and never includes,
links or loads anything produced here; this exists so that the restatement in raster_oracle.cpp and the HIP
product path can be compared with the reference's OWN code running on the same GPU (tests/test_raster_ref_gpu.py,
oracle/make_raster_ref_golden.py).

What it does (only where /root/reference exists, i.e. in the build container -- the GPU box uses the prebuilt
.so files, which travel with the snapshot because oracle/_ref/ is git-ignored but not gpurun-ignored):

  1. hipify-perl each of cuda_rasterizer/{*.h,*.cu} of the reference's diff-gaussian-rasterization submodule into
     oracle/_ref/src/ (git-ignored: no reference source is ever committed), plus four mechanical text fix-ups that
     hipify-perl leaves: `<< <`/`>> >` launch brackets, two CUDA-only includes, `__trap()`, GLM_FORCE_CUDA.
  2. hipcc --offload-arch=gfx950 those three translation units + oracle/ref_shim.hip (our host-pointer C ABI around
     CudaRasterizer::Rasterizer::forward / backward / markVisible) against the reference's vendored glm, twice:
       libdgs_ref_strict.so  -ffp-contract=off  (every a*b+c rounded twice: the arithmetic the oracle restates)
       libdgs_ref_fast.so    compiler defaults  (fused multiply-adds wherever hipcc likes, as nvcc's --fmad=true
                                                 default does in its own places: what a user's build looks like)

rasterize_points.cu / ext.cpp (the torch binding) are not built: ref_shim.hip plays that role without torch.

  3. copy the reference's render glue diffusionGS/models/gsrenderer/{gs_core.py, renderer.py} VERBATIM into oracle/_ref/py/ under
     the reference's own package path (git-ignored like the rest of oracle/_ref/): oracle/ref_glue.py imports it on top of the
     drop-in `diff_gaussian_rasterization` package -- the reference's Camera / render_opencv_cam / DeferredGaussianRender running
     unchanged over the product (tests/test_ref_glue_gpu.py).
"""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/submodules/diff-gaussian-rasterization"
OUT = os.path.join(HERE, "_ref")
SRC = os.path.join(OUT, "src")
FILES = ["auxiliary.h", "config.h", "forward.h", "forward.cu", "backward.h", "backward.cu", "rasterizer.h",
         "rasterizer_impl.h", "rasterizer_impl.cu"]
VARIANTS = {"strict": ["-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt"], "fast": []}


def lib_path(variant):
    return os.path.join(OUT, f"libdgs_ref_{variant}.so")


def available():
    return all(os.path.exists(lib_path(v)) for v in VARIANTS)


def _translate():
    os.makedirs(SRC, exist_ok=True)
    hipify = shutil.which("hipify-perl") or "/opt/rocm/bin/hipify-perl"
    for f in FILES:
        text = subprocess.run([hipify, os.path.join(REF, "cuda_rasterizer", f)], check=True, capture_output=True, text=True).stdout
        text = text.replace("<< <", "<<<").replace(">> >", ">>>")
        text = re.sub(r'#include ""\n', "", text)
        text = re.sub(r"#include <cub/device/device_radix_sort.cuh>\n", "", text)
        text = re.sub(r"#include <cooperative_groups/reduce.h>\n", "", text)
        text = text.replace("__trap()", "__builtin_trap()").replace("#define GLM_FORCE_CUDA", "")
        with open(os.path.join(SRC, f.replace(".cu", ".hip")), "w") as fh:
            fh.write(text)


GLUE_SRC = "/root/reference/diffusionGS/models/gsrenderer"
GLUE_DST = os.path.join(OUT, "py", "diffusionGS", "models", "gsrenderer")


DIFF_SRC = "/root/reference/diffusionGS/models/diffusion"
DIFF_DST = os.path.join(OUT, "py", "diffusionGS", "models", "diffusion")
DIFF_FILES = ("__init__.py", "gaussian_diffusion.py", "respace.py", "diffusion_utils.py")
SYSTEM_SRC = "/root/reference/diffusionGS/systems/diffusion_gs_system.py"
UTILS_SRC = "/root/reference/diffusionGS/systems/utils.py"
CALLERS_DST = os.path.join(OUT, "py", "ref_callers.py")


def _function_source(path, name, cls=None):
    """Verbatim source text of a function (or of a method of `cls`) of a reference file, dedented to module level."""
    import ast
    import textwrap
    text = open(path).read()
    tree = ast.parse(text)
    body = tree.body
    if cls is not None:
        body = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls).body
    fn = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    lines = text.splitlines()[fn.lineno - 1:fn.end_lineno]
    return textwrap.dedent("\n".join(lines)) + "\n", (fn.lineno, fn.end_lineno)


def build_py():
    """Step 3: the reference's Python callers of the hot path, byte for byte, under the git-ignored oracle/_ref/py:
      * diffusionGS/models/gsrenderer/{gs_core.py, renderer.py}   the render glue (oracle/ref_glue.py `load`)
      * diffusionGS/models/diffusion/*.py                          the sampler: create_diffusion, p_sample_loop_progressive
                                                                   (gaussian_diffusion.py:560-603 -> model(input_batch, t) at :350,359)
      * ref_callers.py                                             two functions sliced out by source range: `TransformInput`
                                                                   (systems/utils.py:621-757) and `PointDiffusionSystem.forward`
                                                                   (systems/diffusion_gs_system.py:71-116) -- their files import the
                                                                   whole training stack (Lightning, skimage, cv2, LPIPS), the
                                                                   functions themselves need torch only."""
    if not os.path.isdir(GLUE_SRC):
        return all(os.path.exists(os.path.join(GLUE_DST, f)) for f in ("gs_core.py", "renderer.py"))
    os.makedirs(GLUE_DST, exist_ok=True)
    os.makedirs(DIFF_DST, exist_ok=True)
    for sub in ("gsrenderer", "diffusion"):
        d = os.path.join(OUT, "py")
        for part in ("diffusionGS", "models", sub):
            d = os.path.join(d, part)
            if not (sub == "diffusion" and part == "diffusion"):      # the diffusion package has its own __init__.py
                open(os.path.join(d, "__init__.py"), "a").close()
    for f in ("gs_core.py", "renderer.py"):
        shutil.copyfile(os.path.join(GLUE_SRC, f), os.path.join(GLUE_DST, f))
    for f in DIFF_FILES:
        shutil.copyfile(os.path.join(DIFF_SRC, f), os.path.join(DIFF_DST, f))
    ti, ti_lines = _function_source(UTILS_SRC, "TransformInput")
    fw, fw_lines = _function_source(SYSTEM_SRC, "forward", cls="PointDiffusionSystem")
    with open(CALLERS_DST, "w") as fh:
        fh.write("# GENERATED by oracle/build_ref.py from /root/reference (git-ignored, test infrastructure): verbatim source ranges\n"
                 "from typing import Any, Dict\n\nimport torch\n\n"
                 f"# ---- diffusionGS/systems/utils.py:{ti_lines[0]}-{ti_lines[1]} ----\n{ti}\n\n"
                 f"# ---- diffusionGS/systems/diffusion_gs_system.py:{fw_lines[0]}-{fw_lines[1]} (PointDiffusionSystem.forward) ----\n{fw}")
    return True


def build(force=False, verbose=False):
    """Returns True when both libraries exist afterwards."""
    build_py()
    if not os.path.isdir(REF):
        return available()           # GPU box: prebuilt or nothing
    shim = os.path.join(HERE, "ref_shim.hip")
    deps = [shim, os.path.abspath(__file__)] + [os.path.join(REF, "cuda_rasterizer", f) for f in FILES]
    newest = max(os.path.getmtime(d) for d in deps)
    if not force and available() and all(os.path.getmtime(lib_path(v)) >= newest for v in VARIANTS):
        return True
    _translate()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    glm = os.path.join(REF, "third_party", "glm")
    for variant, flags in VARIANTS.items():
        objs = []
        for s in [os.path.join(SRC, "forward.hip"), os.path.join(SRC, "backward.hip"), os.path.join(SRC, "rasterizer_impl.hip"), shim]:
            o = os.path.join(OUT, f"{variant}_{os.path.basename(s).replace('.hip', '.o')}")
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-I" + glm, "-I" + SRC] + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            objs.append(o)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path(variant)] + objs)
    return True


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv, verbose=True)
    print("oracle/_ref:", "built" if ok else "unavailable (no /root/reference and no prebuilt libraries)")
