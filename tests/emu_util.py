"""Loads the CPU-emulated build of the kernel library for `-m "not gpu"` logic tests (never a product path)."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "hipemu"))

_backend = None
_lib = None
# the emulated "chip" has 6 CUs: GEMM launches with fewer than 6 full tiles put their single-live-block side jobs on workgroups of
# their own (dit_gemm_deep.hip, tail_wgs), launches with 6 or more keep them inside the first tile workgroups -- both get tested
os.environ.setdefault("DGS_EMU_CUS", "6")


def emu_lib():
    global _lib
    if _lib is None:
        import build_emu
        from dgs_amd import _native
        _lib = _native.open_library(build_emu.build())
    return _lib


def emu_backend():
    global _backend
    if _backend is None:
        from dgs_amd.raster import RasterBackend
        _backend = RasterBackend(lib=emu_lib())
    return _backend
