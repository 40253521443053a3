"""Loads the CPU-emulated build of the kernel library for `-m "not gpu"` logic tests (never a product path)."""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(_HERE, "hipemu"))

_backend = None
_lib = None


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
