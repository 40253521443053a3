"""ctypes wrapper around oracle/libdgs_oracle.so (CPU restatement of the reference rasterizer).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product path never imports this module.

Argument order of ``forward`` mirrors CudaRasterizer::Rasterizer::forward
(/root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:198-221).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdgs_oracle.so")


def build(force=False):
    """Compile the oracle with gcc (Makefile in this directory)."""
    src = os.path.join(_HERE, "raster_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdgs_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        L.dgs_oracle_create.restype = ctypes.c_void_p
        L.dgs_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.dgs_oracle_det_expf.restype = ctypes.c_float
        L.dgs_oracle_det_expf.argtypes = [ctypes.c_float]
        L.dgs_oracle_set_threads.restype = ctypes.c_int
        L.dgs_oracle_set_threads.argtypes = [ctypes.c_int]
        L.dgs_oracle_forward.restype = ctypes.c_int
        L.dgs_oracle_forward.argtypes = [
            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
            fp, fp, fp, fp, fp, ctypes.c_float, fp, fp, fp, fp, fp, ctypes.c_float, ctypes.c_float,
            ctypes.c_int]
        L.dgs_oracle_backward.restype = ctypes.c_int
        L.dgs_oracle_backward.argtypes = [ctypes.c_void_p, fp, ctypes.c_int]
        L.dgs_oracle_mark_visible.restype = ctypes.c_int
        L.dgs_oracle_mark_visible.argtypes = [ctypes.c_int, fp, fp, fp, ctypes.POINTER(ctypes.c_uint8)]
        L.dgs_oracle_get.restype = ctypes.c_long
        L.dgs_oracle_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_int)]
        _lib = L
    return _lib


def _f32(a):
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a):
    if a is None or a.size == 0:
        return None
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


_DT = {"out_color": np.float32, "radii": np.int32, "depths": np.float32, "means2D": np.float32,
       "cov3D": np.float32, "conic_opacity": np.float32, "rgb": np.float32, "clamped": np.uint8,
       "tiles_touched": np.uint32, "point_offsets": np.uint32, "keys": np.uint64,
       "point_list": np.uint32, "ranges": np.uint32, "n_contrib": np.uint32, "final_T": np.float32,
       "dL_dmeans2D": np.float32, "dL_dconic": np.float32, "dL_dopacity": np.float32,
       "dL_dcolors": np.float32, "dL_dmeans3D": np.float32, "dL_dcov3D": np.float32,
       "dL_dsh": np.float32, "dL_dscales": np.float32, "dL_drotations": np.float32}


def set_threads(n=0):
    """Threads of the forward's OpenMP loops (0 keeps the default: all host cores).  Returns the count in effect."""
    return int(lib().dgs_oracle_set_threads(int(n)))


def det_expf(x):
    return float(lib().dgs_oracle_det_expf(ctypes.c_float(x)))


class RasterOracle:
    """One CPU rasterizer instance; keeps the Geometry/Binning/Image state of the last forward."""

    def __init__(self):
        self._h = ctypes.c_void_p(lib().dgs_oracle_create())
        self.shape = None

    def __del__(self):
        try:
            lib().dgs_oracle_destroy(self._h)
        except Exception:
            pass

    def forward(self, background, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy,
                image_height, image_width, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, scale_modifier=1.0, sh_degree=0, exp_mode=0):
        means3D = _f32(means3D)
        P = means3D.shape[0]
        shs = _f32(shs)
        M = 0 if shs is None or shs.size == 0 else shs.shape[1]
        args = [_f32(a) for a in (background, means3D, shs, colors_precomp, np.reshape(opacities, (-1,)),
                                   scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos)]
        bg, m3, sh, cp, op, sc, ro, c3, vm, pm, cam = args
        self._keep = args
        n = lib().dgs_oracle_forward(self._h, P, int(sh_degree), M, _ptr(bg), int(image_width),
                                     int(image_height), _ptr(m3), _ptr(sh), _ptr(cp), _ptr(op), _ptr(sc),
                                     float(scale_modifier), _ptr(ro), _ptr(c3), _ptr(vm), _ptr(pm),
                                     _ptr(cam), float(tanfovx), float(tanfovy), int(exp_mode))
        if n < 0:
            raise RuntimeError(f"dgs_oracle_forward failed with code {n}")
        self.shape = (P, M, int(image_height), int(image_width))
        self.num_rendered = n
        return n

    def backward(self, dL_dpix, accum64=False):
        g = _f32(dL_dpix)
        P, M, H, W = self.shape
        assert g.shape == (3, H, W)
        lib().dgs_oracle_backward(self._h, _ptr(g), int(bool(accum64)))

    def get(self, name):
        p = ctypes.c_void_p()
        es = ctypes.c_int()
        n = lib().dgs_oracle_get(self._h, name.encode(), ctypes.byref(p), ctypes.byref(es))
        if n < 0:
            raise KeyError(name)
        dt = np.dtype(_DT[name])
        assert dt.itemsize == es.value
        if n == 0:
            return np.zeros((0,), dtype=dt)
        buf = (ctypes.c_char * (n * es.value)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt).copy()
        P, M, H, W = self.shape
        shapes = {"out_color": (3, H, W), "means2D": (P, 2), "cov3D": (P, 6), "conic_opacity": (P, 4),
                  "rgb": (P, 3), "clamped": (P, 3), "ranges": (-1, 2), "n_contrib": (H, W),
                  "final_T": (H, W), "dL_dmeans2D": (P, 3), "dL_dconic": (P, 2, 2), "dL_dopacity": (P, 1),
                  "dL_dcolors": (P, 3), "dL_dmeans3D": (P, 3), "dL_dcov3D": (P, 6), "dL_dsh": (P, M, 3),
                  "dL_dscales": (P, 3), "dL_drotations": (P, 4)}
        if name in shapes:
            a = a.reshape(shapes[name])
        return a


def mark_visible(means3D, viewmatrix, projmatrix):
    m = _f32(means3D)
    out = np.zeros((m.shape[0],), dtype=np.uint8)
    vm, pm = _f32(viewmatrix), _f32(projmatrix)
    lib().dgs_oracle_mark_visible(m.shape[0], _ptr(m), _ptr(vm), _ptr(pm),
                                  out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out.astype(bool)
