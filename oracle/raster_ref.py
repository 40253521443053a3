"""ctypes wrapper around oracle/_ref/libdgs_ref_{strict,fast}.so: the REFERENCE rasterizer's own CUDA sources,
translated at build time by oracle/build_ref.py and run on the MI355X (needs a GPU).

TEST INFRASTRUCTURE ONLY -- same rule as raster_oracle.py; never imported by the product path.
`RasterRef` has the interface of `raster_oracle.RasterOracle` (forward / backward / get with the same names), so
a test can put the restatement, the reference and the HIP product side by side.
"""
import ctypes
import os

import numpy as np

from . import build_ref
from .raster_oracle import _DT, _f32, _ptr

_libs = {}


def available():
    return build_ref.available()


def lib(variant="strict"):
    if variant not in _libs:
        path = build_ref.lib_path(variant)
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `python oracle/build_ref.py` where /root/reference exists")
        L = ctypes.CDLL(path)
        fp = ctypes.POINTER(ctypes.c_float)
        L.dgs_ref_create.restype = ctypes.c_void_p
        L.dgs_ref_destroy.argtypes = [ctypes.c_void_p]
        L.dgs_ref_forward.restype = ctypes.c_int
        L.dgs_ref_forward.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int,
                                      fp, fp, fp, fp, fp, ctypes.c_float, fp, fp, fp, fp, fp, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_int]
        L.dgs_ref_backward.restype = ctypes.c_int
        L.dgs_ref_backward.argtypes = [ctypes.c_void_p, fp, ctypes.c_int]
        L.dgs_ref_mark_visible.restype = ctypes.c_int
        L.dgs_ref_mark_visible.argtypes = [ctypes.c_int, fp, fp, fp, ctypes.POINTER(ctypes.c_uint8)]
        L.dgs_ref_get.restype = ctypes.c_long
        L.dgs_ref_get.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int)]
        L.dgs_ref_time_ms.restype = ctypes.c_double
        L.dgs_ref_time_ms.argtypes = [ctypes.c_void_p, ctypes.c_int]
        _libs[variant] = L
    return _libs[variant]


class RasterRef:
    """One instance of the reference rasterizer; keeps the state of the last forward on the device."""

    def __init__(self, variant="strict"):
        self._L = lib(variant)
        self._h = ctypes.c_void_p(self._L.dgs_ref_create())
        self.shape = None

    def __del__(self):
        try:
            self._L.dgs_ref_destroy(self._h)
        except Exception:
            pass

    def forward(self, background, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, image_height,
                image_width, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                scale_modifier=1.0, sh_degree=0, exp_mode=0, repeat=1):
        means3D = _f32(means3D)
        P = means3D.shape[0]
        shs = _f32(shs)
        M = 0 if shs is None or shs.size == 0 else shs.shape[1]
        args = [_f32(a) for a in (background, means3D, shs, colors_precomp, np.reshape(opacities, (-1,)), scales, rotations,
                                   cov3D_precomp, viewmatrix, projmatrix, campos)]
        bg, m3, sh, cp, op, sc, ro, c3, vm, pm, cam = args
        n = self._L.dgs_ref_forward(self._h, P, int(sh_degree), M, _ptr(bg), int(image_width), int(image_height), _ptr(m3),
                                    _ptr(sh), _ptr(cp), _ptr(op), _ptr(sc), float(scale_modifier), _ptr(ro), _ptr(c3),
                                    _ptr(vm), _ptr(pm), _ptr(cam), float(tanfovx), float(tanfovy), int(repeat))
        if n < 0:
            raise RuntimeError(f"dgs_ref_forward failed with code {n}")
        self.shape = (P, M, int(image_height), int(image_width))
        self.num_rendered = n
        return n

    def backward(self, dL_dpix, accum64=False, repeat=1):
        g = _f32(dL_dpix)
        P, M, H, W = self.shape
        assert g.shape == (3, H, W)
        rc = self._L.dgs_ref_backward(self._h, _ptr(g), int(repeat))
        if rc < 0:
            raise RuntimeError(f"dgs_ref_backward failed with code {rc}")

    def time_ms(self, which):
        return float(self._L.dgs_ref_time_ms(self._h, 0 if which == "forward" else 1))

    def get(self, name):
        p = ctypes.c_void_p()
        es = ctypes.c_int()
        n = self._L.dgs_ref_get(self._h, name.encode(), ctypes.byref(p), ctypes.byref(es))
        if n < 0:
            raise KeyError(name)
        dt = np.dtype(_DT[name])
        assert dt.itemsize == es.value, (name, dt, es.value)
        if n == 0:
            return np.zeros((0,), dtype=dt)
        buf = (ctypes.c_char * (n * es.value)).from_address(p.value)
        a = np.frombuffer(buf, dtype=dt).copy()
        P, M, H, W = self.shape
        shapes = {"out_color": (3, H, W), "means2D": (P, 2), "cov3D": (P, 6), "conic_opacity": (P, 4), "rgb": (P, 3),
                  "clamped": (P, 3), "ranges": (-1, 2), "n_contrib": (H, W), "final_T": (H, W), "dL_dmeans2D": (P, 3),
                  "dL_dconic": (P, 2, 2), "dL_dopacity": (P, 1), "dL_dcolors": (P, 3), "dL_dmeans3D": (P, 3),
                  "dL_dcov3D": (P, 6), "dL_dsh": (P, M, 3), "dL_dscales": (P, 3), "dL_drotations": (P, 4)}
        if name in shapes:
            a = a.reshape(shapes[name])
        return a


def mark_visible(means3D, viewmatrix, projmatrix, variant="strict"):
    m = _f32(means3D)
    out = np.zeros((m.shape[0],), dtype=np.uint8)
    vm, pm = _f32(viewmatrix), _f32(projmatrix)
    lib(variant).dgs_ref_mark_visible(m.shape[0], _ptr(m), _ptr(vm), _ptr(pm), out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)))
    return out.astype(bool)
