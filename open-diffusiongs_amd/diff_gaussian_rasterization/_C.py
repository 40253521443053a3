"""Stand-in for the reference's compiled pybind module `diff_gaussian_rasterization._C` (ext.cpp:15-18).

Exposes the same three callables with the same positional argument lists and return tuples
(rasterize_points.h:18-66), implemented by ctypes calls into the gfx950 HIP library through the C ABI of
include/dgs_raster.h.  Importing this module without the built library raises -- there is no fallback.
"""
from dgs_amd.raster import default_backend as _backend


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug):
    """-> (num_rendered:int, color[3,H,W], radii[P] int32, geomBuffer, binningBuffer, imgBuffer)"""
    return _backend().rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                          cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                                          image_width, sh, degree, campos, prefiltered, debug)


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, debug):
    """-> (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)"""
    return _backend().rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                                   cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy,
                                                   dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer,
                                                   imageBuffer, debug)


def mark_visible(means3D, viewmatrix, projmatrix):
    """-> bool[P]"""
    return _backend().mark_visible(means3D, viewmatrix, projmatrix)


_backend()  # fail at import time if the HIP library is missing, like a missing compiled extension would
