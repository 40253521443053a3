"""MI355X drop-in for the `diff_gaussian_rasterization` Python package.

Public surface mirrored from /root/reference/submodules/diff-gaussian-rasterization/
diff_gaussian_rasterization/__init__.py (names, argument order, return values, gradient order):

  GaussianRasterizationSettings  (:157-169)   NamedTuple, same field order
  GaussianRasterizer             (:171-220)   nn.Module: forward(...) -> (color, radii), markVisible(positions)
  rasterize_gaussians            (:20-42)     functional entry
  _RasterizeGaussians            (:44-155)    autograd.Function; backward returns grads for
                                              (means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                               cov3Ds_precomp, None)
  cpu_deep_copy_tuple            (:17-19)

so gs_core.py:10-13 imports it unchanged.  The compute happens in hand-written gfx950 kernels behind `_C`.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _C

_FORWARD_DUMP = "snapshot_fw.dump"
_BACKWARD_DUMP = "snapshot_bw.dump"


def cpu_deep_copy_tuple(input_tuple):
    return tuple(x.cpu().clone() if isinstance(x, torch.Tensor) else x for x in input_tuple)


def _call_with_snapshot(fn, args, debug, dump_path, what):
    """debug=True keeps a CPU copy of the arguments and dumps it if the native call raises (:83-91,132-141)."""
    if not debug:
        return fn(*args)
    saved = cpu_deep_copy_tuple(args)
    try:
        return fn(*args)
    except Exception:
        torch.save(saved, dump_path)
        print(f"\nAn error occured in {what}. Please forward {dump_path} for debugging.")
        raise


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        rs = raster_settings
        native_args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                       rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                       rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        num_rendered, color, radii, geom, binning, img = _call_with_snapshot(
            _C.rasterize_gaussians, native_args, rs.debug, _FORWARD_DUMP, "forward")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        rs = ctx.raster_settings
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = ctx.saved_tensors
        native_args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                       rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos,
                       geom, ctx.num_rendered, binning, img, rs.debug)
        (g_means2D, g_colors, g_opacities, g_means3D, g_cov3D, g_sh, g_scales, g_rotations) = _call_with_snapshot(
            _C.rasterize_gaussians_backward, native_args, rs.debug, _BACKWARD_DUMP, "backward")
        return (g_means3D, g_means2D, g_sh, g_colors, g_opacities, g_scales, g_rotations, g_cov3D, None)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


def _absent():
    # The reference marks "not provided" with an empty CPU tensor (:197-207); the native side maps numel()==0 to NULL.
    return torch.Tensor([])


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        have_sr = scales is not None or rotations is not None
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (have_sr and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        shs = _absent() if shs is None else shs
        colors_precomp = _absent() if colors_precomp is None else colors_precomp
        scales = _absent() if scales is None else scales
        rotations = _absent() if rotations is None else rotations
        cov3D_precomp = _absent() if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)
