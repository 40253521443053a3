"""Loss consumers on the device: per-sample MSE + PSNR and the MSE gradient in one pass, the LPIPS
input resize, and the two terms on the denoiser's pixel-aligned points (points-distribution and xyz loss) with their gradient.

Mirrors diffusionGS/utils/losses.py: the `l2_loss` / `psnr` terms of LossComputer.forward (:281-285, :303), `compute_psnr`
(:399-402), `l2_loss_xyz` (:288-292) and `pointsdist_loss` (:325-364).  LPIPS / SSIM (network-based) are out of scope.
csrc/loss.hip through include/dgs_loss.h; no CPU fallback."""
import ctypes

import torch

from . import _native


def _run(rendering, target, clamp01, want_psnr, grad_scale, lib):
    assert rendering.shape == target.shape and rendering.dtype == torch.float32 and target.dtype == torch.float32
    r, t = rendering.contiguous(), target.contiguous()
    B = r.shape[0]
    n = r[0].numel()
    dev = r.device
    l2 = torch.empty(B, dtype=torch.float32, device=dev)
    psnr = torch.empty(B, dtype=torch.float32, device=dev) if want_psnr else None
    grad = torch.empty_like(r) if grad_scale is not None else None
    partial = torch.empty(B, _native.LOSS_CHUNKS, dtype=torch.float32, device=dev)
    a = _native.DgsMseArgs()
    ptr = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
    a.B, a.n, a.rendering, a.target, a.clamp01 = B, n, ptr(r), ptr(t), int(clamp01)
    a.l2, a.psnr, a.grad, a.grad_scale, a.partial = ptr(l2), ptr(psnr), ptr(grad), float(grad_scale or 0.0), ptr(partial)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if r.is_cuda else None
    rc = (lib or _native.lib()).dgs_mse_psnr(ctypes.byref(a), stream)
    if rc != 0:
        raise RuntimeError(f"dgs_mse_psnr failed: {rc}")
    return l2, psnr, grad


def compute_psnr(ground_truth, predicted, lib=None):
    """losses.py:399-402: clamp to [0, 1], per-image mean squared error over (c, h, w), -10 log10."""
    return _run(predicted, ground_truth, True, True, None, lib)[1]


class _Mse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rendering, target, lib):
        B = rendering.shape[0]
        l2, psnr, grad = _run(rendering, target, False, True, 1.0 / B, lib)     # d(mean_b l2_b) / d rendering
        ctx.save_for_backward(grad)
        ctx.mark_non_differentiable(psnr)
        return l2.mean(), l2, psnr

    @staticmethod
    def backward(ctx, g_loss, g_l2, _g_psnr):
        (grad,) = ctx.saved_tensors
        out = grad * g_loss
        if g_l2 is not None:     # per-sample weights: d l2_b / d rendering = B * grad_b
            out = out + grad * (g_l2 * grad.shape[0]).reshape(-1, *([1] * (grad.dim() - 1)))
        return out, None, None


def mse_psnr(rendering, target, lib=None):
    """rendering / target [b, v, 3, h, w] -> (loss = mean_b l2_b, l2 [b], psnr [b]); loss and l2 are differentiable w.r.t.
    `rendering` (the gradient was produced by the same pass that formed the sums)."""
    return _Mse.apply(rendering, target, lib)


def _resize_args(x_shape, size, lib):
    a = _native.DgsResizeArgs()
    a.planes = 1
    for d in x_shape[:-2]:
        a.planes *= int(d)
    a.in_h, a.in_w, a.out_h, a.out_w = int(x_shape[-2]), int(x_shape[-1]), int(size[0]), int(size[1])
    return a, (lib or _native.lib())


class _LpipsInput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, size, mul, add, lib):
        x = x.contiguous().float()
        out = torch.empty(tuple(x.shape[:-2]) + tuple(size), dtype=torch.float32, device=x.device)
        a, L = _resize_args(x.shape, size, lib)
        a.src, a.dst, a.mul, a.add = ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.data_ptr()), float(mul), float(add)
        stream = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream) if x.is_cuda else None
        rc = L.dgs_resize_bilinear(ctypes.byref(a), stream)
        if rc != 0:
            raise RuntimeError(f"dgs_resize_bilinear failed: {rc}")
        ctx.meta = (tuple(x.shape), tuple(size), float(mul), lib)
        return out

    @staticmethod
    def backward(ctx, g):
        shape, size, mul, lib = ctx.meta
        g = g.contiguous().float()
        dx = torch.empty(shape, dtype=torch.float32, device=g.device)
        a, L = _resize_args(shape, size, lib)
        a.ddst, a.dsrc, a.mul = ctypes.c_void_p(g.data_ptr()), ctypes.c_void_p(dx.data_ptr()), mul
        stream = ctypes.c_void_p(torch.cuda.current_stream(g.device).cuda_stream) if g.is_cuda else None
        rc = L.dgs_resize_bilinear_backward(ctypes.byref(a), stream)
        if rc != 0:
            raise RuntimeError(f"dgs_resize_bilinear_backward failed: {rc}")
        return dx, None, None, None, None


def lpips_input(images, size=(256, 256), lib=None):
    """losses.py:304-309: `F.interpolate(images, size=[256, 256], mode='bilinear') * 2.0 - 1.0` -- what the reference feeds its
    LPIPS module for renderings and targets ([n, 3, h, w] in (0, 1) -> [n, 3, 256, 256] in (-1, 1)), one fused launch,
    differentiable.  The network behind it is out of scope."""
    return _LpipsInput.apply(images, tuple(size), 2.0, -1.0, lib)


def _points_call(aligned, ray_o, gt, masks, w_pd, w_xyz, want_grad, lib):
    B, V, C, H, W = aligned.shape
    assert C == 3 and ray_o.shape == aligned.shape
    dev = aligned.device
    L = lib or _native.lib()
    f = lambda x: None if x is None else x.contiguous().float()
    al, ro, gt, masks = f(aligned), f(ray_o), f(gt), f(masks)
    if gt is not None:
        assert masks is not None and tuple(masks.shape) == (B, V, 1, H, W) and gt.shape == aligned.shape
    pd = torch.empty(B, dtype=torch.float32, device=dev)
    xyz = torch.zeros(1, dtype=torch.float32, device=dev)
    grad = torch.empty_like(al) if want_grad else None
    ws = torch.empty(int(L.dgs_points_loss_workspace_floats(B, V)), dtype=torch.float32, device=dev)
    wpd = f(w_pd)
    a = _native.DgsPointsLossArgs()
    ptr = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else None
    a.B, a.V, a.H, a.W = B, V, H, W
    a.aligned, a.ray_o, a.gt, a.masks, a.pointsdist, a.xyz = ptr(al), ptr(ro), ptr(gt), ptr(masks), ptr(pd), ptr(xyz)
    a.w_pointsdist, a.w_xyz, a.grad, a.workspace = ptr(wpd), float(w_xyz), ptr(grad), ptr(ws)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream) if al.is_cuda else None
    rc = L.dgs_points_loss(ctypes.byref(a), stream)
    if rc != 0:
        raise RuntimeError(f"dgs_points_loss failed: {rc}")
    return pd, xyz[0], grad


class _PointsLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, aligned, ray_o, gt, masks, lib):
        pd, xyz, _ = _points_call(aligned, ray_o, gt, masks, None, 0.0, False, lib)
        ctx.save_for_backward(aligned, ray_o, *([gt, masks] if gt is not None else []))
        ctx.lib = lib
        return pd, xyz

    @staticmethod
    def backward(ctx, g_pd, g_xyz):
        aligned, ray_o, *rest = ctx.saved_tensors
        gt, masks = rest if rest else (None, None)
        w_pd = g_pd if g_pd is not None else torch.zeros(aligned.shape[0], device=aligned.device)
        w_xyz = float(g_xyz) if (g_xyz is not None and gt is not None) else 0.0      # one host read of a scalar; 0 without a gt
        _, _, grad = _points_call(aligned, ray_o, gt, masks, w_pd, w_xyz, True, ctx.lib)
        return grad.to(aligned.dtype), None, None, None, None


def points_losses(img_aligned_xyz, ray_o, gt_img_aligned_xyz=None, masks=None, lib=None):
    """losses.py:288-292,325-364 on the device: -> (pointsdist_loss [b], l2_loss_xyz scalar -- 0 without a gt), both differentiable
    w.r.t. `img_aligned_xyz` [b, v, 3, h, w] (the statistics of the points-distribution target are detached, as in the reference).
    `masks` [b, v, 1, h, w] is the reference's `masks_input`."""
    return _PointsLoss.apply(img_aligned_xyz, ray_o, gt_img_aligned_xyz, masks, lib)
