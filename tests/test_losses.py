"""MSE / PSNR loss consumer against the torch expressions the reference uses (losses.py:281-285, 303, 399-402)."""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _check(lib, device):
    from dgs_amd import losses
    g = torch.Generator().manual_seed(5)
    b, v, h, w = 3, 4, 16, 24
    render = (torch.rand(b, v, 3, h, w, generator=g) * 1.4 - 0.2).to(device).requires_grad_(True)
    target = torch.rand(b, v, 3, h, w, generator=g).to(device)
    loss, l2, psnr = losses.mse_psnr(render, target, lib=lib)
    ref_el = F.mse_loss(render.detach().cpu().double(), target.cpu().double(), reduction="none").reshape(b, v, -1, h, w)
    ref_l2 = ref_el.mean(dim=(1, 2, 3, 4))
    assert torch.allclose(l2.detach().cpu().double(), ref_l2, rtol=2e-6)
    assert torch.allclose(psnr.cpu().double(), -10.0 * torch.log10(ref_l2), rtol=1e-5, atol=1e-5)
    assert torch.allclose(loss.detach().cpu().double(), ref_l2.mean(), rtol=2e-6)
    (0.7 * loss + (l2 * torch.tensor([1.0, 0.0, 2.0], device=device)).sum()).backward()
    rr = render.detach().cpu().double().requires_grad_(True)
    rl2 = ((rr - target.cpu().double()) ** 2).mean(dim=(1, 2, 3, 4))
    (0.7 * rl2.mean() + (rl2 * torch.tensor([1.0, 0.0, 2.0], dtype=torch.float64)).sum()).backward()
    assert torch.allclose(render.grad.cpu().double(), rr.grad, rtol=1e-5, atol=1e-9)
    # compute_psnr clamps to [0, 1] and reduces per image
    img_r, img_t = render.detach().reshape(b * v, 3, h, w), target.reshape(b * v, 3, h, w)
    want = -10 * torch.log10(((img_t.cpu().clamp(0, 1) - img_r.cpu().clamp(0, 1)) ** 2).mean(dim=(1, 2, 3)))
    assert torch.allclose(losses.compute_psnr(img_t, img_r, lib=lib).cpu(), want, rtol=1e-5, atol=1e-5)
    # deterministic: same inputs, same bits
    assert torch.equal(losses.mse_psnr(render.detach(), target, lib=lib)[1], l2.detach())


def _check_points(lib, device, b=2, v=3, h=20, w=28):
    """losses.py:288-292,325-364 (l2_loss_xyz, pointsdist_loss) against the reference's torch expressions in fp64: values and the
    gradient w.r.t. img_aligned_xyz under per-sample / scalar upstream weights; with and without a gt; deterministic."""
    from dgs_amd import losses
    g = torch.Generator().manual_seed(11)
    ray_o = (torch.randn(b, v, 3, 1, 1, generator=g) * 0.3 + torch.tensor([0.0, 0.0, 2.0]).reshape(1, 1, 3, 1, 1)).expand(b, v, 3, h, w).contiguous()
    ray_d = torch.nn.functional.normalize(torch.randn(b, v, 3, h, w, generator=g), dim=2)
    aligned = (ray_o + ray_d * (1.5 + 0.5 * torch.rand(b, v, 1, h, w, generator=g))).to(device).requires_grad_(True)
    gt = ray_o + ray_d * (1.4 + 0.6 * torch.rand(b, v, 1, h, w, generator=g))
    masks = (torch.rand(b, v, 1, h, w, generator=g) > 0.3).float()
    wts = torch.tensor([0.7, 1.9][:b])

    def ref(al, with_gt):
        ro, al = ray_o.double(), al.double()
        dist = (al - ro).norm(dim=2, p=2, keepdim=True)
        dd = dist.detach()
        trgt = (dd - dd.mean(dim=(2, 3, 4), keepdim=True)) / (dd.std(dim=(2, 3, 4), keepdim=True) + 1e-8) * 0.5 + ro.norm(dim=2, p=2, keepdim=True)
        pd = ((dist - trgt) ** 2).mean(dim=(1, 2, 3, 4))
        xyz = F.mse_loss(al * masks.double(), gt.double() * masks.double(), reduction="sum") / masks.double().sum() if with_gt else torch.zeros((), dtype=torch.float64)
        return pd, xyz

    for with_gt in (True, False):
        aligned.grad = None
        pd, xyz = losses.points_losses(aligned, ray_o.to(device), gt.to(device) if with_gt else None, masks.to(device) if with_gt else None, lib=lib)
        ((pd * wts.to(device)).sum() + 0.6 * xyz).backward()
        ar = aligned.detach().cpu().double().requires_grad_(True)
        rpd, rxyz = ref(ar, with_gt)
        ((rpd * wts.double()).sum() + 0.6 * rxyz).backward()
        assert torch.allclose(pd.detach().cpu().double(), rpd.detach(), rtol=2e-4), (pd, rpd)
        assert torch.allclose(xyz.detach().cpu().double(), rxyz.detach(), rtol=1e-5, atol=1e-12), (xyz, rxyz)
        gerr = float((aligned.grad.cpu().double() - ar.grad).abs().max() / ar.grad.abs().max())
        assert gerr < 2e-4, (with_gt, gerr)
        again = losses.points_losses(aligned.detach(), ray_o.to(device), gt.to(device) if with_gt else None, masks.to(device) if with_gt else None, lib=lib)
        assert torch.equal(again[0], pd.detach()) and torch.equal(again[1], xyz.detach())        # fixed reduction order


def test_points_losses_on_emulator():
    from emu_util import emu_lib
    _check_points(emu_lib(), "cpu")


@pytest.mark.gpu
def test_points_losses_on_gpu():
    _check_points(None, "cuda:0")
    _check_points(None, "cuda:0", b=2, v=4, h=256, w=256)


def _check_resize(lib, device, planes=(3, 3), sizes=((256, 256), (512, 512), (64, 96), (300, 200))):
    """The LPIPS input path (losses.py:304-309) vs F.interpolate: the reference's two cases (256 -> 256, 512 -> 256) and odd
    up / down factors; values and the gradient w.r.t. the rendering."""
    from dgs_amd import losses
    g = torch.Generator().manual_seed(9)
    for (h, w) in sizes:
        x = torch.rand(*planes, h, w, generator=g).to(device).requires_grad_(True)
        y = losses.lpips_input(x, lib=lib)
        xr = x.detach().cpu().clone().requires_grad_(True)
        ref = F.interpolate(xr, size=[256, 256], mode="bilinear") * 2.0 - 1.0
        assert y.shape == (*planes, 256, 256)
        assert torch.allclose(y.detach().cpu(), ref.detach(), rtol=0, atol=2e-6), (h, w, float((y.detach().cpu() - ref.detach()).abs().max()))
        wgt = torch.randn(ref.shape, generator=g)
        (y * wgt.to(device)).sum().backward()
        (ref * wgt).sum().backward()
        assert torch.allclose(x.grad.cpu(), xr.grad, rtol=1e-4, atol=1e-5 * float(xr.grad.abs().max())), (h, w)     # sums of up to ~16 fp32 terms, order differs


def test_resize_on_emulator():
    from emu_util import emu_lib
    _check_resize(emu_lib(), "cpu", planes=(1, 2), sizes=((256, 256), (512, 512), (64, 96)))


@pytest.mark.gpu
def test_resize_on_gpu():
    _check_resize(None, "cuda:0")


def test_on_emulator():
    from emu_util import emu_lib
    _check(emu_lib(), "cpu")


@pytest.mark.gpu
def test_on_gpu():
    _check(None, "cuda:0")
