"""The REFERENCE's own callers of the hot path, run unchanged on top of the product (SURVEY.md 8b "what calls it"):

  * its sampler -- `create_diffusion("30").p_sample_loop_progressive(model=DGSDenoiser, ...)` (gaussian_diffusion.py:560-603, which
    calls `model(input_batch, t)` through respace.py's `_WrappedModel` at :350,359) -- against `dgs_amd.sampler`;
  * its training forward -- `PointDiffusionSystem.forward` (systems/diffusion_gs_system.py:71-116: TransformInput, q_sample,
    `shape_model.image_to_gaussians`, `shape_model.render_gaussians`, the loss computer) with Lightning and the LPIPS / SSIM networks
    out of the picture -- against the product's own calls on the same noisy inputs, forward and backward.

The reference sources are copied / sliced verbatim into the git-ignored oracle/_ref/py at build time (oracle/build_ref.py, step 3:
`/root/reference` does not exist on the GPU box, the directory travels with the snapshot); nothing of them is on the product path.
Runs on the CPU emulation of the kernels here and on the gfx950 build under `-m gpu`."""
import types

import numpy as np
import pytest
import torch

from dgs_amd import cameras, denoiser as dn, losses, sampler as sm
from oracle import ref_glue

needs_ref = pytest.mark.skipif(not ref_glue.diffusion_available(), reason="oracle/_ref/py not built (oracle/build_ref.py needs /root/reference)")


def _model(device, lib, width, layers, seed=3):
    m = dn.DGSDenoiser(dict(width=width, in_channels=9, patch_size=8, num_layers=layers), device=device, lib=lib)
    m.reset_parameters(seed=seed)
    with torch.no_grad():
        m.image_token_decoder.linear.weight.mul_(5.0)            # renders that depend visibly on the inputs
    return m.to(device)


def _batch(B, V, res, device, seed=0):
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(B, V, 3, res, res, generator=g)
    c2w = torch.tensor(np.stack([cameras.ring_cameras(V, phase_deg=11.0 * b) for b in range(B)]), dtype=torch.float32)
    k = torch.tensor(cameras.default_fxfycxcy(res), dtype=torch.float32).expand(B, V, 4).contiguous()
    return {k_: v.to(device) for k_, v in dict(image=image, c2w=c2w, fxfycxcy=k).items()}


def _sampler_case(device, lib, width, layers, res):
    callers = ref_glue.load_callers()
    refd = ref_glue.load_diffusion()
    m = _model(device, lib, width, layers).eval()
    B, V = 2, 3
    base = _batch(B, V, res, device)
    base["ray_o"], base["ray_d"] = callers.TransformInput(base["image"], base["c2w"], base["fxfycxcy"])   # the reference's own rays
    x_T = torch.randn(B, V - 1, 3, res, res, generator=torch.Generator().manual_seed(5)).to(device)

    def fresh():
        b = {k: v.clone() for k, v in base.items()}
        b["image_noisy"] = x_T.clone()
        return b

    # ---- the reference's loop, driving the product's model ----
    rd = refd.create_diffusion("30")
    torch.manual_seed(11)
    ref_steps = []
    for out in rd.p_sample_loop_progressive(m, shape=(B,), input_batch=fresh(), clip_denoised=True, device=device):
        ref_steps.append((out["sample"].clone(), out["pred_xstart"].clone()))
    assert len(ref_steps) == 30
    ref_final = out
    # ---- the product's sampler, step by step with the same noise stream (th.randn_like(x) behind the model call, :504) ----
    d = sm.create_diffusion("30", device=device, lib=lib)
    torch.manual_seed(11)
    b = fresh()
    x = b["image_noisy"]
    worst = 0.0
    for n, i in enumerate(reversed(range(d.num_timesteps))):
        t = torch.full((B,), i, dtype=torch.int64, device=device)
        b["image"] = torch.cat([b["image"][:, 0:1], b["image_noisy"]], dim=1)
        with torch.no_grad():
            render, _ = m(b, d.model_timesteps(t))
        pred = torch.empty_like(x)
        x = d.step(render.float(), b["image_noisy"].float(), t, clip_denoised=True, pred_xstart=pred)
        b["image_noisy"] = x
        if n < 3:       # the first steps: both loops have seen bit-identical inputs so far -- the sampler arithmetic alone
            assert float((x - ref_steps[n][0]).abs().max()) <= 2e-6 * max(1.0, float(ref_steps[n][0].abs().max())), n
            assert float((pred - ref_steps[n][1]).abs().max()) <= 2e-6, n
        worst = max(worst, float((x - ref_steps[n][0]).abs().max()))
    assert worst <= 1e-4, worst                                   # 30 steps: differences of 1e-7 go through the model each step
    # ---- and the product's own loop (graph replays on the GPU) ----
    torch.manual_seed(11)
    mine = d.p_sample_loop(m, fresh(), clip_denoised=True)
    assert float((mine["sample"] - ref_final["sample"]).abs().max()) <= 1e-4
    assert float((mine["pred_xstart"] - ref_final["pred_xstart"]).abs().max()) <= 1e-4
    r_ref, r_mine = ref_final["denoiser_output_dict"]["render_images"], mine["denoiser_output_dict"]["render_images"]
    assert float((r_ref - r_mine).abs().max()) <= 1e-4
    g_ref, g_mine = ref_final["denoiser_output_dict"]["pred_gaussians"], mine["denoiser_output_dict"]["pred_gaussians"]
    assert len(g_ref) == len(g_mine) == B and float((g_ref[0].get_xyz - g_mine[0].get_xyz).abs().max()) <= 1e-3


def _pipeline_case(device, lib, width, layers, res, half):
    """The call convention of the reference's inference pipeline (pipline_obj.py:264-308), reproduced with its own sampler: `input_batch` an
    EasyDict whose `image` holds ONLY the conditioning view (p_mean_variance rebuilds it as cat(image[:, :1], image_noisy) every step,
    gaussian_diffusion.py:349), rays of all four views from its `TransformInput`, every tensor cast to the system dtype (half precision
    in the shipped pipeline), the loop under `torch.autocast`, `clip_denoised=False`; consumed like the pipeline does
    (`final_out['denoiser_output_dict']['pred_gaussians'][0]`, `['render_images'][0]`)."""
    callers = ref_glue.load_callers()
    refd = ref_glue.load_diffusion()
    from oracle.ref_glue import _EasyDict as edict
    m = _model(device, lib, width, layers, seed=8).eval()
    base = _batch(1, 4, res, device, seed=3)
    image_torch = base["image"][:, 0]                                   # [1, 3, H, W]: the preprocessed input image
    sample_noise = torch.randn(1, 3, 3, res, res, generator=torch.Generator().manual_seed(2)).to(device)
    rgbs_input = torch.cat((image_torch.unsqueeze(1), sample_noise), dim=1)
    ray_o, ray_d = callers.TransformInput(rgbs_input, base["c2w"], base["fxfycxcy"])
    dt = half if half is not None else torch.float32
    input_batch = edict(image=rgbs_input[:, :1].to(dt), c2w=base["c2w"].to(dt), fxfycxcy=base["fxfycxcy"].to(dt), ray_o=ray_o.to(dt), ray_d=ray_d.to(dt))
    input_batch["image_noisy"] = sample_noise.to(dt)
    system = types.SimpleNamespace(diffusion_inference=refd.create_diffusion("30", predict_xstart=True), shape_model=m)
    torch.manual_seed(4)
    steps = 0
    with torch.autocast(device_type=device.type, dtype=dt, enabled=half is not None):
        for out in system.diffusion_inference.p_sample_loop_progressive(system.shape_model, sample_noise.shape, input_batch, clip_denoised=False,
                                                                        progress=False, device=device):
            final_out = out
            steps += 1
    assert steps == 30
    gaussians = final_out["denoiser_output_dict"]["pred_gaussians"][0]
    pred_images = final_out["denoiser_output_dict"]["render_images"][0]
    assert pred_images.shape == (4, 3, res, res) and pred_images.dtype == torch.float32 and bool(torch.isfinite(pred_images).all())
    assert gaussians.get_xyz.shape == (2 + 4 * res * res, 3) and bool(torch.isfinite(gaussians.get_xyz).all())
    assert float(gaussians.get_opacity.min()) >= 0.0 and float(gaussians.get_scaling.max()) <= float(np.exp(-1.2)) + 1e-6   # to_gs clamp, denoiser.py:103-120
    assert final_out["sample"].shape == sample_noise.shape and bool(torch.isfinite(final_out["sample"].float()).all())
    # the render of the conditioning view of the final Gaussians through the product's OWN entry gives the pipeline's image back
    with torch.no_grad():
        again = m.gs_renderer(gaussians._xyz[None], gaussians.get_features[None], gaussians._scaling[None], gaussians._rotation[None],
                              gaussians._opacity[None], res, res, C2W=base["c2w"].to(dt).float(), fxfycxcy=base["fxfycxcy"].to(dt).float())
    assert float((again[0] - pred_images).abs().max()) <= 1e-5


class _LossComputer:
    """LossComputer.forward's signature and return tuple (utils/losses.py:261-369) with the terms the product has on the device; the
    LPIPS / SSIM networks (out of scope, no weights offline) contribute zeros."""

    def __init__(self, lib):
        self.lib = lib

    def __call__(self, rendering, target, masks_all, masks, ray_o, img_aligned_xyz=None, gt_img_aligned_xyz=None):
        _loss, l2, _psnr = losses.mse_psnr(rendering, target, lib=self.lib)
        pd, xyz = losses.points_losses(img_aligned_xyz, ray_o, gt_img_aligned_xyz, masks, lib=self.lib)
        z = torch.zeros_like(l2)
        return l2, z, z, pd, xyz


def _system_case(device, lib, width, layers, res):
    callers = ref_glue.load_callers()
    refd = ref_glue.load_diffusion()
    m = _model(device, lib, width, layers, seed=4).train()
    B, V_in, V_all = 2, 4, 6
    g = torch.Generator().manual_seed(2)
    allv = _batch(B, V_all, res, device, seed=1)
    batch = {"rgbs_input": allv["image"][:, :V_in].clone(), "c2ws_input": allv["c2w"][:, :V_in], "fxfycxcys_input": allv["fxfycxcy"][:, :V_in],
             "depths_input": (1.5 + torch.rand(B, V_in, 1, res, res, generator=g)).to(device),
             "c2ws": allv["c2w"], "fxfycxcys": allv["fxfycxcy"], "rgbs": allv["image"],
             "masks": torch.ones(B, V_all, 1, res, res, device=device), "masks_input": (torch.rand(B, V_in, 1, res, res, generator=g) > 0.3).float().to(device)}
    clean = batch["rgbs_input"].clone()
    system = types.SimpleNamespace(
        cfg=types.SimpleNamespace(noise_scheduler=types.SimpleNamespace(num_train_timesteps=1000)),
        diffusion_training=refd.create_diffusion(str(1000), predict_xstart=True), shape_model=m, loss_computer=_LossComputer(lib))
    torch.manual_seed(21)
    out = callers.forward(system, batch)                          # PointDiffusionSystem.forward, verbatim
    for k in ("loss_diffusion", "loss_lpips", "loss_ssim", "loss_xyz", "loss_pointsdist"):
        assert out[k].dim() == 0 and torch.isfinite(out[k]), k
    assert out["noise_pred"].shape == (B, V_all, 3, res, res) and out["timesteps"].dtype == torch.int64
    x_t = out["x_t"]
    assert torch.equal(x_t[:, 0], clean[:, 0]) and not torch.equal(x_t[:, 1:], clean[:, 1:])       # view 0 stays clean, the others carry q_sample's noise
    loss = out["loss_diffusion"] + 0.5 * out["loss_pointsdist"] + 0.1 * out["loss_xyz"]           # lambda-weighted sum, training_step :118-124
    m.zero_grad()
    loss.backward()
    gref = {n: p.grad.clone() for n, p in m.named_parameters()}
    assert all(torch.isfinite(v).all() for v in gref.values()) and float(gref["transformer.0.attn.qkv.weight"].abs().max()) > 0
    # ---- the same step as the product's own calls on the same noisy inputs: HIP rays, HIP losses ----
    be = m.gs_renderer.backend()
    ray_o, ray_d = be.rays_from_c2w(batch["c2ws_input"], batch["fxfycxcys_input"], res, res)
    m.zero_grad()
    params, aligned = m.image_to_gaussians(x_t.detach(), ray_o, ray_d, out["timesteps"])
    rendered = m.render_gaussians(params, batch["c2ws"], batch["fxfycxcys"], res, res)
    l2 = losses.mse_psnr(rendered, batch["rgbs"], lib=lib)[0]
    pd, xyz = losses.points_losses(aligned, ray_o, ray_o + ray_d * batch["depths_input"], batch["masks_input"], lib=lib)
    # the two paths differ in ONE thing: the rays (the reference's torch TransformInput vs the HIP ray kernel: 1e-6 .. 2e-5 apart),
    # which move the pixel-aligned Gaussians by that much -- compare as images (mean / PSNR), losses and gradient norms
    d_img = (rendered.detach() - out["noise_pred"].detach()).abs()
    report = {"render_mean_abs": float(d_img.mean()), "render_max_abs": float(d_img.max()),
              "l2": (float(l2), float(out["loss_diffusion"])), "pd": (float(pd.mean()), float(out["loss_pointsdist"])),
              "xyz": (float(xyz), float(out["loss_xyz"]))}
    (l2 + 0.5 * pd.mean() + 0.1 * xyz).backward()
    worst = ("", 0.0)
    for n, p in m.named_parameters():
        e = float((p.grad - gref[n]).double().norm() / gref[n].double().norm().clamp_min(1e-30))
        if e > worst[1]:
            worst = (n, e)
    report["worst_grad_rel_l2"] = worst
    rel = lambda a, b: abs(a - b) / max(1e-12, abs(b))
    # (on the GPU the DiT's bf16 operands turn a 1e-6 input difference into ~1e-3 of a Gaussian parameter: measured at width 1024, 4 blocks,
    # 64^2: render mean |diff| 2.4e-4, losses 1.3e-4 relative)
    ok = (report["render_mean_abs"] <= 1e-3 and report["render_max_abs"] <= 5e-2 and rel(*report["l2"]) <= 1e-3 and rel(*report["pd"]) <= 2e-3
          and rel(*report["xyz"]) <= 2e-3 and worst[1] <= 5e-2)
    assert ok, report


@needs_ref
def test_reference_sampler_over_the_product_emulated():
    from emu_util import emu_lib
    _sampler_case(torch.device("cpu"), emu_lib(), 256, 1, 16)


@needs_ref
def test_reference_system_forward_over_the_product_emulated():
    from emu_util import emu_lib
    _system_case(torch.device("cpu"), emu_lib(), 256, 1, 16)


@needs_ref
@pytest.mark.parametrize("half", [None, torch.bfloat16], ids=["fp32", "bf16_autocast"])
def test_reference_pipeline_call_convention_emulated(half):
    from emu_util import emu_lib
    _pipeline_case(torch.device("cpu"), emu_lib(), 256, 1, 16, half)


@needs_ref
@pytest.mark.gpu
def test_reference_pipeline_call_convention_gpu():
    _pipeline_case(torch.device("cuda:0"), None, 1024, 24, 64, torch.float16)


@needs_ref
@pytest.mark.gpu
def test_reference_sampler_over_the_product_gpu():
    """64^2, the shipped architecture (width 1024, 24 blocks, L = 194 at 3 views): 30 steps of the reference's loop and of the product's."""
    _sampler_case(torch.device("cuda:0"), None, 1024, 24, 64)


@needs_ref
@pytest.mark.gpu
def test_reference_system_forward_over_the_product_gpu():
    _system_case(torch.device("cuda:0"), None, 1024, 4, 64)
