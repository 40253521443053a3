"""Generates tests/golden/c1_smoke.npz: BASELINE.json configs[0] -- the reference's own CPU-runnable case -- as a fixture.

    "Single examp_data image, 1 denoise step, 64^2 render, PyTorch-CPU DiT + reference rasterizer (no GPU)"

Runs only in the build container (needs /root/reference); the fixture it writes is committed, so nothing at test / bench time
touches /root/reference.  What runs here is the inference pipeline of diffusionGS/pipline_obj.py:264-305 cut down to ONE
step of its sampling loop at 64^2, with every Python part being THE REFERENCE'S OWN code:

    input image      examp_data/example/debug_objaverse_dataset/data_examples/image.png (a strip of eight 256^2 views): view 0,
                     box-filtered to 64^2, [0, 1] -- stored in the fixture as uint8 (the reference file itself is not copied)
    cameras          the shipped camera_template.pt comes from the model hub (not available offline): the 4-camera ring of
                     SURVEY.md 8d (radius 3, look-at origin) instead; view 0 = the input view, views 1-3 generated (pipline_obj.py:278-287)
    rays             the reference's TransformInput (systems/utils.py:621-757, extracted by source slice)
    denoiser         the reference's DGSDenoiser.image_to_gaussians (models/denoiser/denoiser.py, imported by file path with the
                     stubs of oracle/make_dit_golden.py) at the SHIPPED architecture: width 1024, 24 blocks, patch 8; weights =
                     dit_oracle.parity_state_dict(seed) loaded with load_state_dict(strict=True) (no checkpoint offline)
    rasterizer       the reference's is CUDA: the C++ restatement oracle/raster_oracle.cpp, which tests/test_raster_ref_gpu.py pins
                     bit for bit to the reference's own kernels (oracle/_ref) on the MI355X; background (1, 1, 1)
    sampler          the reference's diffusion package (models/diffusion), create_diffusion("30").p_sample at the FIRST loop index
                     (29 -> model timestep 999), clip_denoised=False like the pipeline (pipline_obj.py:301)

Stored: the inputs (image, initial noise, the step's noise, cameras, seeds), and the outputs of the step: every 4th Gaussian of
the denoiser's five parameter tensors + float64 sums of all of them, the four 64^2 renders, pred_xstart and x_{t-1}.
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, os.path.join(ROOT, "open-diffusiongs_amd"), HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

REF_IMAGE = "/root/reference/examp_data/example/debug_objaverse_dataset/data_examples/image.png"
OUT = os.path.join(ROOT, "tests", "golden", "c1_smoke.npz")
RES, V, SEED, STRIDE = 64, 4, 21, 4
FIELDS = ("xyz", "features", "scaling", "rotation", "opacity")


def input_view_64():
    """View 0 of the example strip, 256^2 -> 64^2 with a 4 x 4 box filter (exact integer means, rounded to uint8)."""
    from PIL import Image
    strip = np.asarray(Image.open(REF_IMAGE).convert("RGB"), dtype=np.float64)       # [256, 2048, 3]
    v0 = strip[:, :256]
    box = v0.reshape(RES, 256 // RES, RES, 256 // RES, 3).mean(axis=(1, 3))
    return np.clip(np.rint(box), 0, 255).astype(np.uint8).transpose(2, 0, 1)          # [3, 64, 64]


def render_views(params, c2w, fxfycxcy, exp_mode=0):
    """The reference rasterizer's restatement on the activated Gaussians of sample 0 (gs_core.py:330-334,874-945) -> [V, 3, H, W]."""
    from oracle import dit_oracle as D
    from oracle import raster_oracle as RO
    RO.build()
    view, proj, campos, tanfov = D.camera_matrices(c2w, fxfycxcy, RES, RES)
    xyz, shs = params["xyz"][0].numpy(), params["features"][0].numpy()
    op = torch.sigmoid(params["opacity"][0]).numpy()
    sc = torch.exp(params["scaling"][0]).numpy()
    rot = torch.nn.functional.normalize(params["rotation"][0]).numpy()
    out = []
    for v in range(view.shape[0]):
        o = RO.RasterOracle()
        o.forward(np.ones(3, np.float32), xyz, op, view[v].numpy(), proj[v].numpy(), campos[v].numpy(), float(tanfov[v, 0]),
                  float(tanfov[v, 1]), RES, RES, shs=shs, scales=sc, rotations=rot, exp_mode=exp_mode)
        out.append(o.get("out_color"))
    return np.stack(out)


def main():
    import make_dit_golden as G
    import make_sampler_golden as S
    from dgs_amd import cameras
    from oracle import dit_oracle as D
    obj, _ = G.install_stubs()
    TransformInput = G.load_transform_input()
    refdiff = S.load_reference_package()

    cfg = dict(width=1024, in_channels=9, patch_size=8, n_gaussians=2, dim_heads=64, num_layers=24, gaussians_sh_degree=0,
               hard_pixelalign=True, ray_pe_type="relative_plk")
    model = obj.DGSDenoiser(cfg).float().eval()
    model.load_state_dict(D.parity_state_dict(D.Cfg(), SEED), strict=True)

    img_u8 = input_view_64()
    image0 = torch.from_numpy(img_u8.astype(np.float32) / 255.0)[None, None]               # [1, 1, 3, 64, 64]
    g = torch.Generator().manual_seed(SEED + 1)
    noise_T = torch.randn(1, V - 1, 3, RES, RES, generator=g)                              # the sample at timestep T (pipline_obj.py:284)
    c2w = torch.tensor(cameras.ring_cameras(V, phase_deg=10.0))[None]
    fxfycxcy = torch.tensor(cameras.default_fxfycxcy(RES)).expand(1, V, 4).contiguous()
    rgbs = torch.cat((image0, noise_T), dim=1)
    ray_o, ray_d = TransformInput(rgbs, c2w, fxfycxcy)
    batch = dict(image=image0.clone(), c2w=c2w, fxfycxcy=fxfycxcy, ray_o=ray_o, ray_d=ray_d, image_noisy=noise_T.clone())

    kept, timing = {}, {}

    def denoiser(input_batch, t):                       # the model protocol of p_mean_variance (gaussian_diffusion.py:348-352)
        t0 = time.perf_counter()
        with torch.no_grad():
            params, _ = model.image_to_gaussians(input_batch["image"], input_batch["ray_o"], input_batch["ray_d"], t)
        timing["dit_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        render = torch.from_numpy(render_views(params, input_batch["c2w"][0], input_batch["fxfycxcy"][0]))[None]
        timing["raster_s"] = time.perf_counter() - t0
        kept.update(params=params, render=render, model_t=t.clone())
        return render, None

    diffusion = refdiff.create_diffusion(timestep_respacing="30")
    index = diffusion.num_timesteps - 1
    torch.manual_seed(SEED + 2)                         # p_sample draws th.randn_like(x)
    res = diffusion.p_sample(denoiser, batch, torch.tensor([index]), clip_denoised=False)
    torch.manual_seed(SEED + 2)
    step_noise = torch.randn_like(noise_T)

    out = dict(res=np.int64(RES), views=np.int64(V), seed=np.int64(SEED), stride=np.int64(STRIDE), loop_index=np.int64(index),
               model_t=kept["model_t"].numpy(), in_image_u8=img_u8, in_noise_T=noise_T.numpy(), in_step_noise=step_noise.numpy(),
               in_c2w=c2w.numpy(), in_fxfycxcy=fxfycxcy.numpy(), out_render=kept["render"].numpy().astype(np.float32),
               out_pred_xstart=res["pred_xstart"].numpy(), out_sample=res["sample"].numpy(),
               cpu_dit_s=np.float64(timing["dit_s"]), cpu_raster_s=np.float64(timing["raster_s"]), cpu_threads=np.int64(torch.get_num_threads()))
    for k in FIELDS:
        a = kept["params"][k][0].numpy()
        out["out_" + k] = a[::STRIDE].copy()
        out["sum_" + k] = np.float64(a.astype(np.float64).sum())
        out["abs_" + k] = np.float64(np.abs(a.astype(np.float64)).sum())
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes; model_t", out["model_t"], "dit %.2f s raster %.2f s (%d threads)"
          % (timing["dit_s"], timing["raster_s"], torch.get_num_threads()))
    print("render range", float(kept["render"].min()), float(kept["render"].max()), "sample std", float(res["sample"].std()))


if __name__ == "__main__":
    main()
