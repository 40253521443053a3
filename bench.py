#!/usr/bin/env python
"""Contract benchmark: novel-view renders/sec of one DiffusionGS sampling step (DiT step + GS raster) at 256^2.

A "step" is one `DGSDenoiser.forward(input_batch, t)` of the object model (width 1024, 24 blocks, patch 8 -- the shipped
`diffusion-gs-model`): `image_to_gaussians` on B samples x 4 input views (L = 4098 tokens, 4.139 TFLOP / sample, bf16
MFMA) followed by the rasterization of every sample's P = 262,146 per-pixel Gaussians into its 4 views (fp32) -- what
the reference's sampler calls once per denoising step (gaussian_diffusion.py:350 -> denoiser.py:284-287).
value = renders / s = B * 4 * n_gpus / t_step, inputs resident in HBM, synthetic data, random-init weights.

    python bench.py [--gpus N --steps K --warmup W]            (N > 1: launched by torch.distributed.run, one rank per GPU)

Prints ONE JSON line (rank 0).  Extra objects: `roofline` (dominant kernel, HIP events on the launch stream inside the
timed region) and `cpu_baseline` (the CPU oracle timed on the host cores on ONE sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "open-diffusiongs_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np
import torch

PEAK_BF16_MFMA = 2.5e15      # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PROF_KINDS = {"attention": 1, "gemm_qkv": 2, "gemm_gate_residual": 3, "gemm_fc1_gelu": 4, "layernorm": 5}


def dit_flops(L, width=1024, layers=24, n_img_tokens=None, patch=8, gs_ch=14):
    """algorithmic forward FLOPs per sample."""
    n_img = L - 2 if n_img_tokens is None else n_img_tokens
    per_layer = L * (2 * width * (3 * width + width + 8 * width) + 4 * L * width) + 2 * width * 6 * width
    return layers * per_layer + n_img * 2 * (9 * patch * patch) * width + n_img * 2 * width * (patch * patch * gs_ch) \
        + 2 * (256 * width + width * width)


def kernel_flops(kind, L, B, width=1024):
    """Algorithmic FLOPs of ONE launch of the profiled kernel class (valid tokens only; padding rows are overhead)."""
    if kind == "attention":
        return 4.0 * L * L * width * B
    n = {"gemm_qkv": 3 * width * width, "gemm_gate_residual": (width * width + 4 * width * width) / 2.0,
         "gemm_fc1_gelu": 4 * width * width}[kind]
    return 2.0 * L * n * B


def measured_traffic(kernel, batch):
    """HBM bytes per launch of the roofline kernel from the committed PMC passes (profiles/r01_pmc_traffic.json; collected
    with tools/pmc_run.py exactly as MI355X_MICROARCH.md prescribes).  Only valid for the shape it was measured on (batch 1)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))
        v = d.get(kernel, {}).get("traffic_bytes_per_launch")
        return int(v) if (v is not None and batch == 1) else None
    except Exception:
        return None


def synth_batch(B, V, res, device, seed):
    from dgs_amd import synth
    return synth.make_batch(B, res, V=V, device=device, seed=seed, with_t=True)


def cpu_baseline(model, batch, t, res, V, hip_gaussians, hip_render):
    """CPU leg (rank 0, N = 1 only): the oracle -- fp32 PyTorch-CPU restatement of the denoiser on all host cores + the
    C++ restatement of the reference rasterizer (1 thread) -- timed on ONE sample of the same workload (1 DiT step +
    V rasterizations).  Also returns the PSNR of the HIP render against the oracle render of the same Gaussians."""
    from oracle import dit_oracle as D
    from oracle import raster_oracle as RO
    RO.build()
    cores = min(os.cpu_count() or 1, 64)           # more threads than this only adds oversubscription at L=4098
    torch.set_num_threads(cores)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cpu = {k: v[:1].cpu() for k, v in batch.items()}
    # bounded sample: 4 of the 24 (identical-cost) DiT blocks are timed and scaled by 6; tokenizer + heads are timed in
    # full (they are inside both runs, so subtract the 0-block run)
    def run(n_layers):
        cfg = D.Cfg(num_layers=n_layers)
        t0 = time.perf_counter()
        with torch.no_grad():
            g_, _ = D.image_to_gaussians(sd, cfg, cpu["image"], cpu["ray_o"], cpu["ray_d"], t[:1].cpu())
        return time.perf_counter() - t0, g_
    t_0, _ = run(0)
    t_4, g = run(4)
    t_dit = t_0 + (t_4 - t_0) * 6.0
    view, proj, campos, tanfov = D.camera_matrices(cpu["c2w"][0], cpu["fxfycxcy"][0], res, res)
    act = lambda gm: dict(xyz=gm["xyz"][0].numpy(), shs=gm["features"][0].numpy(),
                          op=torch.sigmoid(gm["opacity"][0]).numpy(), sc=torch.exp(gm["scaling"][0]).numpy(),
                          rot=torch.nn.functional.normalize(gm["rotation"][0]).numpy())
    a = act(g)
    t0 = time.perf_counter()
    for v in range(V):
        o = RO.RasterOracle()
        o.forward(np.ones(3, np.float32), a["xyz"], a["op"], view[v].numpy(), proj[v].numpy(), campos[v].numpy(),
                  float(tanfov[v, 0]), float(tanfov[v, 1]), res, res, shs=a["shs"], scales=a["sc"], rotations=a["rot"])
    t_raster = time.perf_counter() - t0
    # PSNR (utils/losses.py:399-402) of the HIP render vs the oracle render of the SAME (HIP-produced) Gaussians, view 0
    h = act({k: v.cpu() for k, v in hip_gaussians.items()})
    o = RO.RasterOracle()
    o.forward(np.ones(3, np.float32), h["xyz"], h["op"], view[0].numpy(), proj[0].numpy(), campos[0].numpy(),
              float(tanfov[0, 0]), float(tanfov[0, 1]), res, res, shs=h["shs"], scales=h["sc"], rotations=h["rot"], exp_mode=1)
    ref = np.clip(o.get("out_color"), 0, 1)
    mine = np.clip(hip_render[0, 0].cpu().numpy(), 0, 1)
    mse = float(np.mean((ref.astype(np.float64) - mine) ** 2))
    psnr = 200.0 if mse == 0 else -10.0 * np.log10(mse)
    return dict(value=V / (t_dit + t_raster), unit="renders/s", cores=cores, kind="port",
                sample=f"1 sample: DiT step at L=4098 extrapolated from 4 of 24 blocks ({t_4:.1f} s measured -> {t_dit:.1f} s, "
                       f"torch-CPU fp32 oracle, {cores} threads) + {V} oracle rasterizations at {res}^2 ({t_raster:.1f} s "
                       f"measured, C++ oracle, 1 thread)"), float(psnr)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="samples per GPU per step (pipline_obj.py samples one object)")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--roofline-kernel", default="attention", choices=sorted(PROF_KINDS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # RCCL
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from dgs_amd import denoiser as dn
    model = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="relative_plk"), device=dev)
    model.reset_parameters(seed=0)          # every rank the same random-init weights (pure data parallel inference)
    B, V, res = a.batch, a.views, a.res
    batch, t = synth_batch(B, V, res, dev, seed=rank)
    eng = model.engine()
    L = eng.num_tokens(V, res, res)

    def step(prof=None):
        # == DGSDenoiser.forward (denoiser.py:284-287); the profiling hook only adds event records on the stream
        params, _ = eng.image_to_gaussians(batch["image"], batch["ray_o"], batch["ray_d"], t, prof=prof)
        pc = params.pop("prof_count", 0)
        p = dn.AttrDict(params)
        rendered = model.render_gaussians(p, batch["c2w"], batch["fxfycxcy"], res, res)
        return rendered, model.prepare_to_save(p), pc

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    for _ in range(a.warmup):
        step()
    per_step = {"attention": 24, "gemm_qkv": 24, "gemm_gate_residual": 48, "gemm_fc1_gelu": 24, "layernorm": 48}[a.roofline_kernel]
    events = [[torch.cuda.Event(enable_timing=True) for _ in range(2 * per_step)] for _ in range(a.steps)]
    for ev in events:           # materialise the HIP event handles before the timed region
        for e in ev:
            e.record()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_prof = 0
    for i in range(a.steps):
        rendered, gaussians, pc = step(prof=(PROF_KINDS[a.roofline_kernel], events[i]))
        n_prof += pc
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    # Informational: the reference's 30-step sampling loop end to end -- DGSDenoiser.forward + the device sampler
    # step (dgs_amd/sampler.py) per iteration; one untimed loop, one timed.  Not part of `value`.
    from dgs_amd import sampler as sm
    diffusion = sm.create_diffusion("30", device=dev)
    loop_batch = dict(batch)
    loop_ms = None
    for timed in (False, True):
        loop_batch["image"] = batch["image"].clone()
        loop_batch["image_noisy"] = torch.randn_like(batch["image"][:, 1:])
        torch.cuda.synchronize()
        l0 = time.perf_counter()
        diffusion.p_sample_loop(model, loop_batch)
        torch.cuda.synchronize()
        if timed:
            loop_ms = (time.perf_counter() - l0) * 1e3

    if rank == 0:
        ms = elapsed / a.steps * 1e3
        value = B * V * world / (elapsed / a.steps)
        kern_ms = [events[i][2 * j].elapsed_time(events[i][2 * j + 1]) for i in range(a.steps) for j in range(per_step)]
        avg_s = float(np.mean(kern_ms)) * 1e-3
        achieved = kernel_flops(a.roofline_kernel, L, B) / avg_s / 1e12
        out = {
            "metric": "novel-view renders/sec (DiT step + GS raster) at 256^2", "value": round(value, 2), "unit": "renders/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"obj-{res} sampling step (BASELINE.json configs[2]): DGSDenoiser.forward = DiT "
                                   f"image_to_gaussians (width 1024, 24 blocks, L={L}, bf16 MFMA, fp32 accumulate) + {V} fp32 "
                                   f"rasterizations of P={2 + V * res * res} Gaussians at {res}^2 per sample; random-init weights",
                       "batch_per_gpu": B, "views": V, "resolution": res, "tokens": L, "gaussians": 2 + V * res * res,
                       "dit_tflop_per_sample": round(dit_flops(L) / 1e12, 3), "parallelism": f"dp{world}"},
            "roofline": {"kernel": a.roofline_kernel, "bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_MFMA / 1e12,
                         "unit": "TFLOP/s", "frac": round(achieved * 1e12 / PEAK_BF16_MFMA, 4),
                         "traffic": measured_traffic(a.roofline_kernel, B) if res == 256 and V == 4 else None,
                         "launches_timed": len(kern_ms), "avg_launch_us": round(avg_s * 1e6, 2)},
        }
        out["sampling_loop_30_steps"] = {"ms_per_loop": round(loop_ms, 2), "renders_per_s": round(B * V * 30 / (loop_ms * 1e-3), 1),
                                         "note": "per GPU; informational, not part of value"}
        if world == 1 and not a.no_cpu_baseline:
            final = {k: getattr(gaussians[0], "_" + n)[None] for k, n in
                     (("xyz", "xyz"), ("features", "features_dc"), ("scaling", "scaling"), ("rotation", "rotation"), ("opacity", "opacity"))}
            base, psnr = cpu_baseline(model, batch, t, res, V, final, rendered)
            out["cpu_baseline"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in base.items()}
            out["psnr_vs_oracle_db"] = round(psnr, 2)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
