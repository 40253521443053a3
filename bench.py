#!/usr/bin/env python
"""Contract benchmark: novel-view renders/sec of one DiffusionGS sampling step (DiT step + GS raster) at 256^2.

A "step" is one `DGSDenoiser.forward(input_batch, t)` of the object model (width 1024, 24 blocks, patch 8 -- the shipped
`diffusion-gs-model`): `image_to_gaussians` on B samples x 4 input views (L = 4098 tokens, 4.139 TFLOP / sample, bf16
MFMA) followed by the rasterization of every sample's P = 262,146 per-pixel Gaussians into its 4 views (fp32) -- what
the reference's sampler calls once per denoising step (gaussian_diffusion.py:350 -> denoiser.py:284-287).
value = renders / s = B * 4 * n_gpus / t_step, inputs resident in HBM, synthetic data, random-init weights.

    python bench.py [--gpus N --steps K --warmup W]

With N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks (torch.distributed.run, 127.0.0.1, one
per GPU, RCCL); launched BY torch.distributed.run (the driver's way) it reads RANK / LOCAL_RANK / WORLD_SIZE.

Prints ONE JSON line (rank 0).  Extra objects next to the contract keys:
  roofline      dominant DiT kernel, HIP events on the launch stream inside the timed region; `traffic` from the PMC passes of
                tools/pmc_traffic.py IF they were taken on exactly these kernel sources (else null)
  raster        the rasterizer on its HBM and fp32-VALU rooflines, forward and forward+backward, in the regime the step renders
                (random-init Gaussians) and the trained-like regime; bytes and pair evaluations are those the call
                WALKS (from its own state: tile_work / list_len / tile_stats), not those of the N instances a full list would hold
  scene_512     BASELINE configs[4] at inference: the scene model's sampling step at 512^2 (L = 16,386, P = 1,048,578), informational
  train_step    BASELINE configs[3]: B = 4 samples / GPU, 10 rendered views, forward + backward + gradient all-reduce
                (overlapped, RCCL) + AdamW step + weight refresh, `DataParallelTrainer.step` end to end
  cpu_baseline  the CPU oracle on the host cores on ONE sample of the same workload (N = 1 only)
`--mode train` makes the training step the timed region of the line instead (metric: training samples/s).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time
import types

# dmabuf IPC is the only mode this pool's host driver supports: without it RCCL's cross-process buffer exchange fails with
# `hipIpcGetMemHandle: invalid argument`.  Set before the HIP runtime loads, whoever launched this process.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "open-diffusiongs_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_BF16_MFMA = 2.5e15      # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_HBM = 8.0e12            # spec, same guide "HBM3E peak BW"
PROF_KINDS = {"attention": 1, "gemm_qkv": 2, "gemm_gate_residual": 3, "gemm_fc1_gelu": 4, "layernorm": 5, "gemm_proj": 6, "gemm_fc2": 7}


MODEL_CFG = dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="relative_plk")     # the shipped diffusion-gs-model
DRY = {"on": False, "lib": None}    # --dry-run-cpu: CPU emulator build + gloo, a tiny model: exercises the N > 1 plumbing, measures nothing
DIST = {"on": False}                 # a process group exists (world > 1, or --force-dist on one GPU): barriers + max-over-ranks reductions are issued


def _sync():
    if not DRY["on"]:
        import torch
        torch.cuda.synchronize()


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
         "gemm_fc1_gelu": 4 * width * width, "gemm_proj": width * width, "gemm_fc2": 4 * width * width}[kind]
    return 2.0 * L * n * B


# A PMC measurement belongs to the sources of the kernels it counted: the whole csrc/ tree, or -- when only another family's
# files changed since -- the family's own files (the attention kernel does not compile from raster_*.hip, nor the reverse).
FAMILIES = {
    "dit": lambda f: f.startswith("dit_") or f in ("dgs_device.h", "raster_state.h"),
    "raster": lambda f: f.startswith("raster_") or f in ("camera.hip", "dgs_device.h"),
}


def kernel_source_sha(family=None, read=None, names=None):
    """SHA-256 over the csrc/ sources and headers (of one family, or all).  `read` / `names` let tools/pmc_traffic.py hash the
    files of a git revision instead of the working tree."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "open-diffusiongs_amd", "csrc")
    read = read or (lambda f: open(os.path.join(d, f), "rb").read())
    for f in sorted(names if names is not None else os.listdir(d)):
        if f.endswith((".hip", ".h")) and (family is None or FAMILIES[family](f)):
            h.update(f.encode())
            h.update(read(f))
    return h.hexdigest()


def pmc_traffic(family):
    """profiles/pmc_traffic.json (tools/pmc_traffic.py: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over this
    very script, FETCH doubled per MI355X_MICROARCH.md).  Only returned when it was measured on the current sources of `family`."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if d.get("kernel_source_sha") == kernel_source_sha() or d.get("family_sha", {}).get(family) == kernel_source_sha(family):
            return d
        return None
    except Exception:
        return None


def respawn(a):
    """`python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU, RCCL over xGMI)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def cpu_baseline(model, batch, t, res, V, hip_gaussians, hip_render):
    """CPU leg (rank 0, N = 1 only): the oracle on ONE sample of the same workload -- the fp32 PyTorch-CPU restatement of the
    denoiser, all 24 blocks, all host cores, and the C++ restatement of the reference rasterizer with OpenMP over Gaussians /
    tiles on all host cores.  Also returns the PSNR of the HIP render against the oracle render of the same
    Gaussians."""
    import numpy as np
    import torch
    from oracle import dit_oracle as D
    from oracle import raster_oracle as RO
    RO.build()
    cores = os.cpu_count() or 1
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cpu = {k: v[:1].cpu() for k, v in batch.items()}
    # The DiT leg is 97 % of the baseline's time.  SURVEY.md 8d asks for all host cores; beyond ~64 threads torch's CPU GEMMs at L = 4098
    # mostly add oversubscription (256 threads: 89 s against 12.7 s with 64 on the round-5 box).  To keep the leg inside its ~30 s budget
    # both thread counts are timed on a TWO-block probe of the same sample (the 24 blocks are identical work), the full 24 blocks run once
    # with the faster count; `value` / `cores` are that run, the probe times of both counts are stated.
    probe_cfg = D.Cfg(num_layers=2)
    probe = {}
    for n_thr in sorted({cores, min(cores, 64)}, reverse=True):
        torch.set_num_threads(n_thr)
        RO.set_threads(n_thr)                      # torch and the oracle share one OpenMP runtime: keep its pool at the torch size for the DiT leg
        t0 = time.perf_counter()
        with torch.no_grad():
            D.image_to_gaussians(sd, probe_cfg, cpu["image"], cpu["ray_o"], cpu["ray_d"], t[:1].cpu())
        probe[n_thr] = time.perf_counter() - t0
    dit_threads = min(probe, key=probe.get)
    torch.set_num_threads(dit_threads)
    RO.set_threads(dit_threads)
    t0 = time.perf_counter()
    with torch.no_grad():
        g, _ = D.image_to_gaussians(sd, D.Cfg(), cpu["image"], cpu["ray_o"], cpu["ray_d"], t[:1].cpu())
    t_dit = time.perf_counter() - t0
    dit_runs = probe
    view, proj, campos, tanfov = D.camera_matrices(cpu["c2w"][0], cpu["fxfycxcy"][0], res, res)
    act = lambda gm: dict(xyz=gm["xyz"][0].numpy(), shs=gm["features"][0].numpy(),
                          op=torch.sigmoid(gm["opacity"][0]).numpy(), sc=torch.exp(gm["scaling"][0]).numpy(),
                          rot=torch.nn.functional.normalize(gm["rotation"][0]).numpy())
    a = act(g)
    raster_threads = RO.set_threads(0)             # all host cores for the rasterizer leg
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
    return dict(value=V / (t_dit + t_raster), unit="renders/s", cores=dit_threads, kind="port",
                sample=f"1 sample of the same step: DiT forward at L=4098, all 24 blocks ({t_dit:.1f} s, torch-CPU fp32 oracle, "
                       f"{dit_threads} threads) + {V} oracle rasterizations at {res}^2 ({t_raster:.2f} s, C++ oracle, OpenMP over "
                       f"Gaussians / tiles, {raster_threads} threads); `cores` = the threads of the faster DiT leg (97 % of the time); "
                       f"two-block probe of the DiT leg by thread count: " + ", ".join(f"{k} threads {v:.1f} s" for k, v in sorted(dit_runs.items())) + f"; host has {cores} cores"), float(psnr)


PEAK_FP32_VALU = 157.3e12   # same guide, "Peak FP32 (vector)"
FLOP_PER_PAIR_FWD, FLOP_PER_PAIR_BWD = 25.0, 65.0    # fp32 operations of one (pixel, Gaussian) evaluation, forward.cu:332-358 / backward.cu:463-532


def raster_roofline(dev, res, V, iters=10):
    """Rasterizer on its two rooflines: forward and forward+backward of V views of the DiffusionGS-shaped synthetic scenes of
    SURVEY.md 8d, timed with HIP events on the launch stream (the rasterizer launches on torch's current stream).
    `hbm`: algorithmic bytes of what the call WALKS -- per view 104 P + 20 HW (forward; + 251 P + 20 HW backward), per list
      entry walked 4 (index) + 40 (record gather), per entry listed 4, per depth rank scanned 4 (scan form), per instance of a
      materialised list 44 more (emit + sort: the list forms) -- the counts come from the call's own state (`tile_work`,
      `list_len`, `tile_stats`), not from N: a saturated tile walks a few per cent of its list.
    `valu`: the blend loops are fp32-VALU bound: pair evaluations actually executed (16 x the cell-list entries walked) x 25 (65 in
      the backward) fp32 operations / call time against the 157.3 TFLOP/s vector peak; `lane_use` = executed pairs / lane slots the
      waves issued (four 16-lane rows walk their cells' lists in lockstep)."""
    import numpy as np
    import torch
    from dgs_amd import _native, cameras, synth
    from dgs_amd.raster import RasterBackend, default_backend, render_views_autograd
    be = default_backend()
    # pair-evaluation counters (tile_stats) exist in the tools' build only (csrc/raster_common.h kRasterStats): ONE untimed forward +
    # backward on it counts what the call walks -- the same kernels minus the counters are what is timed below
    instr = os.path.join(ROOT, "open-diffusiongs_amd", "lib", "libdgs_hip_instr.so")
    be_count = None
    if os.path.exists(instr):
        try:
            be_count = RasterBackend(lib=_native.open_library(instr), exact_exp=be.exact_exp)
        except RuntimeError as e:                 # a stale tools' build (ABI mismatch): no pair counts, the timings stand
            print(f"[bench] {e}", file=sys.stderr, flush=True)
    out = {"views": V, "resolution": res, "exact_exp": bool(be.exact_exp), "pair_counts_from": "lib/libdgs_hip_instr.so" if be_count else None,
           "hbm": {"bound": "hbm", "peak": PEAK_HBM / 1e9, "unit": "GB/s"}, "valu": {"bound": "valu", "peak": PEAK_FP32_VALU / 1e12, "unit": "TFLOP/s"}}
    tt = lambda x: torch.as_tensor(np.ascontiguousarray(x), dtype=torch.float32, device=dev)
    traffic = pmc_traffic("raster")
    T = ((res + 15) // 16) ** 2
    for regime in ("init", "trained"):
        sc = synth.gaussian_scene(res, regime=regime, seed=0, activated=False)
        leaves = [tt(sc[k])[None].requires_grad_(True) for k in ("xyz", "shs", "scales", "rotations", "opacities")]
        c2w = tt(cameras.ring_cameras(V, phase_deg=10))[None]
        k = tt(cameras.default_fxfycxcy(res)).expand(1, V, 4).contiguous()
        w = torch.randn(1, V, 3, res, res, device=dev) / (3 * res * res)
        P = int(leaves[0].shape[1])
        view, proj, campos, tanfov = be.cameras_from_c2w(c2w, k, res, res)
        det = [x.detach() for x in leaves]
        fwd_on = lambda b, planned: b.forward_views(torch.ones(3, device=dev), det[0], None, det[4].reshape(1, -1), det[2], det[3], 1.0, None, view, proj,
                                                    campos, tanfov, 0.0, 0.0, res, res, det[1], 0, False, False, views_per_set=V, raw_activations=True,
                                                    planned=planned)
        fwd = lambda: fwd_on(be, True)          # the product's call: planned, no host synchronisation after the first call of the shape
        bc = be_count or be
        N, _color, radii, geom, binning, img = fwd_on(bc, False)
        N = int(N)
        rd = lambda name, cnt: bc.state_read(name, P, res, res, V, N, geom, binning, img, torch.int32, cnt).long()
        bc.backward_views(torch.ones(3, device=dev), det[0], radii, None, det[4].reshape(1, -1), det[2], det[3], 1.0, None, view, proj, campos,
                          tanfov, 0.0, 0.0, w.reshape(V, 3, res, res), det[1], 0, geom, N, binning, img, False, views_per_set=V, raw_activations=True)
        sf, sb = rd("tile_stats", V * T * 4).reshape(V * T, 4).sum(0).tolist(), rd("tile_stats_bwd", V * T * 4).reshape(V * T, 4).sum(0).tolist()
        walked, listed = int(rd("tile_work", V * T).sum()), int(rd("list_len", V * T).sum())
        scanned = int(rd("tile_scanned", V * T).sum())
        counted = be_count is not None
        inst = 0 if scanned else N                                       # the list forms materialise and sort all N instances
        bytes_f = V * (104 * P + 20 * res * res) + 44 * walked + 4 * listed + 4 * scanned + 44 * inst
        bytes_b = V * (251 * P + 20 * res * res) + 44 * walked
        flop_f, flop_b = FLOP_PER_PAIR_FWD * 16 * sf[0], FLOP_PER_PAIR_BWD * 16 * sb[0]

        def fb():
            for x in leaves:
                x.grad = None
            render_views_autograd(be, *leaves, res, res, c2w, k).backward(w)

        rec = {"P": P, "N_per_view": N // V, "binning": "scan" if scanned else "list", "entries_walked_per_view": walked // V,
               "entries_listed_per_view": listed // V, "ranks_scanned_per_view": scanned // V}
        if counted:
            rec.update({"pair_evals_forward": 16 * sf[0], "pair_evals_backward": 16 * sb[0],
                        "lane_use_forward": round(16 * sf[0] / max(64 * sf[1], 1), 3), "lane_use_backward": round(16 * sb[0] / max(64 * sb[1], 1), 3)})
        for name, fn, nbytes, flops in (("forward", fwd, bytes_f, flop_f), ("forward_backward", fb, bytes_f + bytes_b, flop_f + flop_b)):
            for _ in range(3):
                fn()
                # the plan picks a call's ordering form from the statistics that have ARRIVED: let this regime's arrive before the timed
                # calls (a host that runs 13 calls ahead would time the previous regime's form -- valid, but not this scene's)
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            rec[name] = {"ms": round(ms, 4), "views_per_s": round(V / ms * 1e3, 1), "algorithmic_bytes": int(nbytes),
                         "hbm_achieved": round(nbytes / ms / 1e6, 1), "hbm_frac": round(nbytes / (ms * 1e-3) / PEAK_HBM, 4)}
            if counted:
                rec[name].update({"valu_achieved": round(flops / (ms * 1e-3) / 1e12, 2), "valu_frac": round(flops / (ms * 1e-3) / PEAK_FP32_VALU, 4),
                                  "pair_evals_per_s": round((16 * sf[0] + (16 * sb[0] if name != "forward" else 0)) / (ms * 1e-3), 0)})
            if traffic and traffic.get("raster", {}).get(regime, {}).get(name) is not None:
                rec[name]["traffic"] = traffic["raster"][regime][name]
        out[regime] = rec
    if traffic and traffic.get("fetch_calibration"):
        out["fetch_calibration"] = traffic["fetch_calibration"]
    return out


def scene_512(dev, steps=5):
    """BASELINE configs[4] at inference: the scene model (`diffusion-gs-model-scene`, plk ray embedding) at 512^2, one sample, 4 views:
    L = 16,386 tokens, 36.3 TFLOP per DiT step, P = 1,048,578 Gaussians rendered into 4 views.  Informational, not part of `value`."""
    import torch
    from dgs_amd import denoiser as dn, synth
    model = dn.DGSDenoiserScene(dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="plk"), device=dev)
    model.reset_parameters(seed=0)
    batch, t = synth.make_batch(1, 512, V=4, device=dev, seed=0, with_t=True)
    eng = model.engine()
    L = eng.num_tokens(4, 512, 512)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * 24)]
    for e in ev:
        e.record()

    def step(prof=None):
        with torch.no_grad():
            params, _ = eng.image_to_gaussians(batch["image"], batch["ray_o"], batch["ray_d"], t, prof=prof)
            params.pop("prof_count", 0)
            return model.render_gaussians(dn.AttrDict(params), batch["c2w"], batch["fxfycxcy"], 512, 512)

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        step(prof=(PROF_KINDS["attention"], ev) if i == steps - 1 else None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    attn_ms = sum(ev[2 * j].elapsed_time(ev[2 * j + 1]) for j in range(24)) / 24
    return {"workload": "scene-512 sampling step (BASELINE.json configs[4] at inference): DiT L=%d + 4 rasterizations of P=%d Gaussians at 512^2" % (L, 2 + 4 * 512 * 512),
            "ms_per_step": round(ms, 2), "renders_per_s": round(4 / (ms * 1e-3), 1), "dit_tflop_per_sample": round(dit_flops(L) / 1e12, 2),
            "step_frac_of_peak": round(dit_flops(L) / (ms * 1e-3) / PEAK_BF16_MFMA, 4),
            "attention": {"avg_launch_us": round(attn_ms * 1e3, 1), "achieved_tflops": round(4.0 * L * L * 1024 / (attn_ms * 1e-3) / 1e12, 1),
                          "frac": round(4.0 * L * L * 1024 / (attn_ms * 1e-3) / PEAK_BF16_MFMA, 4)}}


def batch_sweep(dev, model, res, V, batches=(2, 4), steps=8):
    """Informational: the same sampling step with more than one sample per GPU (the reference samples one object at a time,
    pipline_obj.py; a serving deployment would batch).  renders/s = B x V / step time; eager steps, events on the stream."""
    import torch
    from dgs_amd import synth
    out = {}
    for B in batches:
        batch, t = synth.make_batch(B, res, V=V, device=dev, seed=7, with_t=True)
        with torch.no_grad():
            for _ in range(2):
                model(batch, t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                model(batch, t)
            e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        out[f"batch_{B}"] = {"ms_per_step": round(ms, 3), "renders_per_s": round(B * V / (ms * 1e-3), 1)}
    return out


def train_bench(a, dev, rank, world, steps, warmup, scene=None):
    """BASELINE configs[3] (train_obj_stage1.sh, diffusionGS_rel.yaml): per GPU B samples x 4 input views at 256^2, `--train-views`
    rendered views; one `DataParallelTrainer.step` = DiT forward (activations saved) + rasterization + MSE + rasterizer backward +
    DiT backward with the gradient all-reduce of the 460 M parameters overlapped bucket by bucket (RCCL when world > 1) + fused
    AdamW + in-place refresh of the engine's bf16 / transposed weights.
    scene = dict(batch, views, rendered_views, res, recompute): BASELINE configs[4] (train_scene_stage2.sh, diffusionGS_scene_512.yaml:
    `diffusion-gs-model-scene`, batch_size 12 per rank, sel_views + 1 = 4 input views, 3 + 4 = 7 rendered views at 512^2, AdamW
    lr 3e-5 betas (0.9, 0.95) eps 1e-6, gradient_clip_val 0.5) -- the 12 samples of a rank run as micro-batches of 4 inside ONE
    optimizer step (dgs_amd/train.py); recompute=True forces the reference's per-block checkpointing (`use_checkpoint: true`,
    denoiser.py:343-354), None leaves it to the engine's policy (recompute only if the saved activations would not fit)."""
    import numpy as np
    import torch
    from dgs_amd import cameras, denoiser as dn, synth
    from dgs_amd.train import DataParallelTrainer
    B, V, res, RV = a.train_batch, a.views, a.res, a.train_views
    cfg, hyper = MODEL_CFG, dict(lr=1e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=0.05)
    if scene is not None:
        B, V, res, RV = scene["batch"], scene["views"], scene["res"], scene["rendered_views"]
        cfg = dict(MODEL_CFG, ray_pe_type="plk", use_checkpoint=True)
        hyper = dict(lr=3e-5, betas=(0.9, 0.95), eps=1e-6, weight_decay=0.05)
        model = dn.DGSDenoiserScene(cfg, device=dev, lib=DRY["lib"])
        if scene.get("recompute") is True:
            model.activation_budget_bytes = 0               # nothing fits: every block is recomputed in the backward
        elif scene.get("recompute") is False:
            model.activation_budget_bytes = 1 << 62
    else:
        model = dn.DGSDenoiser(MODEL_CFG, device=dev, lib=DRY["lib"])
    model.reset_parameters(seed=0)          # identical replicas on every rank
    model = model.to(dev)                   # fp32 master parameters + optimizer state on the GPU
    model.train()
    if a.optimizer == "fused":      # AdamW + refresh of the engine's bf16 / transposed weight copies in one launch (include/dgs_optim.h)
        from dgs_amd.optim import FusedAdamW
        opt = FusedAdamW(model, **hyper)
    else:                           # torch's multi-tensor AdamW, then ~600 torch copies for the refresh
        opt = torch.optim.AdamW(model.parameters(), fused=not DRY["on"], **hyper)
    # gradient_clip_val: 0.5 (configs/diffusionGS_rel.yaml:76-77): the reference's optimizer step is clip + AdamW
    tr = DataParallelTrainer(model, opt, bucket_bytes=(a.bucket_mb << 20) if a.bucket_mb > 0 else None, compress=a.grad_exchange if a.grad_exchange != "fp32" else None,
                             force_collectives=a.force_dist, max_grad_norm=a.clip if a.clip > 0 else None)
    batch, t = synth.make_batch(B, res, V=V, device=dev, seed=100 + rank, with_t=True)
    rc2w = torch.tensor(np.stack([cameras.ring_cameras(RV, phase_deg=5.0 + 7 * b) for b in range(B)])).to(dev)
    rk = torch.tensor(cameras.default_fxfycxcy(res)).expand(B, RV, 4).contiguous().to(dev)
    target = torch.rand(B, RV, 3, res, res, device=dev)
    for _ in range(warmup):
        loss = tr.step(batch, t, target, rc2w, rk)
    if DIST["on"]:
        torch.distributed.barrier()
    _sync()
    if not DRY["on"]:
        tr.reducer.timing = {}                      # events around finish(): GPU time the compute stream waits for collectives
    rbe = model.gs_renderer.backend() if hasattr(getattr(model, "gs_renderer", None), "backend") else None
    wait0 = getattr(rbe, "verify_wait_s", 0.0)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(batch, t, target, rc2w, rk)
    t_enq = time.perf_counter()
    _sync()
    t_local = time.perf_counter() - t0
    if DIST["on"]:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    verify_wait_ms = (getattr(rbe, "verify_wait_s", 0.0) - wait0) / steps * 1e3
    # one more step, untimed, with an event behind every block_done callback of the backward: when the callback ran on the host against
    # when the device got there -- is the host at least a block ahead when the callbacks launch the buckets' collectives?
    lead = None
    if not DRY["on"]:
        tr.lead_probe = []
        _sync()
        h0 = time.perf_counter()
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        tr.step(batch, t, target, rc2w, rk)
        _sync()
        probe, tr.lead_probe = tr.lead_probe, None
        if len(probe) >= 3:
            dev_ms = [e0.elapsed_time(ev) for _, _, ev in probe]
            host_ms = [(th - h0) * 1e3 for _, th, _ in probe]
            leads = sorted(d - h for d, h in zip(dev_ms, host_ms))
            gaps = sorted(b - a2 for a2, b in zip(dev_ms[:-1], dev_ms[1:]))
            blk = gaps[len(gaps) // 2]
            lead = {"callbacks": len(probe), "device_behind_host_ms": {"min": round(leads[0], 2), "median": round(leads[len(leads) // 2], 2), "max": round(leads[-1], 2)},
                    "block_backward_ms_median": round(blk, 3), "blocks_ahead_min": round(leads[0] / blk, 1) if blk > 0 else None,
                    "collectives_in_callbacks": bool(tr.reducer.active),
                    "note": "per block_done callback of one untimed step: device completion time of the block's kernels minus the host time of the callback"}
    rank_ms = [t_local / steps * 1e3]
    ranks_seen = 1
    if DIST["on"]:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tl = torch.tensor([t_local], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(tl) for _ in range(world)]
        torch.distributed.all_gather(allt, tl)
        rank_ms = [float(x.item()) / steps * 1e3 for x in allt]
        one = torch.ones(1, dtype=torch.float32, device=dev)
        torch.distributed.all_reduce(one)               # every rank adds 1: the number of ranks that really took part
        ranks_seen = int(one.item())
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    exposed = tr.reducer.exposed_ms() if not DRY["on"] else None
    eng = model.engine()
    L = eng.num_tokens(V, res, res)
    ms = elapsed / steps * 1e3
    recompute = bool(eng._train.get("recompute"))
    flops = (4 if recompute else 3) * dit_flops(L, MODEL_CFG["width"], MODEL_CFG["num_layers"]) * B
    log = tr.reducer.launch_log
    trainer_backend = model.gs_renderer.backend() if hasattr(getattr(model, "gs_renderer", None), "backend") else None   # the trainer's own (dgs_amd/train.py): gone from the model after close()
    tr.close()
    micro = min(B, getattr(model, "MAX_DIFFERENTIABLE_BATCH", 4))
    mem = None
    if not DRY["on"]:
        free_b, total_b = torch.cuda.mem_get_info(dev)
        mem = {"peak_allocated": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1), "reserved": round(torch.cuda.memory_reserved(dev) / 2 ** 30, 1),
               "free": round(free_b / 2 ** 30, 1), "total": round(total_b / 2 ** 30, 1)}
    raster_note = None
    if trainer_backend is not None:
        be = trainer_backend
        raster_note = {"deterministic_backward": bool(getattr(be, "last_backward_deterministic", False)),
                       "plans": [{"capacity": pl.capacity, "seen_max": pl.seen_max, "calls": dict(pl.calls)} for pl in be._plans.values()][-3:]}
    head = {"gpu_memory_gib": mem, "raster": raster_note} if scene is None else {
        "workload": "scene-512 training step (BASELINE.json configs[4]: train_scene_stage2.sh, diffusionGS_scene_512.yaml): DGSDenoiserScene, "
                    "%d samples / GPU as micro-batches of %d inside one optimizer step, %d input + %d rendered views at %d^2, L=%d, P=%d" %
                    (B, micro, V, RV, res, L, 2 + V * res * res),
        "micro_batch": micro, "resolution": res, "tokens": L, "dit_tflop_per_sample_fwd": round(dit_flops(L) / 1e12, 2),
        "gpu_memory_gib": mem, "raster": raster_note}
    return {**head, "ms_per_step": round(ms, 2), "samples_per_s": round(B * world / (ms * 1e-3), 2), "batch_per_gpu": B, "rendered_views": RV,
            "steps": steps, "warmup": warmup, "loss": round(float(loss), 6), "recompute": recompute, "optimizer": a.optimizer,
            "dit_tflops_per_gpu": round(flops / (ms * 1e-3) / 1e12, 1), "frac_of_bf16_peak": round(flops / (ms * 1e-3) / PEAK_BF16_MFMA, 4),
            "saved_activation_gib": round(eng._train["saved"].numel() / 2 ** 30, 2),
            "gradient_clip_val": a.clip if a.clip > 0 else None,
            "grad_norm_before_clip": round(float(tr.last_grad_sumsq.sqrt()), 6) if tr.last_grad_sumsq is not None else None,
            "host_enqueue_ms_per_step": round((t_enq - t0) / steps * 1e3, 2),
            # ... of which the host spent waiting for the device inside the rasterizer's verified calls (plans at risk: every training render;
            # dgs_amd/raster.py) -- the rest is enqueue work proper
            "host_verify_wait_ms_per_step": round(verify_wait_ms, 2),
            "host_enqueue_work_ms_per_step": round((t_enq - t0) / steps * 1e3 - verify_wait_ms, 2),
            "host_lead_in_backward": lead,
            "per_rank_ms": {"min": round(min(rank_ms), 2), "max": round(max(rank_ms), 2)}, "ranks_seen": ranks_seen,
            "parameter_broadcast_bytes": int(tr.broadcast_bytes),
            "allreduce": {"world": world, "collectives_issued": bool(tr.reducer.active), "buckets": len(tr.reducer.bounds), "exchange": a.grad_exchange,
                          "exposed_ms_per_step": round(exposed, 3) if exposed is not None else None,
                          "bucket_mib": [round((e - b) * 4 / 2 ** 20, 1) for b, e in tr.reducer.bounds],
                          "launched_during_backward": sum(1 for _, tag in log if isinstance(tag, int)),
                          "last_bucket_mib": round((tr.reducer.bounds[-1][1] - tr.reducer.bounds[-1][0]) * 4 / 2 ** 20, 1),
                          "gradient_bytes": int(tr.fg.flat.numel() * 4)},
            "note": "DataParallelTrainer.step end to end: fwd + raster + MSE + bwd (+ overlapped all-reduce) + global-norm clip + AdamW + weight refresh (betas / eps / gradient_clip_val of diffusionGS_rel.yaml)"}


def scene_train_bench(a, dev, rank, world, steps, warmup):
    """BASELINE configs[4] as a training step in the mode(s) `--scene-recompute` names.  `both`: the reference's memory-saving mode
    (`use_checkpoint: true`, diffusionGS_scene_512.yaml:86-89 / denoiser.py:343-354: every block's forward runs twice) AND the mode an
    MI355X affords (every activation kept: 87 GiB at micro-batch 4 of 288), one after the other in this process; the returned object
    is the faster mode's, with both under `modes` (`frac_of_bf16_peak` of each on the FLOPs that mode requires: 4 x / 3 x the forward)."""
    import torch
    spec = dict(batch=a.scene_train_batch, views=4, rendered_views=7, res=512)
    if a.scene_recompute != "both":
        return train_bench(a, dev, rank, world, steps, warmup, scene=dict(spec, recompute={"auto": None, "on": True, "off": False}[a.scene_recompute]))
    runs = {}
    for name, rc in (("recompute_on", True), ("save_all", False)):
        if not DRY["on"]:
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats(dev)
        runs[name] = train_bench(a, dev, rank, world, steps, warmup, scene=dict(spec, recompute=rc))
    best = min(runs, key=lambda k: runs[k]["ms_per_step"])
    keep = ("ms_per_step", "samples_per_s", "recompute", "frac_of_bf16_peak", "dit_tflops_per_gpu", "saved_activation_gib", "gpu_memory_gib", "host_enqueue_ms_per_step", "loss")
    return {**runs[best], "headline_mode": best, "modes": {k: {f: v[f] for f in keep} for k, v in runs.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="infer", choices=["infer", "train", "train-scene"])
    ap.add_argument("--batch", type=int, default=1, help="samples per GPU per step (pipline_obj.py samples one object)")
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--views", type=int, default=4)
    ap.add_argument("--roofline-kernel", default="attention", choices=sorted(PROF_KINDS))
    ap.add_argument("--train-batch", type=int, default=4)
    ap.add_argument("--train-views", type=int, default=10)
    ap.add_argument("--train-steps", type=int, default=5)
    ap.add_argument("--scene-train-batch", type=int, default=12, help="samples per GPU of the scene-512 training step (diffusionGS_scene_512.yaml:16)")
    ap.add_argument("--scene-recompute", default="both", choices=["both", "auto", "on", "off"], help="scene-512 training step: per-block activation recompute -- "
                    "on = the reference's `use_checkpoint: true` (4 x the forward's FLOPs, 10 GiB of activations), off = save everything (3 x, 87 GiB at "
                    "micro-batch 4: what 288 GB of HBM are for), auto = the engine's policy (recompute only if the saved activations would not fit), "
                    "both (default) = on AND off, each reported, the faster one is the object's headline")
    ap.add_argument("--clip", type=float, default=0.5, help="training step: global-norm gradient clip (gradient_clip_val of configs/diffusionGS_rel.yaml:76-77); 0 = none")
    ap.add_argument("--bucket-mb", type=int, default=0, help="all-reduce bucket size; 0 = 32 MiB per rank (dgs_amd/parallel.py)")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"], help="training step: dgs_amd.optim.FusedAdamW (one launch: AdamW + the "
                    "engine's weight copies) or torch.optim.AdamW + the torch-copy refresh")
    ap.add_argument("--grad-exchange", default="fp32", choices=["fp32", "bf16"], help="dtype of the gradient all-reduce (bf16: half the xGMI bytes)")
    ap.add_argument("--dry-run-cpu", action="store_true", help="NOT a measurement: the same script on the CPU emulator build of the kernels with "
                    "gloo and a tiny model (width 256, 2 blocks, 64^2) -- what tests/test_bench_dry_run.py uses to exercise the N > 1 path")
    ap.add_argument("--force-dist", action="store_true", help="create the process group and issue every barrier / all-reduce even for a world of one "
                    "(the one-GPU RCCL smoke run: same init, streams and collective calls as N > 1)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("DGS_GRAPH", "1")), help="1 (default): the timed steps are replays of DGSDenoiser.forward captured as ONE hipGraph "
                    "(dgs_amd/graph.py), except the steps that carry the roofline kernel's HIP events (every 4th), which are enqueued eagerly; 0: every step eager")
    ap.add_argument("--preheat-s", type=float, default=1.0, help="seconds of the same step run (untimed) before the warm-up steps: a timed region of "
                    "~0.15 s that starts from idle clocks measures the DVFS ramp, not the kernels (BENCH_r03 vs the builder's runs: -8 %%)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the timed region (profiling runs)")
    ap.add_argument("--extras-timeout", type=float, default=420.0, help="seconds the informational objects may take before the line is printed without them")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn(a)
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.dry_run_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from emu_util import emu_lib                 # test infrastructure: the csrc/*.hip sources compiled for the CPU emulator
        DRY.update(on=True, lib=emu_lib())
        MODEL_CFG.update(width=256, num_layers=2)
        a.res, a.train_batch, a.train_views, a.no_extras = 64, min(a.train_batch, 2), min(a.train_views, 2), True
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    DIST["on"] = world > 1 or a.force_dist
    if DIST["on"]:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")          # only --force-dist without a launcher gets here without one
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.dry_run_cpu:
            dist.init_process_group("gloo")
        else:
            os.environ.setdefault("NCCL_DEBUG", "VERSION")    # RCCL prints its version line into stderr when its communicator comes up: the
            dist.init_process_group("nccl", device_id=dev)   # run's own evidence that the collectives below went through RCCL
        # every rank adds one: the number of ranks that really took part in THIS process group, said before anything is timed
        seen = torch.ones(1, dtype=torch.float32, device=dev if not a.dry_run_cpu else "cpu")
        dist.all_reduce(seen)
        if not a.dry_run_cpu:
            torch.cuda.synchronize()
        print(f"[bench] rank {rank}: process group up, backend {dist.get_backend()}, world_size {dist.get_world_size()}, ranks_seen {int(seen.item())}"
              f"{'' if int(seen.item()) == a.gpus else '  != --gpus ' + str(a.gpus)}", file=sys.stderr, flush=True)
        assert int(seen.item()) == world, f"all-reduce over the process group counted {int(seen.item())} ranks, WORLD_SIZE={world}"
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    dry_note = {"dry_run": "CPU emulator + gloo + tiny model: exercises launch / timing / reduction plumbing, NOT a measurement"} if a.dry_run_cpu else {}

    if a.mode == "train-scene":      # BASELINE configs[4] as the timed region (profiling runs; the default line carries it as `train_step_scene_512`)
        tb = scene_train_bench(a, dev, rank, world, a.steps, a.warmup)
        if rank == 0:
            print(json.dumps({
                "metric": "training samples/sec (DiT fwd+bwd + GS raster fwd+bwd + grad all-reduce + AdamW) at 512^2, scene model",
                "value": tb["samples_per_s"], "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": tb["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                "data": "synthetic", "config": {"workload": tb["workload"], "parallelism": f"dp{world}"}, "train_step_scene_512": tb, **dry_note}), flush=True)
        if DIST["on"]:
            torch.distributed.destroy_process_group()
        return
    if a.mode == "train":
        tb = train_bench(a, dev, rank, world, a.steps, a.warmup)
        if rank == 0:
            print(json.dumps({
                "metric": f"training samples/sec (DiT fwd+bwd + GS raster fwd+bwd + grad all-reduce + AdamW) at {a.res}^2",
                "value": tb["samples_per_s"], "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": tb["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
                "data": "synthetic", "config": {"workload": f"obj-{a.res} training step (BASELINE.json configs[3]): B={a.train_batch} samples/GPU, "
                                                            f"4 input views, {a.train_views} rendered views, 460 M parameters, random init",
                                                "parallelism": f"dp{world}"}, "train_step": tb, **dry_note}), flush=True)
        if DIST["on"]:
            torch.distributed.destroy_process_group()
        return

    from dgs_amd import denoiser as dn, synth
    model = dn.DGSDenoiser(MODEL_CFG, device=dev, lib=DRY["lib"])
    model.reset_parameters(seed=0)          # every rank the same random-init weights (pure data parallel inference)
    B, V, res = a.batch, a.views, a.res
    batch, t = synth.make_batch(B, res, V=V, device=dev, seed=rank, with_t=True)
    eng = model.engine()
    L = eng.num_tokens(V, res, res)

    def step(prof=None):
        # == DGSDenoiser.forward (denoiser.py:284-287); the profiling hook only adds event records on the stream
        with torch.no_grad():
            params, _ = eng.image_to_gaussians(batch["image"], batch["ray_o"], batch["ray_d"], t, prof=prof)
            pc = params.pop("prof_count", 0)
            p = dn.AttrDict(params)
            rendered = model.render_gaussians(p, batch["c2w"], batch["fxfycxcy"], res, res)
            return rendered, model.prepare_to_save(p), pc

    def barrier():
        if DIST["on"]:
            torch.distributed.barrier()

    from dgs_amd.dit import DitOps
    clock = {}
    if not a.dry_run_cpu:
        clock["at_start_mhz"] = round(DitOps().shader_clock_mhz(dev), 0)      # before any load: the idle / ramping state

    use_graph = bool(a.graph) and not a.dry_run_cpu
    graph_step = None
    graph_error = None
    if use_graph:
        # DGSDenoiser.forward as one captured graph: same kernels (the capture goes through the same C calls), one host call per step
        gb = {k: batch[k] for k in ("image", "ray_o", "ray_d", "c2w", "fxfycxcy")}
        try:
            graphed = model.graphed(gb, t)

            def graph_step():
                rendered, gaussians = graphed.replay()
                return rendered, gaussians, 0
        except Exception as e:                      # noqa: BLE001 -- the eager step is the same work: measure that, and say why
            graph_error, use_graph = f"{type(e).__name__}: {str(e)[:300]}", False
            model.drop_graphs()

    def timed_loop(n, fn):
        """n steps of fn: (wall ms per step incl. the final synchronisation, host ms per step until the last step was ENQUEUED, GPU ms
        per step between two events on the stream)."""
        e0, e1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if not a.dry_run_cpu else (None, None)
        _sync()
        w0 = time.perf_counter()
        if e0 is not None:
            e0.record()
        for _ in range(n):
            fn()
        if e1 is not None:
            e1.record()
        w1 = time.perf_counter()
        _sync()
        w2 = time.perf_counter()
        return (w2 - w0) / n * 1e3, (w1 - w0) / n * 1e3, (e0.elapsed_time(e1) / n if e0 is not None else None)

    # pre-heat (untimed): the same step until the clocks have settled
    if not a.dry_run_cpu and a.preheat_s > 0:
        h0 = time.perf_counter()
        while time.perf_counter() - h0 < a.preheat_s:
            for _ in range(10):
                (graph_step or step)()
            _sync()
        clock["after_preheat_mhz"] = round(DitOps().shader_clock_mhz(dev), 0)

    loop_ms = None
    if not a.no_extras:
        # Informational: the reference's 30-step sampling loop end to end -- DGSDenoiser.forward + the device
        # sampler step (dgs_amd/sampler.py) per iteration; one untimed loop, one timed.  Not part of `value`.
        from dgs_amd import sampler as sm
        diffusion = sm.create_diffusion("30", device=dev)
        loop_batch = dict(batch)
        for timed in (False, True):
            loop_batch["image"] = batch["image"].clone()
            loop_batch["image_noisy"] = torch.randn_like(batch["image"][:, 1:])
            _sync()
            l0 = time.perf_counter()
            with torch.no_grad():
                diffusion.p_sample_loop(model, loop_batch, use_graph=use_graph)
            _sync()
            if timed:
                loop_ms = (time.perf_counter() - l0) * 1e3

    variants = None
    if not a.no_extras and not a.dry_run_cpu:
        # the step both ways, 20 steps each, untimed region: what the graph buys on this box, and how far the host runs ahead
        variants = {}
        for name, fn in (("eager", step), ("graph", graph_step)):
            if fn is None:
                continue
            for _ in range(3):
                fn()
            wall, host, gpu = timed_loop(20, fn)
            variants[name] = {"ms_per_step": round(wall, 3), "host_enqueue_ms_per_step": round(host, 3), "gpu_ms_per_step": round(gpu, 3)}

    families = None
    if not a.no_extras and not a.dry_run_cpu and B == 1:
        # every kernel family of the block on its own roofline (informational; two eager steps per family with HIP events around its
        # launches, outside the timed region): MFMA for attention and the K-heavy GEMMs, HBM for LayerNorm; the gate-residual family
        # (proj + fc2) moves the fp32 residual stream and is priced on both
        nlay = MODEL_CFG["num_layers"]
        Wd = MODEL_CFG["width"]
        fam = {"attention": (nlay, 4.0 * L * L * Wd, None), "gemm_qkv": (nlay, 2.0 * L * 3 * Wd * Wd, 2 * L * Wd + 6 * Wd * Wd + 6 * L * Wd),
               "gemm_fc1_gelu": (nlay, 2.0 * L * 4 * Wd * Wd, 2 * L * Wd + 8 * Wd * Wd + 8 * L * Wd),
               # the two gated-residual GEMMs each on its own (round 5 averaged them into one family, which hid the worse one): bf16 A [L, K] +
               # bf16 W [W, K] in, the fp32 residual stream [L, W] read and written
               "gemm_proj": (nlay, 2.0 * L * Wd * Wd, 2 * L * Wd + 2 * Wd * Wd + 8 * L * Wd),
               "gemm_fc2": (nlay, 2.0 * L * 4 * Wd * Wd, 8 * L * Wd + 8 * Wd * Wd + 8 * L * Wd),
               "layernorm": (2 * nlay, None, 6 * L * Wd)}
        families = {}
        for kind, (per, flops, nbytes) in fam.items():
            us = []
            for _ in range(2):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * per)]
                for e in ev:
                    e.record()
                step(prof=(PROF_KINDS[kind], ev))
                torch.cuda.synchronize()
                us += [ev[2 * j].elapsed_time(ev[2 * j + 1]) * 1e3 for j in range(per)]
            avg = float(np.mean(us))
            rec = {"launches_per_step": per, "avg_launch_us": round(avg, 2)}
            if flops:
                rec.update({"tflops": round(flops / avg / 1e6, 1), "frac_of_bf16_mfma_peak": round(flops / (avg * 1e-6) / PEAK_BF16_MFMA, 4)})
            if nbytes:
                rec.update({"algorithmic_gbps": round(nbytes / avg / 1e3, 1), "frac_of_hbm_peak": round(nbytes / (avg * 1e-6) / PEAK_HBM, 4)})
            families[kind] = rec

    run_step = graph_step or step
    if use_graph:
        graphed(gb, t)          # the sampling loop above went through the same graph: its static tensors hold the loop's last inputs
    for _ in range(a.warmup):
        run_step()
    nl = MODEL_CFG["num_layers"]
    per_step = {"attention": nl, "gemm_qkv": nl, "gemm_gate_residual": 2 * nl, "gemm_fc1_gelu": nl, "layernorm": 2 * nl, "gemm_proj": nl, "gemm_fc2": nl}[a.roofline_kernel]
    # HIP events around every launch of the roofline kernel on every 4th step of the timed region: an event record is a packet
    # of its own between two kernels (~2 us), 48 of them per step were 1.5 % of the step they measure.  Those steps are enqueued
    # eagerly (events cannot be re-armed inside a captured graph); the others are graph replays
    prof_steps = [] if a.dry_run_cpu else [i for i in range(a.steps) if i % 4 == 0]
    events = {i: [torch.cuda.Event(enable_timing=True) for _ in range(2 * per_step)] for i in prof_steps}
    for ev in events.values():  # materialise the HIP event handles before the timed region
        for e in ev:
            e.record()
    g0, g1 = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if not a.dry_run_cpu else (None, None)
    barrier(); _sync()
    t0 = time.perf_counter()
    if g0 is not None:
        g0.record()
    for i in range(a.steps):
        if i in events:
            rendered, gaussians, _pc = step(prof=(PROF_KINDS[a.roofline_kernel], events[i]))
        else:
            rendered, gaussians, _pc = run_step()
    if g1 is not None:
        g1.record()
    t_enq = time.perf_counter()
    _sync(); barrier()
    elapsed = time.perf_counter() - t0
    local_ms = elapsed / a.steps * 1e3
    # the timed steps produced what they claim: no deferred device-side failure (asynchronous renders report one call late), finite images
    if not a.dry_run_cpu:
        if use_graph:
            graphed.check(wait=True)
        model.gs_renderer.backend().check_async(wait=True)
        assert bool(torch.isfinite(rendered).all()), "the timed region rendered non-finite images"
        # the last step's outputs are the captured graph's own tensors: keep copies for the PSNR check behind the informational objects
        # (the graph is dropped there to free its memory)
        rendered = rendered.clone()
        g0_ = gaussians[0]
        gaussians = [types.SimpleNamespace(_xyz=g0_._xyz.clone(), _features_dc=g0_._features_dc.clone(), _scaling=g0_._scaling.clone(),
                                           _rotation=g0_._rotation.clone(), _opacity=g0_._opacity.clone())]
    if not a.dry_run_cpu:
        clock["after_timed_region_mhz"] = round(DitOps().shader_clock_mhz(dev), 0)
    rank_ms = [local_ms]
    if DIST["on"]:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        torch.distributed.all_gather(allt, tt)
        rank_ms = [float(x.item()) / a.steps * 1e3 for x in allt]
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    out = None
    if rank == 0:
        ms = elapsed / a.steps * 1e3
        value = B * V * world / (elapsed / a.steps)
        kern_ms = [events[i][2 * j].elapsed_time(events[i][2 * j + 1]) for i in prof_steps for j in range(per_step)]
        avg_s = float(np.mean(kern_ms)) * 1e-3 if kern_ms else float("inf")
        achieved = kernel_flops(a.roofline_kernel, L, B) / avg_s / 1e12
        traffic = pmc_traffic("dit")
        tr_bytes = None
        if traffic and B == 1 and res == 256 and V == 4:
            tr_bytes = traffic.get("dit", {}).get(a.roofline_kernel, {}).get("traffic_bytes_per_launch")
        out = {
            "metric": "novel-view renders/sec (DiT step + GS raster) at 256^2", "value": round(value, 2), "unit": "renders/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"obj-{res} sampling step (BASELINE.json configs[2]): DGSDenoiser.forward = DiT "
                                   f"image_to_gaussians (width {MODEL_CFG['width']}, {nl} blocks, L={L}, bf16 MFMA, fp32 accumulate) + {V} fp32 "
                                   f"rasterizations of P={2 + V * res * res} Gaussians at {res}^2 per sample; random-init weights",
                       "batch_per_gpu": B, "views": V, "resolution": res, "tokens": L, "gaussians": 2 + V * res * res,
                       "dit_tflop_per_sample": round(dit_flops(L, MODEL_CFG["width"], nl) / 1e12, 3), "parallelism": f"dp{world}"},
            "roofline": {"kernel": a.roofline_kernel, "bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_MFMA / 1e12,
                         "unit": "TFLOP/s", "frac": round(achieved * 1e12 / PEAK_BF16_MFMA, 4), "traffic": tr_bytes,
                         "launches_timed": len(kern_ms), "avg_launch_us": round(avg_s * 1e6, 2),
                         "step_frac_of_peak": round(dit_flops(L, MODEL_CFG["width"], nl) * B / (ms * 1e-3) / PEAK_BF16_MFMA, 4)},
            # what the timed region looked like from the host and from the device (rank 0): `host_enqueue_ms` = until the last step
            # was enqueued (the host runs ahead of the device by ms_per_step - this), `gpu_ms` = between two events on the stream,
            # `shader_clock_mhz` = s_memtime / s_memrealtime of a probe kernel at three moments, `per_rank_ms` = every rank's own time
            "timed_region": {"graph_replays": sum(1 for i in range(a.steps) if i not in events) if use_graph else 0, "graph_error": graph_error,
                             "eager_steps_with_events": len(events), "preheat_s": a.preheat_s if not a.dry_run_cpu else 0.0,
                             "host_enqueue_ms_per_step": round((t_enq - t0) / a.steps * 1e3, 3),
                             "gpu_ms_per_step": round(g0.elapsed_time(g1) / a.steps, 3) if g0 is not None else None,
                             "shader_clock_mhz": clock, "per_rank_ms": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3)},
                             "ranks_seen": len(rank_ms)},
            **dry_note,
        }
        if variants:
            out["step_variants"] = variants
        if families:
            out["kernel_families"] = families
        if loop_ms is not None:
            out["sampling_loop_30_steps"] = {"ms_per_loop": round(loop_ms, 2), "renders_per_s": round(B * V * 30 / (loop_ms * 1e-3), 1),
                                             "note": "per GPU; informational, not part of value"}
    # Everything below is informational.  The line that counts is complete at this point; if an informational object fails or
    # stalls (the training step is the only place a collective runs), the line still goes out, with the reason, and the job ends.
    def emit_and_leave(reason):
        if rank == 0:
            out["extras_error"] = reason
            out.setdefault("cpu_baseline", None)
            print(json.dumps(out), flush=True)
        os._exit(0)

    watchdog = threading.Timer(a.extras_timeout, emit_and_leave, args=(f"informational objects did not finish within {a.extras_timeout:.0f} s",))
    watchdog.daemon = True
    watchdog.start()
    x0 = time.perf_counter()
    stage = lambda name: print(f"[bench] {name} ({time.perf_counter() - x0:.1f} s)", file=sys.stderr, flush=True) if rank == 0 else None
    if not a.no_extras:
        try:
            stage("extras: raster roofline")
            rr = raster_roofline(dev, res, V) if rank == 0 else None
            stage("extras: batch sweep")
            sweep = batch_sweep(dev, model, res, V) if rank == 0 and not DRY["on"] else None
            stage("extras: scene 512")
            s512 = scene_512(dev) if rank == 0 else None
            del model, eng
            graph_step = graphed = run_step = None      # the captured graph holds the model's buffers
            torch.cuda.empty_cache()
            stage("extras: training step")
            tb = train_bench(a, dev, rank, world, a.train_steps, 2)      # every rank: the step has a collective
            tb512 = None
            if a.res == 256 and not DRY["on"]:
                stage("extras: scene-512 training step")
                torch.cuda.empty_cache()
                torch.cuda.reset_peak_memory_stats(dev)
                tb512 = scene_train_bench(a, dev, rank, world, 2, 1)
            stage("extras: done")
        except Exception as e:                                          # noqa: BLE001 -- whatever it was, the line goes out
            emit_and_leave(f"{type(e).__name__}: {e}")
        if rank == 0:
            out["raster"] = rr
            out["batch_sweep"] = sweep
            out["scene_512"] = s512
            out["train_step"] = tb
            out["train_step_scene_512"] = tb512
            out["extras_s"] = round(time.perf_counter() - x0, 1)       # wall time of the informational objects above
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline and not a.no_extras:
            model = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=24, ray_pe_type="relative_plk"), device=dev)
            model.reset_parameters(seed=0)
            final = {k: getattr(gaussians[0], "_" + n)[None] for k, n in
                     (("xyz", "xyz"), ("features", "features_dc"), ("scaling", "scaling"), ("rotation", "rotation"), ("opacity", "opacity"))}
            base, psnr = cpu_baseline(model, batch, t, res, V, final, rendered)
            out["cpu_baseline"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in base.items()}
            out["psnr_vs_oracle_db"] = round(psnr, 2)
        else:
            out["cpu_baseline"] = None
        watchdog.cancel()
        print(json.dumps(out), flush=True)
    watchdog.cancel()
    if DIST["on"]:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
