"""Shared helpers for the DiT parity tests: seeded synthetic inputs + error metrics."""
import numpy as np
import torch

from dgs_amd import cameras
from oracle import dit_oracle as D


def bf16_round_state_dict(sd):
    """The HIP path keeps GEMM weights in bf16; give the fp32 oracle the SAME (bf16-representable) weights so the parity
    tolerance only has to cover activation rounding, not weight rounding."""
    out = {}
    for k, v in sd.items():
        is_gemm_w = k.endswith("weight") and v.dim() == 2
        out[k] = v.to(torch.bfloat16).float() if is_gemm_w else v.clone()
    return out


def synth_inputs(cfg, B, V, res, seed=0):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(B, V, 3, res, res, generator=g)
    c2w = torch.tensor(np.stack([cameras.ring_cameras(V, phase_deg=17.0 * b) for b in range(B)], 0))
    k = torch.tensor(cameras.default_fxfycxcy(res)).expand(B, V, 4).contiguous()
    ray_o, ray_d = D.transform_input_rays(c2w, k, res, res)
    t = torch.randint(0, 1000, (B,), generator=g)
    return images, ray_o.contiguous(), ray_d.contiguous(), t, c2w, k


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def golden_case(kind, tag="hip256"):
    """(cfg, state_dict, inputs, reference outputs) of tests/golden/dit_golden_hip256.npz for kind in {'obj','scene'}."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"dit_golden_{tag}.npz"))
    cfg = D.Cfg(width=int(z["width"]), num_layers=int(z["layers"]), scene=(kind == "scene"), range_far=50.0,
                ray_pe_type="plk" if kind == "scene" else "relative_plk",
                gaussians_sh_degree=int(z["sh_degree"]) if "sh_degree" in z.files else 0)
    sd = D.parity_state_dict(cfg, int(z["seed"]))
    t = lambda k: torch.tensor(z[kind + "_" + k])
    res = int(z["res"])
    ray_o, ray_d = D.transform_input_rays(t("in_c2w"), t("in_fxfycxcy"), res, res)
    inp = dict(images=t("in_images"), ray_o=ray_o.contiguous(), ray_d=ray_d.contiguous(), t=t("in_t"))
    ref = {k: t("out_" + k) for k in ("xyz", "features", "scaling", "rotation", "opacity", "aligned")}
    return cfg, sd, inp, ref


FIELDS = ("xyz", "features", "scaling", "rotation", "opacity")


def oracle_gradients(sd, cfg, images, ray_o, ray_d, t, wts, dev, checkpoint_blocks=False):
    """Parameter gradients of sum_k <out_k, wts_k> by torch autograd through the fp32 oracle evaluated on `dev`, ONE SAMPLE AT A
    TIME (at L = 4098 the oracle's attention matrices are ~26 GB per sample) and summed over samples in fp64.  checkpoint_blocks:
    every block under torch.utils.checkpoint (L = 16,386: 24 blocks x 2 score matrices of 17 GB would not fit; one block's do).
    Returns (outputs per field [B, ...] fp32, {state-dict key: gradient fp64})."""
    B = images.shape[0]
    leaf = {k: v.to(dev).requires_grad_(True) for k, v in sd.items()}
    total = {k: torch.zeros(v.shape, dtype=torch.float64, device=dev) for k, v in leaf.items()}
    outs = {k: [] for k in FIELDS}
    for b in range(B):
        s = slice(b, b + 1)
        ref, _ = D.image_to_gaussians(leaf, cfg, images[s].to(dev), ray_o[s].to(dev), ray_d[s].to(dev), t[s].to(dev), checkpoint_blocks=checkpoint_blocks)
        sum((ref[k] * wts[k][s]).sum() for k in FIELDS).backward()
        for k, v in leaf.items():
            if v.grad is not None:
                total[k] += v.grad.double()
                v.grad = None
        for k in FIELDS:
            outs[k].append(ref[k].detach())
        del ref
    return {k: torch.cat(v, 0) for k, v in outs.items()}, total


def gradient_errors(grads, ref):
    """Per tensor: rel-L2, max |diff| / max |ref|, and for 2-D tensors the largest ROW error norm relative to the largest row norm
    (a single wrong output feature / bias row cannot hide in the tensor norm; relative to the row's OWN norm the measure is noise
    for rows of the adaLN weight gradients, which are one scalar dmod[n] times a common vector)."""
    out = {}
    for k, gv in grads.items():
        r = ref[k].double()
        g = gv.reshape(r.shape).double()
        d = g - r
        rec = {"rel_l2": float(d.norm() / r.norm().clamp_min(1e-30)), "max_abs": float(d.abs().max() / r.abs().max().clamp_min(1e-30))}
        if r.dim() == 2 and r.shape[0] > 1:
            rec["worst_row"] = float(d.norm(dim=1).max() / r.norm(dim=1).max().clamp_min(1e-30))
        out[k] = rec
    return out
