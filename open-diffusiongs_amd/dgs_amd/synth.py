"""Seeded synthetic inputs of the shapes BASELINE.json / SURVEY.md section 8(d) name.

"random-Gaussian" scenes are DiffusionGS-shaped: one Gaussian per pixel of each of the
`v_in` input views, placed on that pixel's ray (mirrors the hard pixel-alignment of
/root/reference/diffusionGS/models/denoiser/denoiser.py:382-392), plus `n_free` free
Gaussians; activations follow gs_core.py:330-334 (exp / normalize / sigmoid) and
denoiser.py:118-119 (scale bias -2.3 clamp -1.2, opacity bias -2.0).
"""
import torch
import numpy as np

from . import cameras

REGIMES = {"init": -2.3, "trained": -4.5, "small": -5.5}


def gaussian_scene(res, v_in=4, n_free=2, regime="trained", seed=0, sh_degree=0, activated=True):
    """Returns dict of float32 arrays: xyz[P,3], shs[P,M,3], scales[P,3], rotations[P,4],
    opacities[P,1] (activated unless activated=False -> raw pre-activation values)."""
    rng = np.random.default_rng(seed)
    mu = REGIMES[regime] if isinstance(regime, str) else float(regime)
    c2ws = cameras.ring_cameras(v_in)
    fxfycxcy = cameras.default_fxfycxcy(res)
    pts = [rng.normal(0.0, 0.1, size=(n_free, 3)).astype(np.float32)]
    for v in range(v_in):
        o, d = cameras.pixel_rays(c2ws[v], fxfycxcy, res, res)
        u = rng.uniform(0.3, 0.7, size=(res, res, 1)).astype(np.float32)
        o_dot_d = np.sum(-o * d, axis=-1, keepdims=True)
        t = (2.0 * u - 1.0) * np.float32(1.8) + o_dot_d
        pts.append((o + t * d).reshape(-1, 3).astype(np.float32))
    xyz = np.concatenate(pts, 0)
    P = xyz.shape[0]
    M = (sh_degree + 1) ** 2
    raw_scale = np.minimum(rng.normal(mu, 0.5, size=(P, 3)), -1.2).astype(np.float32)
    raw_op = (rng.normal(0.0, 1.0, size=(P, 1)) - 2.0).astype(np.float32)
    raw_rot = rng.normal(0.0, 1.0, size=(P, 4)).astype(np.float32)
    shs = rng.uniform(-1.77, 1.77, size=(P, M, 3)).astype(np.float32)
    if M > 1:
        shs[:, 1:, :] *= 0.25
    if activated:
        scales = np.exp(raw_scale).astype(np.float32)
        rot = (raw_rot / np.maximum(np.linalg.norm(raw_rot, axis=1, keepdims=True), 1e-12)).astype(np.float32)
        op = (1.0 / (1.0 + np.exp(-raw_op.astype(np.float64)))).astype(np.float32)
    else:
        scales, rot, op = raw_scale, raw_rot, raw_op
    return dict(xyz=xyz, shs=shs, scales=scales, rotations=rot, opacities=op)


def render_cameras(res, n_views, phase_deg=0.0, res_h=None):
    """List of camera dicts (cameras.camera_from_c2w) for `n_views` ring poses."""
    res_h = res if res_h is None else res_h
    c2ws = cameras.ring_cameras(n_views, phase_deg=phase_deg)
    fxfycxcy = cameras.default_fxfycxcy(res, res_h)
    return [cameras.camera_from_c2w(c2ws[k], fxfycxcy, res_h, res) for k in range(n_views)], c2ws, fxfycxcy


def make_batch(B, res, V=4, device="cuda", seed=0, with_t=False):
    """Synthetic inputs of the reference's shapes: U[0,1) images, ring cameras radius 3, G-Objaverse
    intrinsics; rays as TransformInput (systems/utils.py:621-757) computes them -- upstream of the timed step."""
    from . import cameras
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(B, V, 3, res, res, generator=g)
    c2w = np.stack([cameras.ring_cameras(V, phase_deg=13.0 * b + seed) for b in range(B)], 0)
    k = np.broadcast_to(cameras.default_fxfycxcy(res), (B, V, 4)).copy()
    rays = [[cameras.pixel_rays(c2w[b, v], k[b, v], res, res) for v in range(V)] for b in range(B)]
    ray_o = torch.tensor(np.stack([[r[0] for r in row] for row in rays])).permute(0, 1, 4, 2, 3).contiguous()
    ray_d = torch.tensor(np.stack([[r[1] for r in row] for row in rays])).permute(0, 1, 4, 2, 3).contiguous()
    t = torch.randint(0, 1000, (B,), generator=g)
    batch = dict(image=images, ray_o=ray_o, ray_d=ray_d, c2w=torch.tensor(c2w), fxfycxcy=torch.tensor(k))
    batch = {a: b.to(device) for a, b in batch.items()}
    return (batch, t.to(device)) if with_t else batch


