"""Small seeded scenes for oracle / parity tests (numpy float32)."""
import numpy as np

from dgs_amd import cameras


def small_scene(P, res_w, res_h=None, seed=0, sh_degree=0, spread=0.6, log_scale=-2.6, n_views=1, phase=25.0):
    res_h = res_w if res_h is None else res_h
    rng = np.random.default_rng(seed)
    M = (sh_degree + 1) ** 2
    xyz = rng.uniform(-spread, spread, size=(P, 3)).astype(np.float32)
    scales = np.exp(rng.normal(log_scale, 0.4, size=(P, 3))).astype(np.float32)
    q = rng.normal(size=(P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    op = (1.0 / (1.0 + np.exp(-rng.normal(0.5, 1.0, size=(P, 1))))).astype(np.float32)
    shs = rng.uniform(-1.5, 1.5, size=(P, M, 3)).astype(np.float32)
    if M > 1:
        shs[:, 1:] *= 0.3
    c2ws = cameras.ring_cameras(n_views, phase_deg=phase)
    fxfycxcy = cameras.default_fxfycxcy(res_w, res_h)
    cams = [cameras.camera_from_c2w(c2ws[k], fxfycxcy, res_h, res_w) for k in range(n_views)]
    return dict(xyz=xyz, shs=shs, scales=scales, rotations=q.astype(np.float32), opacities=op), cams


def oracle_forward(o, sc, cam, H, W, bg=(1.0, 1.0, 1.0), sh_degree=0, exp_mode=0, **kw):
    args = dict(shs=sc.get("shs"), scales=sc.get("scales"), rotations=sc.get("rotations"))
    args.update(kw)
    return o.forward(np.asarray(bg, np.float32), sc["xyz"], sc["opacities"], cam["viewmatrix"], cam["projmatrix"],
                     cam["campos"], cam["tanfovx"], cam["tanfovy"], H, W, sh_degree=sh_degree,
                     exp_mode=exp_mode, **args)
