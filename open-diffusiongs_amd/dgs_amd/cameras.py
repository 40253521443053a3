"""Camera math shared by tests / bench (numpy, float32).

Follows /root/reference/diffusionGS/models/gsrenderer/gs_core.py:277-316 (`Camera`): OpenCV
camera-to-world + pixel intrinsics -> the transposed view / full-projection matrices the
rasterizer consumes (the kernels index them column-major, auxiliary.h:58-77).
"""
import numpy as np

ZNEAR = 0.01
ZFAR = 100.0


def look_at_c2w(cam_pos, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """OpenCV-convention (+x right, +y down, +z forward) camera-to-world looking at `target`."""
    c = np.asarray(cam_pos, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - c
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, dtype=np.float64))
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = r, d, f, c
    return m.astype(np.float32)


def ring_cameras(n_views, radius=3.0, elevation_deg=11.0, phase_deg=0.0):
    """`n_views` poses on a ring looking at the origin."""
    out = []
    el = np.deg2rad(elevation_deg)
    for k in range(n_views):
        az = np.deg2rad(phase_deg) + 2.0 * np.pi * k / n_views
        pos = radius * np.array([np.cos(el) * np.cos(az), np.cos(el) * np.sin(az), np.sin(el)])
        out.append(look_at_c2w(pos))
    return np.stack(out, 0)


def default_fxfycxcy(res_w, res_h=None):
    """G-Objaverse intrinsics, /root/reference/diffusionGS/data/base.py:55,233-235."""
    res_h = res_w if res_h is None else res_h
    f = np.float32(1422.222 / 1024)
    return np.array([f * res_w, f * res_h, 0.5 * res_w, 0.5 * res_h], dtype=np.float32)


def camera_from_c2w(c2w, fxfycxcy, h, w):
    """Returns dict(viewmatrix[4,4], projmatrix[4,4], campos[3], tanfovx, tanfovy) in float32.

    viewmatrix = W2C^T, projmatrix = W2C^T @ P^T  (gs_core.py:307-315)."""
    c2w = np.asarray(c2w, dtype=np.float32)
    fx, fy, cx, cy = [np.float32(v) for v in fxfycxcy]
    w2c = np.linalg.inv(c2w.astype(np.float64)).astype(np.float32)
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 2 * fx / w
    P[1, 1] = 2 * fy / h
    P[0, 2] = 2 * (cx / w) - 1
    P[1, 2] = 2 * (cy / h) - 1
    P[2, 2] = -(ZFAR + ZNEAR) / (ZFAR - ZNEAR)
    P[3, 2] = 1.0
    P[2, 3] = -(2 * ZFAR * ZNEAR) / (ZFAR - ZNEAR)
    view = np.ascontiguousarray(w2c.T)
    proj = np.ascontiguousarray((view @ P.T).astype(np.float32))
    return dict(viewmatrix=view, projmatrix=proj, campos=np.ascontiguousarray(c2w[:3, 3]),
                tanfovx=float(np.float32(w) / (2 * fx)), tanfovy=float(np.float32(h) / (2 * fy)))


def pixel_rays(c2w, fxfycxcy, h, w):
    """Per-pixel ray origin / unit direction in world space, [h,w,3] each.

    Follows TransformInput, /root/reference/diffusionGS/systems/utils.py:636-684 (pixel centres
    at +0.5)."""
    fx, fy, cx, cy = [np.float32(v) for v in fxfycxcy]
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    x = (xs + np.float32(0.5) - cx) / fx
    y = (ys + np.float32(0.5) - cy) / fy
    d = np.stack([x, y, np.ones_like(x)], -1)
    d = d @ np.asarray(c2w, np.float32)[:3, :3].T
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    o = np.broadcast_to(np.asarray(c2w, np.float32)[:3, 3], d.shape)
    return o.astype(np.float32), d.astype(np.float32)
