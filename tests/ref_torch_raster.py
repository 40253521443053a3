"""Independent float64 PyTorch restatement of the rasterizer's *math* (not its schedule).

Used only to pin the CPU oracle: forward colours and, through autograd, all gradients.
Dense O(P * H * W); keep P and the image tiny.  Follows the numerical spec of
cuda_rasterizer/forward.cu:74-113,118-152,155-256,261-374.
"""
import math

import torch

SH0 = 0.28209479177387814
SH1 = 0.4886025119029199
SH2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
       -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


def _sh_rgb(deg, xyz, campos, shs):
    d = xyz - campos[None]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = SH0 * shs[:, 0]
    if deg > 0:
        res = res - SH1 * y * shs[:, 1] + SH1 * z * shs[:, 2] - SH1 * x * shs[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        res = (res + SH2[0] * xy * shs[:, 4] + SH2[1] * yz * shs[:, 5] + SH2[2] * (2 * zz - xx - yy) * shs[:, 6]
               + SH2[3] * xz * shs[:, 7] + SH2[4] * (xx - yy) * shs[:, 8])
    if deg > 2:
        res = (res + SH3[0] * y * (3 * xx - yy) * shs[:, 9] + SH3[1] * xy * z * shs[:, 10]
               + SH3[2] * y * (4 * zz - xx - yy) * shs[:, 11] + SH3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
               + SH3[4] * x * (4 * zz - xx - yy) * shs[:, 13] + SH3[5] * z * (xx - yy) * shs[:, 14]
               + SH3[6] * x * (xx - 3 * yy) * shs[:, 15])
    res = res + 0.5
    return torch.clamp(res, min=0.0)


def render(xyz, shs, scales, rotations, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W,
           bg, sh_degree=0, scale_modifier=1.0, colors_precomp=None, cov3D_precomp=None):
    """All tensor args float64 (requires_grad allowed).  Returns dict(color[3,H,W], radii, n_contrib, final_T)."""
    dt = torch.float64
    P = xyz.shape[0]
    W2C = viewmatrix.t()            # viewmatrix tensor = W2C^T
    PW = projmatrix.t()             # projmatrix tensor = (P @ W2C)^T
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([xyz, ones], 1) @ PW.t()
    pv = torch.cat([xyz, ones], 1) @ W2C.t()
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :3] * pw[:, None]
    depth = pv[:, 2]
    fx = W / (2.0 * tanfovx)
    fy = H / (2.0 * tanfovy)
    if cov3D_precomp is None:
        r, x, y, z = rotations[:, 0], rotations[:, 1], rotations[:, 2], rotations[:, 3]
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
        S = torch.diag_embed(scale_modifier * scales)
        Sig = R @ S @ S @ R.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sig = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(P, 3, 3)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = pv[:, 2]
    tx = torch.clamp(pv[:, 0] / tz, -limx, limx) * tz
    ty = torch.clamp(pv[:, 1] / tz, -limy, limy) * tz
    zero = torch.zeros_like(tz)
    Jac = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).reshape(P, 2, 3)
    Rw = W2C[:3, :3]
    A = Jac @ Rw[None]
    cov2 = A @ Sig @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c_ = cov2[:, 1, 1] + 0.3
    det = a * c_ - b * b
    conic = torch.stack([c_ / det, -b / det, a / det], 1)
    mid = 0.5 * (a + c_)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam))
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + 15) // 16, (H + 15) // 16
    with torch.no_grad():
        rad = radius.detach()
        x0 = torch.clamp(torch.trunc((px.detach() - rad) / 16), 0, gx).long()
        y0 = torch.clamp(torch.trunc((py.detach() - rad) / 16), 0, gy).long()
        x1 = torch.clamp(torch.trunc((px.detach() + rad + 15) / 16), 0, gx).long()
        y1 = torch.clamp(torch.trunc((py.detach() + rad + 15) / 16), 0, gy).long()
        visible = (depth.detach() > 0.2) & (det.detach() != 0) & ((x1 - x0) * (y1 - y0) > 0)
        radii = torch.where(visible, rad.long(), torch.zeros_like(x0))
        order = sorted(range(P), key=lambda i: (float(depth[i].detach().float()), i))
    if colors_precomp is None:
        rgb = _sh_rgb(sh_degree, xyz, campos, shs)
    else:
        rgb = colors_precomp
    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    tyy, txx = (ys / 16).floor().long(), (xs / 16).floor().long()
    T = torch.ones(H, W, dtype=dt)
    C = torch.zeros(3, H, W, dtype=dt)
    done = torch.zeros(H, W, dtype=torch.bool)
    contributor = torch.zeros(H, W, dtype=torch.long)
    last = torch.zeros(H, W, dtype=torch.long)
    for g in order:
        if not bool(visible[g]):
            continue
        in_tile = (txx >= x0[g]) & (txx < x1[g]) & (tyy >= y0[g]) & (tyy < y1[g])
        live = in_tile & ~done
        contributor = contributor + live.long()
        dx, dy = px[g] - xs, py[g] - ys
        power = -0.5 * (conic[g, 0] * dx * dx + conic[g, 2] * dy * dy) - conic[g, 1] * dx * dy
        alpha = torch.clamp(opacities.reshape(-1)[g] * torch.exp(power), max=0.99)
        m = live & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = T * (1 - alpha)
        newly_done = m & (test_T < 1e-4)
        done = done | newly_done
        m = m & ~newly_done
        C = C + torch.where(m[None], rgb[g][:, None, None] * (alpha * T)[None], torch.zeros_like(C))
        T = torch.where(m, test_T, T)
        last = torch.where(m, contributor, last)
    color = C + T[None] * bg[:, None, None]
    return dict(color=color, radii=radii, n_contrib=last, final_T=T)
