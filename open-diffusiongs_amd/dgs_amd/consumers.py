"""Forward-only consumers of the rasterizer: the turntable render and the 3DGS .ply wire format.

Mirrors diffusionGS/models/gsrenderer/gs_core.py:
    get_turntable_cameras                      :50-85
    render_turntable                           :1203-1219   (reference: one rasterizer call + ~15 camera ops + syncs PER VIEW;
                                                             here all views are one batched launch sequence of the HIP rasterizer)
    render_generic                             :1300-1316   (same, for caller-supplied cameras; [v, h, w, 3] uint8)
    GaussianModel.construct_dtypes / save_ply / load_ply   :578-760   (the layout 3DGS viewers read: x y z, red green blue,
                                                             f_dc_*, f_rest_* padded to SH degree 3, opacity, scale_*, rot_*)
The reference writes / reads the file through the `plyfile` package (binary_little_endian 1.0, one "vertex" element, scalar
properties in dtype order); that byte layout is restated here with numpy structured arrays -- no extra dependency.
"""
import os

import numpy as np
import torch

C0 = 0.28209479177387814       # SH2RGB, gs_core.py (utils.sh_utils): rgb = sh * C0 + 0.5

_PLY_TYPES = {"f4": "float", "f2": None, "u1": "uchar"}


def get_turntable_cameras(hfov=50, num_views=8, w=384, h=384, radius=2.7, elevation=0, up_vector=np.array([0, 0, 1])):
    """gs_core.py:50-85.  -> (w, h, num_views, fxfycxcy [v, 4], c2w [v, 4, 4]) float64, OpenCV convention."""
    fx = w / (2 * np.tan(np.deg2rad(hfov) / 2.0))
    fxfycxcy = np.array([fx, fx, w / 2.0, h / 2.0]).reshape(1, 4).repeat(num_views, axis=0)
    c2ws = []
    for azim in np.linspace(0, 360, num_views, endpoint=False):
        elev, azim = np.deg2rad(elevation), np.deg2rad(azim)
        base = radius * np.cos(elev)
        cam_pos = np.array([base * np.cos(azim), base * np.sin(azim), radius * np.sin(elev)])
        forward = -cam_pos / np.linalg.norm(cam_pos)
        right = np.cross(forward, up_vector)
        right = right / np.linalg.norm(right)
        up = np.cross(right, forward)
        up = up / np.linalg.norm(up)
        c2w = np.eye(4)
        c2w[:3, :4] = np.concatenate((np.stack((right, -up, forward), axis=1), cam_pos[:, None]), axis=1)
        c2ws.append(c2w)
    return w, h, num_views, fxfycxcy, np.stack(c2ws, axis=0)


def render_turntable(pc, rendering_resolution=384, num_views=8, backend=None):
    """gs_core.py:1203-1219: `pc` a GaussianModel -> uint8 image strip [h, v*w, 3]."""
    from .raster import default_backend
    w, h, v, fxfycxcy, c2w = get_turntable_cameras(h=rendering_resolution, w=rendering_resolution, num_views=num_views)
    dev = pc._xyz.device
    k = torch.from_numpy(fxfycxcy).float().to(dev)[None]
    c = torch.from_numpy(c2w).float().to(dev)[None]
    be = backend if backend is not None else default_backend()
    feats = pc.get_features.float()[None]
    r = be.render_views(pc._xyz.float()[None], feats, pc._scaling.float()[None], pc._rotation.float()[None], pc._opacity.float()[None],
                        h, w, c, k)[0]                                   # [v, 3, h, w]
    r = (r.detach().cpu().numpy() * 255).clip(0, 255).astype(np.uint8)
    return np.ascontiguousarray(r.transpose(2, 0, 3, 1).reshape(h, v * w, 3))       # "v c h w -> h (v w) c"


def render_generic(pc, c2ws, fxfycxcy, h=512, w=512, backend=None):
    """gs_core.py:1300-1316: `pc` a GaussianModel, c2ws [v, 4, 4], fxfycxcy [v, 4] -> uint8 [v, h, w, 3].  The reference loops
    render_opencv_cam over the views; here they are one batched launch sequence."""
    from .raster import default_backend
    dev = pc._xyz.device
    c = torch.as_tensor(c2ws).float().to(dev)[None]
    k = torch.as_tensor(fxfycxcy).float().to(dev)[None]
    be = backend if backend is not None else default_backend()
    with torch.no_grad():
        r = be.render_views(pc._xyz.float()[None], pc.get_features.float()[None], pc._scaling.float()[None], pc._rotation.float()[None],
                            pc._opacity.float()[None], h, w, c, k)[0]                       # [v, 3, h, w]
    r = (r.detach().cpu().numpy() * 255).clip(0, 255).astype(np.uint8)
    return np.ascontiguousarray(r.transpose(0, 2, 3, 1))                                   # "v c h w -> v h w c"


def construct_dtypes(pc, enable_gs_viewer=True):
    """gs_core.py:578-606 (fp32 variant; PLY has no 16-bit float type)."""
    l = [("x", "f4"), ("y", "f4"), ("z", "f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")]
    l += [(f"f_dc_{i}", "f4") for i in range(pc._features_dc.shape[1] * pc._features_dc.shape[2])]
    if enable_gs_viewer:
        assert pc.sh_degree <= 3, "GS viewer only supports SH up to degree 3"
        l += [(f"f_rest_{i}", "f4") for i in range(((3 + 1) ** 2 - 1) * 3)]
    elif pc.sh_degree > 0:
        l += [(f"f_rest_{i}", "f4") for i in range(pc._features_rest.shape[1] * pc._features_rest.shape[2])]
    l.append(("opacity", "f4"))
    l += [(f"scale_{i}", "f4") for i in range(pc._scaling.shape[1])]
    l += [(f"rot_{i}", "f4") for i in range(pc._rotation.shape[1])]
    return l


def save_ply(pc, path, enable_gs_viewer=True, filter_mask=None):
    """gs_core.py:636-712: raw (pre-activation) parameters, DC colour also as 8-bit rgb, f_rest zero-padded to SH degree 3."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    n = lambda t: t.detach().float().cpu().numpy()
    xyz = n(pc._xyz)
    f_dc = n(pc._features_dc.transpose(1, 2).flatten(start_dim=1))
    rgb = ((f_dc * C0 + 0.5) * 255.0).clip(0.0, 255.0).astype(np.uint8)
    if pc.scaling_modifier is not None:
        scale = np.log(n(pc.get_scaling))
    else:
        scale = n(pc._scaling)
    f_rest = n(pc._features_rest.transpose(1, 2).flatten(start_dim=1)) if pc.sh_degree > 0 else None
    if enable_gs_viewer:
        full = np.zeros((xyz.shape[0], 3 * ((3 + 1) ** 2 - 1)), dtype=np.float32)
        if f_rest is not None:
            full[:, :f_rest.shape[1]] = f_rest
        f_rest = full
    dtype = construct_dtypes(pc, enable_gs_viewer)
    el = np.empty(xyz.shape[0], dtype=dtype)
    cols = [xyz, rgb, f_dc] + ([f_rest] if f_rest is not None else []) + [n(pc._opacity), scale, n(pc._rotation)]
    names = [name for name, _ in dtype]
    i = 0
    for block in cols:
        for c in range(block.shape[1]):
            el[names[i]] = block[:, c]
            i += 1
    assert i == len(names)
    if filter_mask is not None:
        el = el[np.asarray(filter_mask)]
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {el.shape[0]}"]
    header += [f"property {_PLY_TYPES[t]} {name}" for name, t in dtype] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(el.astype(el.dtype.newbyteorder("<"), copy=False).tobytes())


def read_ply(path):
    """-> structured array of the 'vertex' element (binary little endian, scalar float / uchar properties)."""
    rev = {"float": "<f4", "float32": "<f4", "uchar": "u1", "uint8": "u1", "double": "<f8", "float64": "<f8", "int": "<i4", "int32": "<i4"}
    with open(path, "rb") as f:
        assert f.readline().strip() == b"ply"
        fmt = f.readline().split()
        assert fmt[:2] == [b"format", b"binary_little_endian"], "only binary little-endian files (what save_ply writes)"
        count, props, in_vertex = 0, [], False
        for line in iter(f.readline, b""):
            tok = line.decode("ascii").split()
            if tok[0] == "end_header":
                break
            if tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
            elif tok[0] == "property" and in_vertex:
                props.append((tok[2], rev[tok[1]]))
        return np.frombuffer(f.read(count * np.dtype(props).itemsize), dtype=props, count=count)


def load_ply(pc, path, device="cpu"):
    """gs_core.py:715-760: fills `pc` (raw parameters) from a file written by save_ply / any 3DGS exporter."""
    v = read_ply(path)
    xyz = np.stack((v["x"], v["y"], v["z"]), axis=1)
    dc = np.stack((v["f_dc_0"], v["f_dc_1"], v["f_dc_2"]), axis=1)[:, :, None]                      # [P, 3, 1]
    idx = lambda prefix: sorted((n for n in v.dtype.names if n.startswith(prefix)), key=lambda s: int(s.split("_")[-1]))
    feats = dc
    if pc.sh_degree > 0:
        k = 3 * (pc.sh_degree + 1) ** 2 - 3
        extra = np.stack([v[n] for n in idx("f_rest_")[:k]], axis=1).reshape(xyz.shape[0], 3, (pc.sh_degree + 1) ** 2 - 1)
        feats = np.concatenate((dc, extra), axis=2)
    scales = np.stack([v[n] for n in idx("scale_")], axis=1)
    rots = np.stack([v[n] for n in idx("rot_")], axis=1)
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)
    return pc.set_data(t(xyz), t(feats).transpose(1, 2).contiguous(), t(scales), t(rots), t(v["opacity"][:, None]))
