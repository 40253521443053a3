"""Debug aid (round 4): PSNR of the HIP render against the oracle's render of the same Gaussians -- eager and graph, plain and noise inputs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import numpy as np, torch
from dgs_amd import denoiser as dn, synth
from oracle import dit_oracle as D, raster_oracle as RO
RO.build()
DEV = torch.device("cuda:0")
m = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=int(os.environ.get("LAYERS", "24"))), device=DEV)
m.reset_parameters(seed=0); m = m.to(DEV).eval()
res = 256
batch, t = synth.make_batch(1, res, V=4, device=DEV, seed=0, with_t=True)

def psnr(render, gm, c2w, k):
    view, proj, campos, tanfov = D.camera_matrices(c2w[0].cpu(), k[0].cpu(), res, res)
    a = dict(xyz=gm._xyz.cpu().numpy(), shs=gm._features_dc.cpu().numpy(), op=torch.sigmoid(gm._opacity).cpu().numpy(),
             sc=torch.exp(gm._scaling).cpu().numpy(), rot=torch.nn.functional.normalize(gm._rotation).cpu().numpy())
    o = RO.RasterOracle()
    o.forward(np.ones(3, np.float32), a["xyz"], a["op"], view[0].numpy(), proj[0].numpy(), campos[0].numpy(), float(tanfov[0, 0]), float(tanfov[0, 1]),
              res, res, shs=a["shs"], scales=a["sc"], rotations=a["rot"], exp_mode=1)
    ref = np.clip(o.get("out_color"), 0, 1); mine = np.clip(render[0, 0].cpu().numpy(), 0, 1)
    mse = float(np.mean((ref.astype(np.float64) - mine) ** 2))
    stats = {k_: (float(np.nanmin(v)), float(np.nanmax(v)), int(np.isnan(v).sum())) for k_, v in a.items()}
    return (200.0 if mse == 0 else -10 * np.log10(mse)), stats

with torch.no_grad():
    for name, bb, tt in (("plain inputs", batch, t),
                         ("noise images, t = 0", dict(batch, image=torch.cat([batch["image"][:, :1], torch.randn_like(batch["image"][:, 1:])], 1)), torch.zeros_like(t)),
                         ("noise images, t = 999", dict(batch, image=torch.cat([batch["image"][:, :1], torch.randn_like(batch["image"][:, 1:])], 1)), torch.full_like(t, 999))):
        r, g = m(bb, tt)
        p, st = psnr(r, g[0], bb["c2w"], bb["fxfycxcy"])
        print(f"eager, {name}: PSNR vs oracle {p:.1f} dB  ranges {st}", flush=True)
        gr = m.graphed(bb, tt)
        r2, g2 = gr(bb, tt)
        torch.cuda.synchronize()
        p2, _ = psnr(r2, g2[0], bb["c2w"], bb["fxfycxcy"])
        print(f"graph, {name}: PSNR vs oracle {p2:.1f} dB  equal to eager: {bool(torch.equal(r, r2))}", flush=True)
