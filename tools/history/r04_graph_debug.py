"""Debug aid (round 4): why a graph replay's rasterizer counts go wrong once its inputs change."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch
from dgs_amd import denoiser as dn, synth, raster

DEV = torch.device("cuda:0")
allocs = []
orig = raster.RasterBackend._allocator

def spy(holder, key, device):
    cb0 = orig(holder, key, device)
    def cb(nbytes, user):
        p = cb0(nbytes, user)
        allocs.append((key, int(holder[key].data_ptr()), int(nbytes), bool(torch.cuda.is_current_stream_capturing())))
        return p
    from dgs_amd import _native
    return _native.ALLOC_FN(cb)
raster.RasterBackend._allocator = staticmethod(spy)

m = dn.DGSDenoiser(dict(width=1024, in_channels=9, patch_size=8, num_layers=int(os.environ.get("LAYERS", "4"))), device=DEV)
m.reset_parameters(seed=1); m = m.to(DEV).eval()
res = int(os.environ.get("RES", "256"))
batch, t = synth.make_batch(1, res, V=4, device=DEV, seed=4, with_t=True)
print("input dtypes", {k: (v.dtype, tuple(v.shape)) for k, v in batch.items()}, t.dtype)
with torch.no_grad():
    ref, refg = m(batch, t); ref = ref.clone(); ref_xyz = refg[0]._xyz.clone()
    allocs.clear()
    g = m.graphed(batch, t)
    cap = [a for a in allocs if a[3]]
    print("allocations by the rasterizer's callbacks during capture:", [(k, hex(p), n) for k, p, n, c in cap])
    def stats(tag, want=None, want_xyz=None):
        torch.cuda.synchronize()
        s = g._stats.tolist()
        r, gs = g.rendered, g.gaussians
        msg = f"{tag}: stats N={s[0] & 0xFFFFFFFF} status={s[1]} longest={s[2]} nan={float(torch.isnan(r).float().mean()):.3f}"
        if want is not None:
            msg += f" render_equal={bool(torch.equal(r, want))} xyz_equal={bool(torch.equal(gs[0]._xyz, want_xyz))} xyz_maxdiff={float((gs[0]._xyz - want_xyz).abs().max()):.3e}"
        print(msg, flush=True)
    for i in range(3):
        g.graph.replay(); stats(f"constant replay {i}", ref, ref_xyz)
    # eager allocations between replays: do they land inside the graph's buffers?
    junk = [torch.full((n,), 7.0, device=DEV) for n in (256, 1024, 4096, 65536, 786432, 4 * 786432)]
    for j in junk:
        lo, hi = j.data_ptr(), j.data_ptr() + j.numel() * 4
        hit = [(k, hex(p)) for k, p, n, c in cap if not (hi <= p or lo >= p + n)]
        print(f"eager tensor {j.numel() * 4} B at {hex(lo)} overlaps capture allocations: {hit}")
    g.graph.replay(); stats("replay after eager allocations (inputs unchanged)", ref, ref_xyz)
    del junk
    b2, t2 = synth.make_batch(1, res, V=4, device=DEV, seed=9, with_t=True)
    for keys in (["t"], ["image"], ["ray_o", "ray_d"], ["c2w"], ["image", "ray_o", "ray_d", "c2w", "fxfycxcy", "t"]):
        bb = dict(batch); tt = t
        for k in keys:
            if k == "t": tt = t2
            else: bb[k] = b2[k]
        want, wg = m(bb, tt); want = want.clone(); wx = wg[0]._xyz.clone()
        torch.cuda.synchronize()
        for k, dst in g.static.items(): dst.copy_(bb[k])
        g.t.copy_(tt)
        g.graph.replay(); stats(f"changed {keys}", want, wx)
    # the same with a device synchronisation between the copies and the replay, and behind it
    for keys in (["image"], ["image", "ray_o", "ray_d", "c2w", "fxfycxcy", "t"]):
        bb = dict(batch); tt = t
        for k in keys:
            if k == "t": tt = t2
            else: bb[k] = b2[k]
        want, wg = m(bb, tt); want = want.clone(); wx = wg[0]._xyz.clone()
        torch.cuda.synchronize()
        for k, dst in g.static.items(): dst.copy_(bb[k])
        g.t.copy_(tt)
        torch.cuda.synchronize()
        g.graph.replay(); stats(f"changed {keys}, synchronised before the replay", want, wx)
    # back to the original inputs
    for k, dst in g.static.items(): dst.copy_(batch[k])
    g.t.copy_(t)
    g.graph.replay(); stats("original inputs again", ref, ref_xyz)
