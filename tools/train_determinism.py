"""Run-to-run determinism of the training forward / backward (development tool): the saved-activation arena and the flat
gradient buffer of N identical passes are compared byte for byte.  The backward has no fp32 atomics (column sums go through slabs of
per-workgroup partial rows summed in a fixed order): every tensor must come out bit-identical; a tensor that does not is a RACE.
    python tools/train_determinism.py [res [recompute [B [passes]]]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "open-diffusiongs_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from dgs_amd.dit import DitEngine
from dit_util import synth_inputs
from oracle import dit_oracle as D
DEV = "cuda:0"
cfg = D.Cfg()
sd = D.parity_state_dict(cfg, seed=13)
B, V, res = (int(sys.argv[3]) if len(sys.argv) > 3 else 2), 4, int(sys.argv[1]) if len(sys.argv) > 1 else 64
recompute = len(sys.argv) > 2 and sys.argv[2] == "recompute"
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 2
images, ray_o, ray_d, t, _, _ = synth_inputs(cfg, B, V, res, seed=6)
eng = DitEngine(sd, device=DEV)
FIELDS = ("xyz", "features", "scaling", "rotation", "opacity")
arenas, grads = [], []
for it in range(passes):
    out, _ = eng.forward_train(images, ray_o, ray_d, t, recompute=recompute)
    torch.cuda.synchronize()
    arenas.append(eng._train["saved"].clone())
    g = torch.Generator(device=DEV).manual_seed(1)
    wts = {k: torch.randn(out[k].shape, generator=g, device=DEV) for k in FIELDS}
    eng.backward(*(wts[k] for k in FIELDS))
    torch.cuda.synchronize()
    grads.append({k: v.clone() for k, v in eng.grad_views().items()})
print(f"res {res}, B {B}, recompute {recompute}, {passes} passes")
worst = 0
for it in range(1, passes):
    diff = (arenas[0] != arenas[it])
    print(f"pass {it} vs 0: arena bytes differing:", int(diff.sum()), "of", diff.numel(), "first at", int(diff.nonzero()[0]) if diff.any() else None)
    bad = [k for k in grads[0] if not torch.equal(grads[0][k], grads[it][k])]
    worst = max(worst, len(bad))
    print(f"pass {it} vs 0: gradient tensors differing run to run:", len(bad), "of", len(grads[0]))
    for k in bad[:12]:
        a, b = grads[0][k].double(), grads[it][k].double()
        print(f"   {k}: {int((a != b).sum())} of {a.numel()} elements, max |diff| {float((a - b).abs().max()):.3g} (max |g| {float(a.abs().max()):.3g})")
print("DETERMINISTIC" if worst == 0 else "NOT DETERMINISTIC")
