"""Run-to-run determinism of the training forward / backward (development tool): the saved-activation arena and the flat
gradient buffer of two identical passes are compared byte for byte (tensors summed with fp32 atomics are listed, not failed)."""
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
B, V, res = 2, 4, int(sys.argv[1]) if len(sys.argv) > 1 else 64
recompute = len(sys.argv) > 2 and sys.argv[2] == "recompute"
images, ray_o, ray_d, t, _, _ = synth_inputs(cfg, B, V, res, seed=6)
eng = DitEngine(sd, device=DEV)
FIELDS = ("xyz", "features", "scaling", "rotation", "opacity")
arenas, grads = [], []
for it in range(2):
    out, _ = eng.forward_train(images, ray_o, ray_d, t, recompute=recompute)
    torch.cuda.synchronize()
    arenas.append(eng._train["saved"].clone())
    g = torch.Generator(device=DEV).manual_seed(1)
    wts = {k: torch.randn(out[k].shape, generator=g, device=DEV) for k in FIELDS}
    eng.backward(*(wts[k] for k in FIELDS))
    torch.cuda.synchronize()
    grads.append({k: v.clone() for k, v in eng.grad_views().items()})
diff = (arenas[0] != arenas[1])
print("arena bytes differing:", int(diff.sum()), "of", diff.numel(), "first at", int(diff.nonzero()[0]) if diff.any() else None)
bad = [k for k in grads[0] if not torch.equal(grads[0][k], grads[1][k])]
print("gradient tensors differing run to run:", len(bad), "of", len(grads[0]))
print("  2-D block weights among them:", [k for k in bad if k.startswith("transformer") and k.endswith("weight") and "adaLN" not in k][:8])
