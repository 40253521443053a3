"""A/B of two builds of the library on the four GEMMs of a DiT block at the shipped shape (one sample at 256^2: 4,352 padded rows,
4,098 valid, incl. the two learned-token rows): bit-compare of the outputs, then alternating timed launches (HIP events).
    tools/ab_build.sh dit_gemm_deep.hip                   # "base" = the file at HEAD, "new" = the working tree
    python tools/gemm_ab.py <base.so> [<new.so>]          (new defaults to the product library)
`--cpu <rows>`: a dry run of this script on the CPU emulator build (both sides the same library, small shapes)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps

if "--cpu" in sys.argv:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emu_util import emu_lib
    DEV, M = "cpu", int(sys.argv[sys.argv.index("--cpu") + 1])
    L, base, new, iters = M - 126, DitOps(emu_lib()), DitOps(emu_lib()), 1
else:
    DEV, M, L, iters = "cuda:0", 4352, 4098, 40
    base = DitOps(_native.open_library(os.path.abspath(sys.argv[1])))
    new = DitOps(_native.open_library(os.path.abspath(sys.argv[2])) if len(sys.argv) > 2 else None)
g = torch.Generator(device=DEV).manual_seed(0)
bf = lambda *s: torch.randn(*s, generator=g, device=DEV).to(torch.bfloat16)
W = 1024 if DEV != "cpu" else 256
SHAPES = [("qkv", 3 * W, W, _native.EPI_QKV), ("proj", W, W, _native.EPI_GATE_RESIDUAL), ("fc1", 4 * W, W, _native.EPI_GELU_BF16),
          ("fc2", W, 4 * W, _native.EPI_GATE_RESIDUAL)]


def sync():
    if DEV != "cpu":
        torch.cuda.synchronize()


for name, N, K, epi in SHAPES:
    A, Wt, bias = bf(M, K), bf(N, K) * 0.05, torch.randn(N, generator=g, device=DEV)
    x0, gate = torch.randn(M, N, generator=g, device=DEV), torch.randn(1, N, generator=g, device=DEV)

    keep = {}                                                 # timed launches: preallocated outputs, nothing but the kernel

    def run(ops, x=None, timed=False):
        kw = dict(rows_per_batch=M, valid_rows=L)
        if timed and epi == _native.EPI_QKV:
            if "qk" not in keep:
                keep["qk"], keep["vt"] = torch.zeros(M, 2 * N // 3, dtype=torch.bfloat16, device=DEV), torch.zeros(1, N // 3, M, dtype=torch.bfloat16, device=DEV)
            return ops.gemm(A, Wt, bias, epi, out=keep["qk"], vt=keep["vt"], **kw)
        if timed and epi == _native.EPI_GELU_BF16:
            if "o" not in keep:
                keep["o"] = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
            return ops.gemm(A, Wt, bias, epi, out=keep["o"], **kw)
        if epi == _native.EPI_GATE_RESIDUAL:
            ops.gemm(A, Wt, bias, epi, out=x, gate=gate, **kw)
            return x
        if epi == _native.EPI_QKV:
            qk, vt = ops.gemm(A, Wt, bias, epi, **kw)
            return torch.cat([qk[:L].float(), vt[0, :, :L].t().float()], dim=1)
        return ops.gemm(A, Wt, bias, epi, **kw)

    if DEV != "cpu":
        base.poison_lds(); new.poison_lds()
    a, b = run(base, x0.clone())[:L], run(new, x0.clone())[:L]
    sync()
    print(f"{name}: outputs bit-identical: {torch.equal(a, b)}; finite: {bool(torch.isfinite(b.float()).all())}; "
          f"max |diff| {float((a.float() - b.float()).abs().max()):.3g}", flush=True)
    if DEV == "cpu":
        continue
    xs = {"base": x0.clone(), "new": x0.clone()}
    ev = {k: [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)] for k in xs}
    for _ in range(3):
        run(base, xs["base"], True); run(new, xs["new"], True)
    for i in range(iters):
        for k, ops in (("base", base), ("new", new)):
            e0, e1 = ev[k][i]
            e0.record(); run(ops, xs[k], True); e1.record()
    sync()
    med = {k: sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in v)[iters // 2] for k, v in ev.items()}
    flops = 2.0 * L * N * K
    print(f"timing {name} [{M} x {N} x {K}]: base {med['base']:.1f} us ({flops / med['base'] / 1e6:.0f} TFLOP/s)  new {med['new']:.1f} us "
          f"({flops / med['new'] / 1e6:.0f} TFLOP/s)  ratio {med['new'] / med['base']:.3f}  (median of {iters}; preallocated outputs)", flush=True)


# ---- training forward at 4 samples: every tile also leaves a transposed copy (`vt`: [sample][feature][token], the weight-gradient
#      GEMMs' operand) -- QKV (V^T only), fc1 + GELU with aux (pre-activations) and vt, the LN-output style plain BF16 with vt ----
if DEV != "cpu":
    B4, M4 = 4, 4 * 4352
    for name, N, K, epi in [("train qkv", 3 * W, W, _native.EPI_QKV), ("train fc1+gelu+vt", 4 * W, W, _native.EPI_GELU_BF16), ("train bf16+vt", W, W, _native.EPI_BF16)]:
        A, Wt, bias = bf(M4, K), bf(N, K) * 0.05, torch.randn(N, generator=g, device=DEV)
        res = {}
        for k, ops in (("base", base), ("new", new)):
            if epi == _native.EPI_QKV:
                out, vt = torch.zeros(M4, 2 * N // 3, dtype=torch.bfloat16, device=DEV), torch.zeros(B4, N // 3, 4352, dtype=torch.bfloat16, device=DEV)
                aux = None
            else:
                out, vt = torch.zeros(M4, N, dtype=torch.bfloat16, device=DEV), torch.zeros(B4, N, 4352, dtype=torch.bfloat16, device=DEV)
                aux = torch.zeros(M4, N, dtype=torch.bfloat16, device=DEV) if epi == _native.EPI_GELU_BF16 else None
            call = lambda ops=ops, out=out, vt=vt, aux=aux: ops.gemm(A, Wt, bias, epi, out=out, vt=vt, aux=aux, rows_per_batch=4352, valid_rows=L)
            for _ in range(3):
                call()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
            for e0, e1 in evs:
                e0.record(); call(); e1.record()
            sync()
            res[k] = (sorted(e0.elapsed_time(e1) * 1e3 for e0, e1 in evs)[10], out.clone(), vt.clone())
        same = torch.equal(res["base"][1], res["new"][1]) and torch.equal(res["base"][2], res["new"][2])
        print(f"{name} [{M4} x {N} x {K}]: outputs + transposed copies bit-identical: {same}; base {res['base'][0]:.1f} us  new {res['new'][0]:.1f} us  "
              f"ratio {res['new'][0] / res['base'][0]:.3f}", flush=True)
