"""Does a small-footprint kernel on a second stream run beside the DiT's one-round kernels without slowing them (tools/ubench/coreside_bench.hip)?
For each main kernel (the four block GEMMs at the shipped shape, attention at L = 4098): its launch time alone, with the background
GEMV-shaped kernel launched on another stream at the same moment, and how long the background kernel takes alone / beside it.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/libcoreside.so tools/ubench/coreside_bench.hip ; python tools/coreside_run.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "open-diffusiongs_amd"))
import torch

from dgs_amd import _native
from dgs_amd.dit import DitOps

DEV = "cuda:0"
bg = ctypes.CDLL(os.path.join(ROOT, "tools", "ubench", "libcoreside.so"))
bg.coreside_bg.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
ops = DitOps()
L, lpad, W = 4098, 4352, 1024
g = torch.Generator(device=DEV).manual_seed(0)
bf = lambda *s: torch.randn(*s, generator=g, device=DEV).to(torch.bfloat16)
xn, h = bf(lpad, W), bf(lpad, 4 * W)
w1, w2, wq, wp = bf(4 * W, W) * 0.02, bf(W, 4 * W) * 0.02, bf(3 * W, W) * 0.02, bf(W, W) * 0.02
x = torch.randn(lpad, W, device=DEV)
gate = torch.randn(1, W, device=DEV)
keep = {"qk": torch.zeros(lpad, 2 * W, dtype=torch.bfloat16, device=DEV), "vt": torch.zeros(1, W, lpad, dtype=torch.bfloat16, device=DEV),
        "o": torch.zeros(lpad, 4 * W, dtype=torch.bfloat16, device=DEV), "ao": torch.zeros(lpad, W, dtype=torch.bfloat16, device=DEV)}
kw = dict(rows_per_batch=lpad, valid_rows=L)
qkb, vtb = bf(lpad, 2 * W), bf(1, W, lpad)
mains = {
    "qkv": lambda: ops.gemm(xn, wq, None, _native.EPI_QKV, out=keep["qk"], vt=keep["vt"], **kw),
    "fc1": lambda: ops.gemm(xn, w1, None, _native.EPI_GELU_BF16, out=keep["o"], **kw),
    "fc2": lambda: ops.gemm(h, w2, None, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, **kw),
    "proj": lambda: ops.gemm(xn, wp, None, _native.EPI_GATE_RESIDUAL, out=x, gate=gate, **kw),
    "attention": lambda: ops.attention(qkb, vtb, L, 16, out=keep["ao"]),
}

# background: a 2-row GEMV's traffic -- all of a [4096, 1024] bf16 weight matrix (8.4 MB), one column (2 KiB = 2 chunks) per wave
NW_WAVES, CHUNKS = 4096, 2
wbuf = torch.randn(NW_WAVES * CHUNKS * 64 * 4, device=DEV, generator=g)            # uint4 = 4 floats
abuf = torch.randn(8 * 64 * 4, device=DEV, generator=g)
obuf = torch.zeros(NW_WAVES, device=DEV)
s_main, s_bg = torch.cuda.Stream(), torch.cuda.Stream()


def run_bg():
    rc = bg.coreside_bg(wbuf.data_ptr(), abuf.data_ptr(), obuf.data_ptr(), NW_WAVES // 4, CHUNKS, CHUNKS * 64, s_bg.cuda_stream)
    assert rc == 0


def timed(fn_main, with_bg, n=30, bg_reps=1):
    em = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    eb = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        torch.cuda.synchronize()
        go = torch.cuda.Event()
        go.record(torch.cuda.current_stream())
        s_main.wait_event(go); s_bg.wait_event(go)
        if fn_main is not None:
            with torch.cuda.stream(s_main):
                em[i][0].record(s_main); fn_main(); em[i][1].record(s_main)
        if with_bg:
            eb[i][0].record(s_bg)
            for _ in range(bg_reps):
                run_bg()
            eb[i][1].record(s_bg)
    torch.cuda.synchronize()
    med = lambda ev: sorted(a.elapsed_time(b) * 1e3 for a, b in ev)[n // 2]
    return (med(em) if fn_main is not None else None), (med(eb) if with_bg else None)


for _ in range(3):
    run_bg()
    for f in mains.values():
        with torch.cuda.stream(s_main):
            f()
torch.cuda.synchronize()
_, bg_alone = timed(None, True)
_, bg5_alone = timed(None, True, bg_reps=5)
print(f"background kernel alone: {bg_alone:.1f} us (8.4 MB as 4096 waves x 2 KiB); five in a row {bg5_alone:.1f} us", flush=True)
for name, f in mains.items():
    alone, _ = timed(f, False)
    both, bgt = timed(f, True)
    both5, bgt5 = timed(f, True, bg_reps=5)
    print(f"{name:10s} alone {alone:6.1f} us | beside 1 background launch {both:6.1f} us (background {bgt:5.1f} us) | beside 5 in a row {both5:6.1f} us (background chain {bgt5:6.1f} us)", flush=True)
