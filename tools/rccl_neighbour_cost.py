"""What a concurrent RCCL-sized kernel costs the backward's one-round kernels on ONE GPU (the input DESIGN.md section 6's scaling
projection lacked): `wgs` workgroups of 512 threads streaming a copy on a second stream (tools/ubench/coreside_bench.hip bg_copy_kernel:
the shape of a ring all-reduce step on one rank) while the attention backward (dK / dV and dQ kernels, L = 4098, B = 4), a forward block
GEMM and a weight-gradient GEMM run on the main stream.  HIP events, median of 20; the copy is sized to outlast the main kernel.
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/ubench/libcoreside.so tools/ubench/coreside_bench.hip ; python tools/rccl_neighbour_cost.py"""
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
bg.coreside_copy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
ops = DitOps()
B, L, lpad, W, H = 4, 4098, 4352, 1024, 16
M = B * lpad
g = torch.Generator(device=DEV).manual_seed(0)
bf = lambda *s: (torch.randn(*s, generator=g, device=DEV) * 0.5).to(torch.bfloat16)
qkv, o, dO = bf(M, 3 * W), bf(M, W), bf(M, W)
qkvT = qkv.view(B, lpad, 3 * W).transpose(1, 2).contiguous()
dOT = dO.view(B, lpad, W).transpose(1, 2).contiguous()
lse2 = torch.zeros(B, H, lpad, device=DEV)
ops.attention(qkv, qkvT, L, H, qkv_layout=True, lse2=lse2, out=o)
xn, w1 = bf(M, W), bf(4 * W, W) * 0.04
hbuf = torch.zeros(M, 4 * W, dtype=torch.bfloat16, device=DEV)
mains = {
    "attention backward (dK/dV + dQ), B=4": lambda: ops.attention_backward(qkv, qkvT, o, dO, dOT, lse2, L, H),
    "fc1 forward GEMM, B=4": lambda: ops.gemm(xn, w1, None, _native.EPI_GELU_BF16, out=hbuf, rows_per_batch=lpad, valid_rows=L),
}
src = torch.empty(1 << 30, dtype=torch.uint8, device=DEV)
dst = torch.empty_like(src)
s_main, s_bg = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, wgs, nbytes, n=20):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    eb = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        torch.cuda.synchronize()
        go = torch.cuda.Event(); go.record(torch.cuda.current_stream())
        s_main.wait_event(go); s_bg.wait_event(go)
        if wgs:
            eb[i][0].record(s_bg)
            bg.coreside_copy(src.data_ptr(), dst.data_ptr(), nbytes, wgs, s_bg.cuda_stream)
            eb[i][1].record(s_bg)
        with torch.cuda.stream(s_main):
            ev[i][0].record(s_main); fn(); ev[i][1].record(s_main)
    torch.cuda.synchronize()
    med = lambda e: sorted(a.elapsed_time(b) * 1e3 for a, b in e)[n // 2]
    return med(ev), (med(eb) if wgs else None)


for name, fn in mains.items():
    for _ in range(3):
        with torch.cuda.stream(s_main):
            fn()
    torch.cuda.synchronize()
    alone, _ = timed(fn, 0, 0)
    line = f"{name}: alone {alone:8.1f} us"
    for wgs in (16, 32, 64):
        nbytes = int(min(1 << 30, max(1 << 26, alone * 1e-6 * wgs * 40e9)))        # ~40 GB/s per workgroup: the copy outlasts the main kernel
        both, copy = timed(fn, wgs, nbytes)
        line += f" | beside {wgs} copy workgroups {both:8.1f} us ({both / alone - 1:+.1%}; copy of {nbytes >> 20} MiB took {copy:.0f} us)"
    print(line, flush=True)
