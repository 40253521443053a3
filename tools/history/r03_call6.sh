#!/bin/bash
# Round-3 GPU call 6: the fork-free / instrumentation-free kernels against the previous build (bit-identity + timing), GPU suite, who
# fills ~1 GB four times per training step, training + inference bench.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03f
mkdir -p $out
cd $R
timeout 300 python tools/gemm_ab.py open-diffusiongs_amd/lib/libdgs_hip_base.so 2>&1 | grep -v amdgpu.ids > $out/gemm_ab.txt; cat $out/gemm_ab.txt
timeout 300 python tools/attn_ab.py open-diffusiongs_amd/lib/libdgs_hip_base.so 2>&1 | grep -v amdgpu.ids > $out/attn_ab.txt; cat $out/attn_ab.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $out/pytest_gpu.txt; cat $out/pytest_gpu.txt
timeout 300 python tools/find_fills.py 2>&1 | grep -v amdgpu.ids | tail -20 > $out/find_fills.txt; cat $out/find_fills.txt
timeout 300 python bench.py --mode train --steps 5 --warmup 2 2>/dev/null > $out/train_bench.json; cut -c1-260 $out/train_bench.json
DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_base.so timeout 300 python bench.py --mode train --steps 5 --warmup 2 2>/dev/null | cut -c1-260
timeout 300 python bench.py --no-extras 2>/dev/null > $out/bench_noextras.json; cut -c1-260 $out/bench_noextras.json
DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_base.so timeout 300 python bench.py --no-extras 2>/dev/null | cut -c1-260
