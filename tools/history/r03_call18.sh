#!/bin/bash
# Round 3, call 18: A/B of the transposed-copy stores (16-byte token octets) and of the one-round-trip LayerNorm against the
# library of the previous commit (lib/libdgs_hip_base.so), the attention kernel with / without the two tail queries, RCCL test.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03r
mkdir -p $out
cd $R
L=open-diffusiongs_amd/lib
timeout 300 python tools/gemm_ab.py $L/libdgs_hip_base.so 2>&1 | grep -v amdgpu.ids > $out/gemm_ab.txt; cat $out/gemm_ab.txt
timeout 200 python tools/ln_ab.py $L/libdgs_hip_base.so 2>&1 | grep -v amdgpu.ids > $out/ln_ab.txt; cat $out/ln_ab.txt
timeout 200 python tools/attn_tail_cost.py 4096 4098 4128 2>&1 | grep -v amdgpu.ids > $out/attn_tail_cost.txt; cat $out/attn_tail_cost.txt
timeout 500 python -m pytest tests/test_rccl_world1_gpu.py -m gpu -q -s 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -30 > $out/r03_rccl_world1_pytest.txt; cut -c1-1500 $out/r03_rccl_world1_pytest.txt
timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | cut -c1-330
