#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03k
mkdir -p $out
cd $R
L=open-diffusiongs_amd/lib
timeout 400 python tools/attn_bwd_ab.py $L/libdgs_hip_base.so $L/libdgs_hip_v6.so $L/libdgs_hip_v8.so 2>&1 | grep -v amdgpu.ids > $out/attn_bwd_ab.txt; cat $out/attn_bwd_ab.txt
timeout 600 python -m pytest tests/test_dit_backward_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v Warning | tail -6 > $out/pytest_dit_bwd.txt; cat $out/pytest_dit_bwd.txt
DGS_ATTN_DBG=16 DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so timeout 200 python tools/attn_bwd_run.py 4098 4 2 2>&1 | grep "attn bwd dbg" | tail -16 > $out/dkv_phase_stamps.txt; cat $out/dkv_phase_stamps.txt
