#!/bin/bash
# Round 3, call 22: attention prologue with the counted wait (A/B vs HEAD), the learned-token side jobs behind a full round of tiles
# (fc1 / fc2), and who fills large tensors in a training step.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03v
mkdir -p $out
cd $R
L=open-diffusiongs_amd/lib
timeout 600 python -m pytest tests/test_dit_gpu.py -m gpu -q -x -k "attention" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 > $out/pytest_attention.txt; cat $out/pytest_attention.txt
timeout 300 python tools/attn_ab.py $L/libdgs_hip_base.so 2>&1 | grep -v amdgpu.ids > $out/attn_ab.txt; cat $out/attn_ab.txt
# (the side jobs behind a full round of tiles: tools/gemm_tailwgs_ab.py + the DGS_GEMM_TAIL_WGS knob existed for this call only -- fc1 43.4 -> 44.3 us at best, dropped)
FILL_MIN_MIB=16 timeout 400 python tools/find_fills.py 2>&1 | grep -v amdgpu.ids | tail -40 > $out/find_fills.txt; cat $out/find_fills.txt | cut -c1-250
