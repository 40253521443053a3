#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 600 python -m pytest tests/test_dit_gpu.py -m gpu -x -q -k "attention" 2>&1 | tail -3
timeout 600 python tools/attn_ab.py open-diffusiongs_amd/lib/libdgs_hip_base.so > $out/attn_ab.txt 2>&1; grep -v amdgpu $out/attn_ab.txt | tail -12
