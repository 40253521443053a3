#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so DGS_ATTN_DBG=28 timeout 200 python tools/attn_timeline.py > $out/attn_timeline.txt 2>&1
grep -v amdgpu $out/attn_timeline.txt | cut -c1-400
