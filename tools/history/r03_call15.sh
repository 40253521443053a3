#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03o
mkdir -p $out
cd $R
L=open-diffusiongs_amd/lib
AB_TIMING_ONLY=1 timeout 400 python tools/attn_bwd_ab.py $L/libdgs_hip.so $L/libdgs_hip_k6.so $L/libdgs_hip_k7.so 2>&1 | grep -v amdgpu.ids > $out/attn_bwd_knockouts.txt; cat $out/attn_bwd_knockouts.txt
