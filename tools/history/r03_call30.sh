#!/bin/bash
# Round 3, call 30: phase stamps inside the attention kernel's tail merge (instrumented library).
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03i2
mkdir -p $out
cd $R
DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so DGS_ATTN_DBG=12 timeout 200 python tools/attn_tail_cost.py 4098 2>&1 | grep "attn dbg" | tail -6 > $out/attn_merge_stamps.txt
cat $out/attn_merge_stamps.txt
