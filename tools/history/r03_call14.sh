#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03n
mkdir -p $out
cd $R
DGS_ATTN_DBG=16 DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so timeout 200 python tools/attn_bwd_run.py 4098 4 2 2>&1 | grep "attn bwd dbg" | tail -16 > $out/dkv_phase_stamps.txt; cat $out/dkv_phase_stamps.txt
