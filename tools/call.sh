#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so DGS_GEMM_DBG=1 timeout 200 python tools/gemm_pair_ab.py > $out/pair_dbg.txt 2>&1
grep "timeline\] 1\|timeline\] 2\|gemm dbg\] M" $out/pair_dbg.txt | tail -8 | cut -c1-330
