#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
INSTR=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so
DGS_AMD_LIBRARY=$INSTR DGS_GEMM_DBG=1 GEMM_CASES=fc2,proj timeout 300 python tools/gemm_check.py 4 > $out/gemm_dbg_256x128.txt 2>&1
grep "gemm dbg\|timeline\] 1\|us  " $out/gemm_dbg_256x128.txt | cut -c1-330
GEMM_CASES=fc2,proj timeout 200 python tools/gemm_check.py 0,4 2>&1 | grep -v amdgpu
timeout 600 python -m pytest tests/test_raster_forward_gpu.py -m gpu -x -q -k "three_kernel" 2>&1 | tail -2
