#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 33): in-kernel timeline of QKV on 256 x 192 tiles against 256 x 256 (full tiles only), tools' library.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
rm -f $out/qkv_bn192_timeline.txt
for off in 1 0; do
  echo "== DGS_GEMM_NO_BN192=$off" >> $out/qkv_bn192_timeline.txt
  DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so DGS_GEMM_NO_BN192=$off GEMM_VALID=4096 GEMM_CASES=qkv DGS_GEMM_DBG=1 timeout 120 python tools/gemm_check.py 0 2>&1 | grep -v amdgpu.ids >> $out/qkv_bn192_timeline.txt
  DGS_GEMM_NO_BN192=$off GEMM_VALID=4096 GEMM_CASES=qkv,fc1 timeout 120 python tools/gemm_check.py 0 2>&1 | grep -v amdgpu.ids >> $out/qkv_bn192_timeline.txt
done
cat $out/qkv_bn192_timeline.txt
