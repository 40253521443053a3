#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 37, EXPERIMENT): fc1 at one sample as 512 tiles of 256 x 128 (two rounds: the second round's loop over the
# first round's stores) against 256 tiles of 256 x 256 (one round); full tiles only.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
rm -f $out/fc1_two_rounds.txt
for rep in 1 2 3; do for f in 0 1; do
  echo "== force_bn128=$f rep $rep" >> $out/fc1_two_rounds.txt
  DGS_GEMM_FORCE_BN128=$f GEMM_VALID=4096 GEMM_CASES=fc1,fc2,proj timeout 120 python tools/gemm_check.py 4 2>&1 | grep -v amdgpu.ids >> $out/fc1_two_rounds.txt
done; done
cat $out/fc1_two_rounds.txt
