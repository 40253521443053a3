#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 31): QKV at one sample on 256 x 192 tiles (256 of them, one per CU) against 256 x 256 (192 tiles;
# DGS_GEMM_NO_BN192=1): DiT GPU tests, the contract bench's step time alternating, kernel stats of both.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 900 python -m pytest tests/test_dit_gpu.py tests/test_abi.py -x -q 2>&1 | tail -5 > $out/bn192_tests.txt
cat $out/bn192_tests.txt
rm -f $out/qkv_bn192_ab.txt
for rep in 1 2 3 4 5; do for off in 1 0; do
  DGS_GEMM_NO_BN192=$off python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('no_bn192=$off rep $rep ms/step', d['ms_per_step'], 'attention us', d['roofline']['avg_launch_us'])" >> $out/qkv_bn192_ab.txt
done; done
for off in 1 0; do
  DGS_GEMM_NO_BN192=$off PROF_LINES=12 tools/prof.sh call_bn192_$off -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --graph 0 > /dev/null
  echo "== kernel stats no_bn192=$off" >> $out/qkv_bn192_ab.txt
  head -10 gpurun_out/call_bn192_$off/kernel_stats.txt | cut -c1-140 >> $out/qkv_bn192_ab.txt
done
cat $out/qkv_bn192_ab.txt
