#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 42): proj (N = K = 1024) on the ring kernel's 128 x 128 tiles with two K groups (AUTO since this change)
# against the 128-wide two-stage kernel (DGS_GEMM_S128_MINK=2048): DiT GPU tests, step time alternating, kernel stats.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 900 python -m pytest tests/test_dit_gpu.py tests/test_dit_backward_gpu.py -x -q 2>&1 | tail -3 > $out/proj_kgroups_ab.txt
for rep in 1 2 3 4 5; do for mink in 2048 1024; do
  DGS_GEMM_S128_MINK=$mink python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('s128_mink=$mink rep $rep ms/step', d['ms_per_step'], 'attention us', d['roofline']['avg_launch_us'])" >> $out/proj_kgroups_ab.txt
done; done
for mink in 2048 1024; do
  DGS_GEMM_S128_MINK=$mink PROF_LINES=8 tools/prof.sh call_pk_$mink -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --graph 0 > /dev/null
  echo "== kernel stats s128_mink=$mink" >> $out/proj_kgroups_ab.txt
  head -7 gpurun_out/call_pk_$mink/kernel_stats.txt | cut -c1-140 >> $out/proj_kgroups_ab.txt
done
cat $out/proj_kgroups_ab.txt
