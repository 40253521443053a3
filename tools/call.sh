#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 35): the adaLN GEMV (rowlinear_kernel<1>, 302 MB of weights) with its first features' weight rows requested
# before the input row is staged, against the three dependent round trips (DGS_ROWLINEAR_EARLY=0): kernel stats + step, alternating.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 600 python -m pytest tests/test_dit_gpu.py -x -q 2>&1 | tail -2 > $out/rowlinear_early_ab.txt
for rep in 1 2 3; do for on in 0 1; do
  DGS_ROWLINEAR_EARLY=$on python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rowlinear_early=$on rep $rep ms/step', d['ms_per_step'], 'attention us', d['roofline']['avg_launch_us'])" >> $out/rowlinear_early_ab.txt
done; done
for on in 0 1; do
  DGS_ROWLINEAR_EARLY=$on PROF_LINES=40 tools/prof.sh call_rowlin_$on -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --graph 0 > /dev/null
  echo "== kernel stats rowlinear_early=$on" >> $out/rowlinear_early_ab.txt
  grep -E "rowlinear|calls" gpurun_out/call_rowlin_$on/kernel_stats.txt | cut -c1-140 >> $out/rowlinear_early_ab.txt
done
cat $out/rowlinear_early_ab.txt
