#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 39): the 4 x 2 XCD map of the one-round ring GEMMs (fc1, QKV) against whole tile rows per XCD
# (DGS_GEMM_NO_MAP2D=1): step time alternating, kernel stats, DiT GPU tests.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 900 python -m pytest tests/test_dit_gpu.py -x -q 2>&1 | tail -2 > $out/gemm_map2d_ab.txt
for rep in 1 2 3 4 5; do for off in 1 0; do
  DGS_GEMM_NO_MAP2D=$off python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('no_map2d=$off rep $rep ms/step', d['ms_per_step'], 'attention us', d['roofline']['avg_launch_us'])" >> $out/gemm_map2d_ab.txt
done; done
for off in 1 0; do
  DGS_GEMM_NO_MAP2D=$off PROF_LINES=8 tools/prof.sh call_map2d_$off -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --graph 0 > /dev/null
  echo "== kernel stats no_map2d=$off" >> $out/gemm_map2d_ab.txt
  head -7 gpurun_out/call_map2d_$off/kernel_stats.txt | cut -c1-140 >> $out/gemm_map2d_ab.txt
done
cat $out/gemm_map2d_ab.txt
