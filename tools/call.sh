#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, calls 13-14): the depth range sort (window segments, 512-thread final stage) against the four-pass radix sort
# (DGS_RASTER_SORT=radix): raster GPU tests, microbenchmark alternating, kernel stats of both, the contract bench, training step.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 1500 python -m pytest tests/test_raster_forward_gpu.py tests/test_raster_backward_gpu.py tests/test_raster_ref_gpu.py tests/test_graph_gpu.py -x -q -m gpu > $out/pytest_raster_gpu.txt 2>&1; tail -3 $out/pytest_raster_gpu.txt
rm -f $out/depth_sort_ab.txt
for rep in 1 2; do
  for sort in radix range; do
    for regime in init trained; do
      echo "== sort=$sort $regime rep $rep" >> $out/depth_sort_ab.txt
      DGS_RASTER_SORT=$sort timeout 300 python tools/raster_microbench.py --res 256 --regime $regime --iters 50 2>&1 | grep -E "async|forward\+backward" >> $out/depth_sort_ab.txt
    done
  done
done
for sort in radix range; do
  DGS_RASTER_SORT=$sort PROF_LINES=30 tools/prof.sh call_sort_$sort -- python $R/tools/raster_microbench.py --res 256 --regime init > /dev/null
  echo "== kernel stats, init regime, sort=$sort" >> $out/depth_sort_ab.txt
  grep -E "radix|range_|preprocess_kernel|blend_forward|rank_rects|scan_tiles" gpurun_out/call_sort_$sort/kernel_stats.txt >> $out/depth_sort_ab.txt
done
for rep in 1 2; do for sort in radix range; do
  DGS_RASTER_SORT=$sort timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sort=$sort rep $rep bench ms/step', d['ms_per_step'])" >> $out/depth_sort_ab.txt
done; done
for sort in radix range; do
  DGS_RASTER_SORT=$sort timeout 300 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sort=$sort train ms/step', d['ms_per_step'])" >> $out/depth_sort_ab.txt
done
cat $out/depth_sort_ab.txt
