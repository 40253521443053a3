#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 24): the per-tile LDS sorter with 512 threads for 4,096 .. 8,192-entry capacities against 256 threads
# (DGS_RASTER_BITONIC_NT=256), trained-like regime: microbenchmark (sync line: capacity from the longest list), kernel stats, raster tests.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
rm -f $out/tile_sort_threads_ab.txt
for rep in 1 2; do for nt in 256 512; do
  echo "== NT=$nt trained rep $rep" >> $out/tile_sort_threads_ab.txt
  DGS_RASTER_BITONIC_NT=$nt timeout 300 python tools/raster_microbench.py --res 256 --regime trained --iters 50 2>&1 | grep -E "sync|async|forward\+backward" >> $out/tile_sort_threads_ab.txt
done; done
for nt in 256 512; do
  DGS_RASTER_BITONIC_NT=$nt PROF_LINES=30 tools/prof.sh call_nt_$nt -- python $R/tools/raster_microbench.py --res 256 --regime trained > /dev/null
  echo "== kernel stats NT=$nt" >> $out/tile_sort_threads_ab.txt
  grep -E "tile_bitonic|emit_instances|blend_forward" gpurun_out/call_nt_$nt/kernel_stats.txt >> $out/tile_sort_threads_ab.txt
done
python - >> $out/tile_sort_threads_ab.txt 2>&1 <<'PY'
import os, subprocess, sys, json
for nt in ("256", "512", "256", "512"):
    env = dict(os.environ, DGS_RASTER_BITONIC_NT=nt)
    out = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True).stdout.strip().splitlines()
    d = json.loads(out[-1])
    r = d["raster"]["trained"]
    print("NT", nt, "bench raster.trained forward ms", r["forward"]["ms"], "forward+backward ms", r["forward_backward"]["ms"])
PY
cat $out/tile_sort_threads_ab.txt
timeout 1500 python -m pytest tests/test_raster_forward_gpu.py tests/test_raster_backward_gpu.py tests/test_raster_ref_gpu.py -x -q -m gpu > $out/pytest_raster_gpu.txt 2>&1; tail -3 $out/pytest_raster_gpu.txt
