#!/bin/bash
# round 4, call 11: gradient records (one atomic instruction per 16-lane group), both walks; deterministic form on the same records
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/c11; mkdir -p $out
timeout 900 python -m pytest tests/test_raster_backward_gpu.py tests/test_raster_ref_gpu.py tests/test_raster_gpu.py -m gpu -q 2>&1 | tail -3 > $out/pytest.txt; cat $out/pytest.txt
for regime in trained init; do
  for walk in 1 2; do
    echo "== $regime walk=$walk" >> $out/ab.txt
    DGS_RASTER_BWD_WALK=$walk python tools/raster_microbench.py --res 256 --regime $regime 2>&1 | grep -E "forward\+backward" >> $out/ab.txt
  done
done
for cfg in "256 16" "512 4"; do set -- $cfg
  for walk in 1 2; do
    echo "== res=$1 views=$2 trained walk=$walk" >> $out/ab.txt
    DGS_RASTER_BWD_WALK=$walk timeout 300 python tools/raster_microbench.py --res $1 --views $2 --regime trained --iters 10 2>&1 | grep -E "forward\+backward" >> $out/ab.txt
  done
done
cat $out/ab.txt
DGS_RASTER_BWD_WALK=1 PROF_LINES=12 tools/prof.sh c11_trained -- python $R/tools/raster_microbench.py --res 256 --regime trained > /dev/null; cp gpurun_out/c11_trained/kernel_stats.txt $out/trained_kernel_stats_walk1.txt
head -5 $out/trained_kernel_stats_walk1.txt
DGS_RASTER_BWD_WALK=1 python tools/raster_det_ab.py > $out/det_ab_walk1.txt 2>&1; tail -12 $out/det_ab_walk1.txt
DGS_RASTER_BWD_WALK=2 python tools/raster_det_ab.py > $out/det_ab_walk2.txt 2>&1; tail -12 $out/det_ab_walk2.txt
