#!/bin/bash
# round 4, call 8: where does the two-pixel backward walk start to pay?  (grid = views x tiles workgroups of 128 threads; 6 fit a CU)
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/c8; mkdir -p $out
for cfg in "256 6" "256 8" "256 16" "512 4"; do
  set -- $cfg
  for regime in trained init; do
    for walk in 2 1; do
      echo "== res=$1 views=$2 $regime walk=$walk" >> $out/ab.txt
      DGS_RASTER_BWD_WALK=$walk timeout 300 python tools/raster_microbench.py --res $1 --views $2 --regime $regime --iters 10 2>&1 | grep -E "forward\+backward" >> $out/ab.txt
    done
  done
done
cat $out/ab.txt
