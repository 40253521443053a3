#!/bin/bash
# round 4, call 9: what bounds the backward blend?  Ablations in the tools' library (raster_common.h kRasterAblate), both walks.
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/c9; mkdir -p $out
export DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so
for regime in trained init; do
  for walk in 1 2; do
    for ab in 0 1 2 3 16 18 4; do
      echo -n "$regime walk=$walk ablate=$ab " >> $out/ablate.txt
      DGS_RASTER_BWD_WALK=$walk DGS_RASTER_BWD_ABLATE=$ab timeout 300 python tools/raster_microbench.py --res 256 --views 4 --regime $regime --iters 10 2>&1 | grep -E "forward\+backward" >> $out/ablate.txt
    done
  done
done
cat $out/ablate.txt
