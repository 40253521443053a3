#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03h
mkdir -p $out
cd $R
timeout 300 python -m pytest tests/test_raster_forward_gpu.py -m gpu -q -x -k "dropin" 2>&1 | grep -E "^E|assert|Error" | head -20 > $out/dropin_fail.txt; cat $out/dropin_fail.txt
timeout 600 python -m pytest tests/test_raster_forward_gpu.py tests/test_raster_backward_gpu.py tests/test_raster_ref_gpu.py tests/test_ref_glue_gpu.py tests/test_smoke_c1.py -m gpu -q 2>&1 | tail -6 > $out/pytest_raster.txt; cat $out/pytest_raster.txt
for rep in 1 2; do
 for lib in base new; do
  for regime in trained init; do
    L=$R/open-diffusiongs_amd/lib/libdgs_hip.so; [ $lib = base ] && L=$R/open-diffusiongs_amd/lib/libdgs_hip_base.so
    DGS_AMD_LIBRARY=$L timeout 120 python tools/raster_microbench.py --res 256 --regime $regime 2>&1 | grep -E "sync|forward\+backward" | sed "s/^/$lib $regime: /" | sed 's/(gpu events).*->/->/' >> $out/raster_walk_loop_ab.txt
  done
 done
done
cat $out/raster_walk_loop_ab.txt
