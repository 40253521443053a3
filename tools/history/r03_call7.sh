#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03g
mkdir -p $out
cd $R
for rep in 1 2; do
 for lib in base new; do
  for regime in trained init; do
    L=$R/open-diffusiongs_amd/lib/libdgs_hip.so; [ $lib = base ] && L=$R/open-diffusiongs_amd/lib/libdgs_hip_base.so
    DGS_AMD_LIBRARY=$L timeout 120 python tools/raster_microbench.py --res 256 --regime $regime 2>&1 | grep -E "sync|forward\+backward" | sed "s/^/$lib $regime: /" >> $out/raster_fast_arith_ab.txt
  done
 done
done
cat $out/raster_fast_arith_ab.txt
timeout 600 python -m pytest tests/test_raster_forward_gpu.py tests/test_raster_backward_gpu.py tests/test_raster_ref_gpu.py tests/test_ref_glue_gpu.py -m gpu -q 2>&1 | tail -5 > $out/pytest_raster.txt; cat $out/pytest_raster.txt
FILL_MIN_MIB=0 timeout 300 python tools/find_fills.py 2>&1 | grep -v amdgpu.ids | tail -32 > $out/find_fills.txt; cat $out/find_fills.txt
