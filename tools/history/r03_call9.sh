#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03i
mkdir -p $out
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short 2>&1 | grep -v Warning | tail -12 > $out/pytest_gpu.txt; cat $out/pytest_gpu.txt
for lib in base new; do
  for regime in trained init; do
    L=$R/open-diffusiongs_amd/lib/libdgs_hip.so; [ $lib = base ] && L=$R/open-diffusiongs_amd/lib/libdgs_hip_base.so
    DGS_AMD_LIBRARY=$L timeout 120 python tools/raster_microbench.py --res 256 --regime $regime 2>&1 | grep -E "sync|forward\+backward" | sed "s/^/$lib $regime: /" | sed 's/(gpu events).*->/->/' >> $out/raster_ab.txt
  done
done
cat $out/raster_ab.txt
timeout 300 python bench.py --no-extras 2>/dev/null | cut -c1-200
