#!/bin/bash
# Round 3, call 26: tile counts from a difference grid (four atomics per Gaussian) -- rasterizer GPU parity tests (bit-exact integer
# artefacts vs the oracle and the reference's own code), microbenchmark of both regimes against the library before, kernel times.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03z
mkdir -p $out
cd $R
L=$R/open-diffusiongs_amd/lib
timeout 900 python -m pytest tests/test_raster_forward_gpu.py tests/test_raster_ref_gpu.py tests/test_raster_backward_gpu.py tests/test_ref_glue_gpu.py -m gpu -q -x 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 > $out/pytest_raster.txt; cat $out/pytest_raster.txt
for regime in init trained; do
  for lib in base new; do
    if [ $lib = base ]; then export DGS_AMD_LIBRARY=$L/libdgs_hip_base.so; else unset DGS_AMD_LIBRARY; fi
    timeout 200 python tools/raster_microbench.py --res 256 --regime $regime 2>&1 | grep -E "ms/call" | sed "s/^/$lib $regime: /" >> $out/raster_diff_grid_ab.txt
  done
done
unset DGS_AMD_LIBRARY
timeout 200 python tools/raster_microbench.py --res 512 --regime trained 2>&1 | grep -E "ms/call" | sed "s/^/new trained 512: /" >> $out/raster_diff_grid_ab.txt
cat $out/raster_diff_grid_ab.txt | cut -c1-200
PROF_LINES=12 timeout 300 tools/prof.sh r03z_raster_init -- python $R/tools/raster_microbench.py --res 256 --regime init > /dev/null
grep -E "preprocess|scan_tiles|blend_forward" gpurun_out/r03z_raster_init/kernel_stats.txt | cut -c1-150 | tee $out/raster_init_kernel_times.txt
timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | cut -c1-330
