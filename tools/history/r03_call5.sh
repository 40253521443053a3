#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03e
mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_raster_forward_gpu.py tests/test_raster_backward_gpu.py tests/test_losses.py tests/test_raster_ref_gpu.py -m gpu -q 2>&1 | tail -40 > $out/pytest_raster.txt
cat $out/pytest_raster.txt
timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_raster_forward_gpu.py --deselect tests/test_raster_backward_gpu.py --deselect tests/test_losses.py --deselect tests/test_raster_ref_gpu.py 2>&1 | tail -6 > $out/pytest_rest.txt
cat $out/pytest_rest.txt
PROF_LINES=60 tools/prof.sh r03e_train -- python $R/bench.py --mode train --steps 3 --warmup 1 > /dev/null
cp gpurun_out/r03e_train/kernel_stats.txt $out/train_step_kernel_stats.txt
grep -E "ms_per_step" gpurun_out/r03e_train/run.log | cut -c1-200
head -45 $out/train_step_kernel_stats.txt | cut -c1-150
timeout 200 python bench.py --mode train --steps 5 --warmup 2 2>/dev/null | cut -c1-250
