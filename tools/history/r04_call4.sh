#!/bin/bash
# round 4, GPU call 4: why a graph replay goes wrong once its inputs change (debug script); deterministic raster backward (per-wave LDS copies)
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c4; mkdir -p $out; cd $R
timeout 300 python tools/history/r04_graph_debug.py > $out/graph_debug.txt 2>&1; cat $out/graph_debug.txt | grep -v amdgpu.ids
timeout 300 python -m pytest tests/test_raster_backward_gpu.py -m gpu -q -k "bit_reproducible or full_size_512" 2>&1 | tail -12 > $out/det_tests.txt; cat $out/det_tests.txt
timeout 300 python tools/raster_det_ab.py > $out/raster_det_ab.txt 2>&1; cat $out/raster_det_ab.txt | grep -v amdgpu.ids
DGS_RASTER_DETERMINISTIC=1 PROF_LINES=14 tools/prof.sh c4_prof_det_init -- python $R/tools/raster_microbench.py --regime init
DGS_RASTER_DETERMINISTIC=0 PROF_LINES=12 tools/prof.sh c4_prof_atomic_init -- python $R/tools/raster_microbench.py --regime init
timeout 300 python -m pytest tests/test_ref_callers.py -m gpu -q 2>&1 | tail -6 > $out/ref_callers.txt; cat $out/ref_callers.txt
