#!/bin/bash
# round 4, call 13: emit_instances_kernel with the tile's first slot folded into the cursor round trip
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/c13; mkdir -p $out
timeout 600 python -m pytest tests/test_raster_forward_gpu.py tests/test_raster_ref_gpu.py -m gpu -q 2>&1 | tail -2 > $out/pytest.txt; cat $out/pytest.txt
PROF_LINES=10 tools/prof.sh c13_trained -- python $R/tools/raster_microbench.py --res 256 --regime trained > /dev/null; cp gpurun_out/c13_trained/kernel_stats.txt $out/trained_kernel_stats.txt; head -8 $out/trained_kernel_stats.txt
python tools/raster_microbench.py --res 256 --regime trained 2>&1 | grep "ms/call"
