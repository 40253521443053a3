#!/bin/bash
# round 4, call 7: two-pixel backward walk (blend_backward_pair_kernel) against the one-pixel walk, same box
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/c7; mkdir -p $out
timeout 900 python -m pytest tests/test_raster_backward_gpu.py tests/test_raster_ref_gpu.py -m gpu -q 2>&1 | tail -3 > $out/pytest.txt; cat $out/pytest.txt
for regime in trained init; do
  for walk in 2 1; do
    echo "== $regime walk=$walk" >> $out/ab.txt
    DGS_RASTER_BWD_WALK=$walk python tools/raster_microbench.py --res 256 --regime $regime 2>&1 | grep -E "ms/call" >> $out/ab.txt
  done
done
cat $out/ab.txt
PROF_LINES=12 tools/prof.sh c7_trained -- python $R/tools/raster_microbench.py --res 256 --regime trained > /dev/null; cp gpurun_out/c7_trained/kernel_stats.txt $out/trained_kernel_stats.txt
PROF_LINES=12 tools/prof.sh c7_init -- python $R/tools/raster_microbench.py --res 256 --regime init > /dev/null; cp gpurun_out/c7_init/kernel_stats.txt $out/init_kernel_stats.txt
head -6 $out/trained_kernel_stats.txt; head -8 $out/init_kernel_stats.txt
python tools/raster_det_ab.py > $out/det_ab.txt 2>&1; tail -12 $out/det_ab.txt
