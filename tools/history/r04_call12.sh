#!/bin/bash
# round 4, call 12: are the blend walks bound by throughput or by the number of waves?  The tools' library launches the blend kernels with
# unused LDS so that 2 workgroups fit a CU instead of 4 (the 1,024 tiles of 4 views at 256^2 then run as two rounds at half the occupancy):
# a throughput-bound kernel takes the same time, a latency-bound one up to twice.
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out/c12; mkdir -p $out
export DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so
for regime in trained init; do
  for pad in 0 45000; do
    echo "== $regime forward pad=$pad" >> $out/occ.txt
    DGS_RASTER_FWD_LDS_PAD=$pad DGS_RASTER_BWD_ABLATE=4 timeout 300 python tools/raster_microbench.py --res 256 --views 4 --regime $regime --iters 10 2>&1 | grep -E "ms/call" >> $out/occ.txt
  done
  for pad in 0 30000; do
    echo "== $regime backward pad=$pad (forward unpadded)" >> $out/occ.txt
    DGS_RASTER_BWD_LDS_PAD=$pad timeout 300 python tools/raster_microbench.py --res 256 --views 4 --regime $regime --iters 10 2>&1 | grep -E "forward\+backward" >> $out/occ.txt
  done
done
cat $out/occ.txt
PROF_LINES=8 tools/prof.sh c12_pad -- env DGS_RASTER_FWD_LDS_PAD=45000 python $R/tools/raster_microbench.py --res 256 --regime trained > /dev/null; head -5 gpurun_out/c12_pad/kernel_stats.txt | tee $out/trained_pad_kernel_stats.txt
PROF_LINES=8 tools/prof.sh c12_nopad -- python $R/tools/raster_microbench.py --res 256 --regime trained > /dev/null; head -5 gpurun_out/c12_nopad/kernel_stats.txt | tee $out/trained_nopad_kernel_stats.txt
