#!/bin/bash
# Round 3, call 33 (29 with the sc1 LDS-DMA consumer mode): what a device-side hand-off between two workgroups of one launch costs (tools/ubench/xcd_handoff_bench): release /
# acquire fences against sc1 accesses, same XCD against the next XCD, with and without other dirty data in the L2s.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03l
mkdir -p $out
cd $R/tools/ubench
(timeout 60 ./xcd_handoff_bench 0; timeout 60 ./xcd_handoff_bench 64) > $out/r03_xcd_handoff_microbench.txt 2>&1
cat $out/r03_xcd_handoff_microbench.txt
