#!/bin/bash
# Round 3, call 19: what the two tail queries cost the attention kernel -- kernel durations at L = 4096 / 4098 under rocprofv3 and
# the phase stamps of workgroup 0 (instrumented library).
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03s
mkdir -p $out
cd $R
for L in 4096 4098; do
  PROF_LINES=6 timeout 200 tools/prof.sh r03s_attn_$L -- python $R/tools/attn_tail_cost.py $L > /dev/null
  grep "attention_fwd" gpurun_out/r03s_attn_$L/kernel_stats.txt | cut -c1-140 | sed "s/^/L=$L  /" >> $out/attn_tail_kernel_times.txt
done
cat $out/attn_tail_kernel_times.txt
for L in 4096 4098; do
  DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so DGS_ATTN_DBG=12 timeout 200 python tools/attn_tail_cost.py $L 2>&1 | grep "attn dbg" | tail -4 | sed "s/^/L=$L  /" >> $out/attn_tail_stamps.txt
done
cat $out/attn_tail_stamps.txt
