#!/bin/bash
# Round 3, call 21: attention with the tail tiles staged in LDS by the loop's last rounds and the sliced merge -- correctness (GPU
# tests from poisoned LDS, A/B bit-compare of the main rows), kernel durations under rocprofv3, phase stamps.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03u
mkdir -p $out
cd $R
L=open-diffusiongs_amd/lib
timeout 600 python -m pytest tests/test_dit_gpu.py -m gpu -q -x -k "attention" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -6 > $out/pytest_attention.txt; cat $out/pytest_attention.txt
timeout 300 python tools/attn_ab.py $L/libdgs_hip_base.so 2>&1 | grep -v amdgpu.ids > $out/attn_ab.txt; cat $out/attn_ab.txt
for LL in 4096 4098; do
  PROF_LINES=6 timeout 200 tools/prof.sh r03u_attn_$LL -- python $R/tools/attn_tail_cost.py $LL > /dev/null
  grep "attention_fwd" gpurun_out/r03u_attn_$LL/kernel_stats.txt | cut -c1-140 | sed "s/^/L=$LL  /" >> $out/attn_tail_kernel_times.txt
done
cat $out/attn_tail_kernel_times.txt
DGS_AMD_LIBRARY=$R/$L/libdgs_hip_instr.so DGS_ATTN_DBG=12 timeout 200 python tools/attn_tail_cost.py 4098 2>&1 | grep "attn dbg" | tail -4 > $out/attn_tail_stamps.txt
cat $out/attn_tail_stamps.txt
