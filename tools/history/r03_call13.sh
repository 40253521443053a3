#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03m
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TCC_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $out/counters.txt
for shape in "4098 4 8"; do
  tag=$(echo $shape | tr ' ' '_')
  timeout 200 rocprofv3 --kernel-trace --stats -d $out/kt_$tag -- python $R/tools/attn_bwd_run.py $shape > /dev/null 2>&1
  python $R/tools/rocpd_stats.py $(find $out/kt_$tag -name "*.db" | head -1) 2>/dev/null | grep -E "^kernel|attention" > $out/stats_$tag.txt; cat $out/stats_$tag.txt
  timeout 400 python $R/tools/pmc_run.py $out/pmc_$tag -- python $R/tools/attn_bwd_run.py $shape | grep -A17 "attention_bwd" > $out/pmc_$tag.txt; cat $out/pmc_$tag.txt
done
find $out -size +2M -delete
