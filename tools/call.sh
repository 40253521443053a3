#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 30): the learned tokens' QKV / fc1 rows inside the LayerNorm launch against the GEMMs' own side jobs
# (DGS_LN_ROWS_GEMV=0): the contract bench's step time, alternating, five repetitions.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
rm -f $out/ln_rows_gemv_steps.txt
for rep in 1 2 3 4 5; do for on in 0 1; do
  DGS_LN_ROWS_GEMV=$on python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rows_in_layernorm=$on rep $rep ms/step', d['ms_per_step'], 'attention us', d['roofline']['avg_launch_us'])" >> $out/ln_rows_gemv_steps.txt
done; done
cat $out/ln_rows_gemv_steps.txt
