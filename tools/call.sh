#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, calls 6-7): A/B of the attention backward's by-products (DGS_ATTN_BWD_BYPRODUCTS=0: transpose_kernel +
# colsum_wide_kernel behind it, as in rounds 3-5) inside one call: training step, alternating, then kernel stats of both.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
rm -f $out/train_ab.txt
for rep in 1 2 3; do
  for v in 0 1; do
    DGS_ATTN_BWD_BYPRODUCTS=$v timeout 300 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('byproducts=$v rep $rep train ms/step', d['ms_per_step'])" >> $out/train_ab.txt
  done
done
for v in 0 1; do
  DGS_ATTN_BWD_BYPRODUCTS=$v timeout 600 python bench.py --mode train-scene --scene-recompute off --steps 2 --warmup 1 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('byproducts=$v scene-512 save-all ms/step', d['ms_per_step'])" >> $out/train_ab.txt
done
cat $out/train_ab.txt
for v in 0 1; do
  DGS_ATTN_BWD_BYPRODUCTS=$v PROF_LINES=30 tools/prof.sh call_train_$v -- python $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > /dev/null
  echo "== byproducts=$v" >> $out/train_ab.txt
  grep -E "attention_bwd|transpose_kernel|colsum_wide|col_reduce" gpurun_out/call_train_$v/kernel_stats.txt >> $out/train_ab.txt
done
tail -14 $out/train_ab.txt
timeout 600 python -m pytest tests/test_dit_backward_gpu.py -x -q -m gpu -k "attention_backward or training_shape" > $out/pytest_bwd.txt 2>&1; tail -3 $out/pytest_bwd.txt
timeout 600 python tools/train_torch_ops.py > $out/train_torch_ops.txt 2>&1; head -40 $out/train_torch_ops.txt
