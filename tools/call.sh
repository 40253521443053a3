#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 23): a world of one: the gradient norm's per-bucket partial sums on the compute stream (default now) against
# the side stream (DGS_NORM_SIDE_STREAM=1), training step, alternating inside one call; then the trainer GPU tests.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
rm -f $out/norm_stream_ab.txt
for rep in 1 2 3; do for v in 1 0; do
  DGS_NORM_SIDE_STREAM=$v timeout 300 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('norm_on_side_stream=$v rep $rep train ms/step', d['ms_per_step'])" >> $out/norm_stream_ab.txt
done; done
cat $out/norm_stream_ab.txt
timeout 900 python -m pytest tests/test_optim.py tests/test_rccl_world1_gpu.py tests/test_two_ranks_gpu.py tests/test_dit_backward_gpu.py -x -q -m gpu -k "trainer or training or rccl or two_ranks or optim" > $out/pytest_train.txt 2>&1; tail -3 $out/pytest_train.txt
