#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 25): the gradient-norm partial pass at smaller chunks (tools/ubench/sumsq_bench.hip); who fills large tensors
# on the training step path (tools/find_fills.py); the torch ops of a step (tools/train_torch_ops.py).
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
hipcc --offload-arch=gfx950 -O3 tools/ubench/sumsq_bench.hip -o /tmp/sumsq_bench 2> $out/sumsq_build.log && /tmp/sumsq_bench > $out/sumsq_chunk_bench.txt 2>&1
cat $out/sumsq_chunk_bench.txt
FILL_MIN_MIB=8 timeout 600 python tools/find_fills.py > $out/find_fills.txt 2>&1
tail -40 $out/find_fills.txt
timeout 600 python tools/train_torch_ops.py > $out/train_torch_ops.txt 2>&1
tail -60 $out/train_torch_ops.txt
