#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development (the experiments' own scripts live in tools/next/*.patch).
# This form: the GPU suite and one bench line.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
