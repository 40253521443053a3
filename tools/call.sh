#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development (the experiments' own scripts live in tools/next/*.patch).
# This form: the GPU suite (slowest tests listed) and the smoke.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 1500 python -m pytest tests -x -q -m gpu --durations=8 > $out/pytest_gpu.txt 2>&1; tail -14 $out/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke > $out/smoke.txt 2>&1; tail -3 $out/smoke.txt
