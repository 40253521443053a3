#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 19): without the per-call zero fill of the tokenizer's operand: DiT / graph / sampler / caller GPU tests, bench.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 1500 python -m pytest tests/test_dit_gpu.py tests/test_graph_gpu.py tests/test_smoke_c1.py tests/test_ref_callers.py tests/test_sampler.py tests/test_denoiser_surface.py -x -q -m gpu > $out/pytest_dit_gpu.txt 2>&1; tail -3 $out/pytest_dit_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2> /dev/null | cut -c1-330
