#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 16): the denoiser heads at gaussians_sh_degree 1 on the GPU, then the whole GPU suite.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 600 python -m pytest tests/test_dit_gpu.py -x -q -m gpu -k "sh_degree or golden" > $out/pytest_sh.txt 2>&1; tail -5 $out/pytest_sh.txt
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.txt 2>&1; tail -3 $out/pytest_gpu.txt
