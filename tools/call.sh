#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 3): the whole GPU suite on the tree with the new exponential, the radix rescue path, the experiments out of
# csrc/; then the contract bench (proj / fc2 as families of their own, scene-512 training step in both modes).
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 600 python -m pytest tests/test_raster_forward_gpu.py -x -q -m gpu -k "radix or neighbour" --durations=5 > $out/pytest_radix.txt 2>&1; tail -12 $out/pytest_radix.txt
timeout 2400 python -m pytest tests -x -q -m gpu --durations=8 > $out/pytest_gpu.txt 2>&1; tail -14 $out/pytest_gpu.txt
timeout 300 python __graft_entry__.py --smoke > $out/smoke.txt 2>&1; tail -3 $out/smoke.txt
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-1500 $out/bench.json; tail -5 $out/bench.err
