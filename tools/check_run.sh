#!/bin/bash
# A short GPU call (from the repo root on the GPU box): the newest GPU test first, then the whole GPU suite, then the contract
# bench; every result lands under gpurun_out/check/ as soon as it exists (the call may be cut off by the GPU-minute budget).
set -u
tag=${1:-r02b}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/check
mkdir -p $out
cd $R
timeout 400 python -m pytest tests/test_smoke_c1.py -m gpu -q 2>&1 | tail -6 > $out/${tag}_c1_gpu.txt
cat $out/${tag}_c1_gpu.txt
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $out/${tag}_pytest_gpu.txt
cat $out/${tag}_pytest_gpu.txt
timeout 400 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
cut -c1-400 $out/${tag}_bench.json
