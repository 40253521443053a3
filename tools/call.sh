#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 300 python tools/rccl_neighbour_cost.py > $out/rccl_neighbour.txt 2>&1; grep -v amdgpu $out/rccl_neighbour.txt
( time timeout 1700 python -m pytest tests -m gpu -q > $out/pytest_gpu_full.txt 2>&1 ) 2>&1 | grep real
grep -v "^  File\|^$" $out/pytest_gpu_full.txt | tail -12 | cut -c1-300
