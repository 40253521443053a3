#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03p
mkdir -p $out
cd $R
L=open-diffusiongs_amd/lib
timeout 300 python tools/ln_bwd_ab.py $L/libdgs_hip_base2.so 2>&1 | grep -v amdgpu.ids > $out/ln_bwd_ab.txt; cat $out/ln_bwd_ab.txt
timeout 900 python -m pytest tests/test_dit_backward_gpu.py tests/test_dit_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v Warning | tail -6 > $out/pytest_dit.txt; cat $out/pytest_dit.txt
timeout 300 python tools/train_determinism.py 256 save_all 4 2 2>&1 | tail -3 > $out/determinism.txt; cat $out/determinism.txt
timeout 300 python bench.py --no-extras 2>/dev/null | cut -c1-300
