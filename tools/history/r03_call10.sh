#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03j
mkdir -p $out
cd $R
timeout 300 python tools/attn_ab.py open-diffusiongs_amd/lib/libdgs_hip_base.so 2>&1 | grep -v amdgpu.ids > $out/attn_rowsum_valu_ab.txt; cat $out/attn_rowsum_valu_ab.txt
timeout 600 python -m pytest tests/test_dit_gpu.py tests/test_dit_backward_gpu.py -m gpu -q --tb=short -x 2>&1 | grep -v Warning | tail -6 > $out/pytest_dit.txt; cat $out/pytest_dit.txt
timeout 300 python bench.py --no-extras 2>/dev/null | cut -c1-200
DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_base.so timeout 300 python bench.py --no-extras 2>/dev/null | cut -c1-200
