#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 300 python tools/ln_ab.py open-diffusiongs_amd/lib/libdgs_hip_base.so > $out/ln_ab.txt 2>&1; grep -v amdgpu $out/ln_ab.txt | cut -c1-200
for i in 1 2; do for r in 1 2; do DGS_LN_RPW=$r timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rpw $r:', d['ms_per_step'], d['kernel_families']['layernorm'])"; done; done
