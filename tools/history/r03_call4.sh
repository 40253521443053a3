#!/bin/bash
# Round-3 GPU call 4: attention experiments (A/B of four builds against the product kernel), attention phase stamps, vendor-library GEMM
# reference point, the GPU suite on the per-block adaLN backward, training step.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03d
mkdir -p $out
cd $R
for v in setprio wide tailfirst all; do
  echo "=== $v" >> $out/attn_ab.txt
  timeout 200 python tools/attn_ab.py open-diffusiongs_amd/lib/libdgs_hip.so open-diffusiongs_amd/lib/libdgs_hip_$v.so 2>&1 | grep -v amdgpu.ids >> $out/attn_ab.txt
done
cat $out/attn_ab.txt
DGS_ATTN_DBG=12 timeout 120 python tools/attn_bench.py 2 2>&1 | grep "attn dbg" | tail -4 > $out/attn_phases.txt
cat $out/attn_phases.txt
timeout 200 python tools/gemm_library_ref.py 2>&1 | grep -v amdgpu.ids > $out/gemm_library_ref.txt
cat $out/gemm_library_ref.txt
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > $out/pytest_gpu.txt
cat $out/pytest_gpu.txt
timeout 300 python bench.py --mode train --steps 5 --warmup 2 > $out/train_bench.json 2> $out/train_bench.err
cut -c1-300 $out/train_bench.json
