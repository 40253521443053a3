#!/bin/bash
# Round-3 GPU call 1: A/B of the two prepared GEMM patches against the round-2 library, the training-shape gradient parity test on the
# current backward (per-tensor errors dumped), run-to-run determinism before the reductions change.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03a
mkdir -p $out
cd $R
timeout 300 python tools/gemm_ab.py open-diffusiongs_amd/lib/libdgs_hip_base.so 2>&1 | grep -v amdgpu.ids > $out/gemm_ab.txt
cat $out/gemm_ab.txt
DGS_GRAD_PARITY_DUMP=$out/grad_parity timeout 900 python -m pytest tests/test_dit_backward_gpu.py -m gpu -q -k "training_shape" 2>&1 | tail -15 > $out/grad_parity_pytest.txt
cat $out/grad_parity_pytest.txt
timeout 300 python tools/train_determinism.py 256 2>&1 | grep -v amdgpu.ids > $out/determinism_before.txt
cat $out/determinism_before.txt
