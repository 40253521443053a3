#!/bin/bash
# Round-3 GPU call 2: the atomics-free backward -- run-to-run determinism, gradient parity at the training shape, GPU suite, training step.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03b
mkdir -p $out
cd $R
timeout 300 python tools/train_determinism.py 256 save 2 3 2>&1 | grep -v amdgpu.ids > $out/determinism.txt
timeout 300 python tools/train_determinism.py 256 recompute 4 2 2>&1 | grep -v amdgpu.ids >> $out/determinism.txt
cat $out/determinism.txt
DGS_GRAD_PARITY_DUMP=$out/grad_parity timeout 900 python -m pytest tests/test_dit_backward_gpu.py -m gpu -q 2>&1 | tail -15 > $out/dit_backward_pytest.txt
cat $out/dit_backward_pytest.txt
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_dit_backward_gpu.py 2>&1 | tail -8 > $out/pytest_gpu.txt
cat $out/pytest_gpu.txt
timeout 300 python bench.py --mode train --steps 5 --warmup 2 > $out/train_bench.json 2> $out/train_bench.err
cut -c1-600 $out/train_bench.json
