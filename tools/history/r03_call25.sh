#!/bin/bash
# Round 3, call 25 (24 re-run with the explicit refresh for torch optimizers): the fused AdamW + weight refresh launch: GPU parity vs torch.optim.AdamW, the training step with it and with
# torch's optimizer + torch-copy refresh (same box), kernel statistics of the step with it.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03y
mkdir -p $out
cd $R
timeout 600 python -m pytest tests/test_optim.py -m gpu -q -s 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -5 > $out/pytest_optim.txt; cat $out/pytest_optim.txt
timeout 300 python bench.py --mode train --steps 5 --warmup 2 --optimizer torch > $out/train_torch_adamw.json 2>/dev/null; cut -c1-330 $out/train_torch_adamw.json
timeout 300 python bench.py --mode train --steps 5 --warmup 2 --optimizer fused > $out/train_fused_adamw.json 2>/dev/null; cut -c1-330 $out/train_fused_adamw.json
timeout 300 python bench.py --mode train --steps 5 --warmup 2 --optimizer torch 2>/dev/null | cut -c1-330
timeout 300 python bench.py --mode train --steps 5 --warmup 2 --optimizer fused 2>/dev/null | cut -c1-330
PROF_LINES=40 timeout 400 tools/prof.sh r03y_train_step -- python $R/bench.py --mode train --steps 3 --warmup 1 > /dev/null
cp gpurun_out/r03y_train_step/kernel_stats.txt $out/train_step_fused_adamw_kernel_stats.txt; grep -n "adamw\|multi_tensor\|transpose_kernel\|copyBuffer\|elementwise_kernel_manual" $out/train_step_fused_adamw_kernel_stats.txt | cut -c1-160
