#!/bin/bash
# Round 3, call 35: FusedAdamW as a torch.optim.Optimizer on the GPU (parity tests + one training bench).
set -u
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_optim.py tests/test_rccl_world1_gpu.py -m gpu -q 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3
timeout 300 python bench.py --mode train --steps 4 --warmup 2 2>/dev/null | cut -c1-330
