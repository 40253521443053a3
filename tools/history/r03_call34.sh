#!/bin/bash
# Round 3, call 34: the end-of-round check on the library built from scratch by __graft_entry__.build(): GPU suite, smoke, contract bench.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03m
mkdir -p $out
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 > $out/r03_end_pytest_gpu.txt; cat $out/r03_end_pytest_gpu.txt
python __graft_entry__.py --smoke 2>&1 | tail -2 | tee $out/r03_end_smoke.txt
timeout 400 python bench.py > $out/r03_end_bench.json 2> $out/r03_end_bench.err; cut -c1-420 $out/r03_end_bench.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03m/r03_end_bench.json"))
print("roofline", d["roofline"])
print("train", d["train_step"]["ms_per_step"], d["train_step"]["optimizer"], "scene512", d["scene_512"]["ms_per_step"], "loop", d["sampling_loop_30_steps"]["ms_per_loop"])
PY
