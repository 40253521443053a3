#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
for c in 0 2; do
  DGS_TAIL_CHAIN=$c timeout 300 python bench.py --no-extras --no-cpu-baseline > $out/bench_chain${c}.json 2>> $out/bench.err
  python - <<PY
import json
d=json.load(open("$out/bench_chain${c}.json"))
print("chain $c:", d["ms_per_step"], "ms", d["roofline"]["avg_launch_us"], "us attention")
PY
done
DGS_TAIL_CHAIN=2 PROF_LINES=12 tools/prof.sh chain2 -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --graph 0 > /dev/null 2>&1
head -12 gpurun_out/chain2/kernel_stats.txt
DGS_TAIL_CHAIN=0 timeout 600 python -m pytest tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -3
DGS_TAIL_CHAIN=2 timeout 600 python -m pytest tests/test_graph_gpu.py -m gpu -x -q 2>&1 | grep -v "^  File" | tail -6
