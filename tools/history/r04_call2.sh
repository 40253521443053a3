#!/bin/bash
# round 4, GPU call 2: graph tests alone first (a broken capture poisons its process), then the suite, bench (graph + eager), profile
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c2; mkdir -p $out; cd $R
timeout 600 python -m pytest tests/test_graph_gpu.py -m gpu -q 2>&1 | tail -30 > $out/graph_tests.txt
cat $out/graph_tests.txt
if grep -q "failed\|error" $out/graph_tests.txt; then export DGS_GRAPH=0; DESEL="--deselect tests/test_graph_gpu.py"; else DESEL=""; fi
DGS_GRAD_PARITY_DUMP=$out/grad_parity timeout 1500 python -m pytest tests -m gpu -q $DESEL 2>&1 | tail -25 > $out/pytest_gpu.txt
cat $out/pytest_gpu.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
cut -c1-1800 $out/bench.json; tail -5 $out/bench.err
timeout 300 python bench.py --graph 0 --no-extras --no-cpu-baseline > $out/bench_eager.json 2>> $out/bench.err
cut -c1-1500 $out/bench_eager.json
PROF_LINES=45 tools/prof.sh c2_prof_bench -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline
