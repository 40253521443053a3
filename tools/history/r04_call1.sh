#!/bin/bash
# round 4, GPU call 1: new parity tests first, then the whole GPU suite, the contract bench (graph + eager), one kernel-stats profile
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c1; mkdir -p $out; cd $R
timeout 900 python -m pytest tests/test_dit_backward_gpu.py tests/test_raster_backward_gpu.py tests/test_ref_callers.py tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -25 > $out/new_tests.txt
cat $out/new_tests.txt
DGS_GRAD_PARITY_DUMP=$out/grad_parity timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $out/pytest_gpu.txt
cat $out/pytest_gpu.txt
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
cut -c1-1500 $out/bench.json; tail -5 $out/bench.err
timeout 300 python bench.py --graph 0 --no-extras --no-cpu-baseline > $out/bench_eager.json 2>> $out/bench.err
cut -c1-1200 $out/bench_eager.json
PROF_LINES=45 tools/prof.sh c1_prof_bench -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline
