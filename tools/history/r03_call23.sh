#!/bin/bash
# Round 3, call 23: the full GPU suite + the round's evidence on the kernels of this commit (contract bench, kernel statistics of the
# bench step and the training step, PMC traffic, bench again with `roofline.traffic` from the fresh PMC pass).
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03w
mkdir -p $out
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -6 > $out/r03_final_pytest_gpu.txt; cat $out/r03_final_pytest_gpu.txt
python __graft_entry__.py --smoke > $out/r03_final_smoke.txt 2>&1; tail -2 $out/r03_final_smoke.txt
prof() {   # name, command...
    local name=$1; shift
    PROF_LINES=40 timeout 400 tools/prof.sh r03w_$name -- "$@" > /dev/null
    cp gpurun_out/r03w_$name/kernel_stats.txt $out/r03_final_${name}_kernel_stats.txt
}
prof bench python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline
head -9 $out/r03_final_bench_kernel_stats.txt | cut -c1-150
prof train_step python $R/bench.py --mode train --steps 3 --warmup 1
head -8 $out/r03_final_train_step_kernel_stats.txt | cut -c1-150
timeout 600 python tools/pmc_traffic.py > $out/r03_final_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json $out/pmc_traffic.json; cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
timeout 400 python bench.py > $out/r03_final_bench.json 2> $out/r03_final_bench.err; cut -c1-700 $out/r03_final_bench.json
