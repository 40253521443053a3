#!/bin/bash
# Round 3, call 17: the round's evidence on HEAD -- RCCL on one GPU (world of one, every collective issued), the contract bench,
# the driver's torchrun launch form, the training step with the collectives issued, kernel statistics, PMC traffic.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03q
mkdir -p $out
cd $R
timeout 500 python -m pytest tests/test_rccl_world1_gpu.py -m gpu -q -s 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -8 > $out/r03_rccl_world1_pytest.txt; cat $out/r03_rccl_world1_pytest.txt | cut -c1-1200
timeout 400 python bench.py > $out/r03_bench.json 2> $out/r03_bench.err; cut -c1-400 $out/r03_bench.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --force-dist > $out/r03_bench_torchrun_rccl_world1.json 2> $out/r03_bench_torchrun.err; cut -c1-300 $out/r03_bench_torchrun_rccl_world1.json; tail -3 $out/r03_bench_torchrun.err
timeout 300 python bench.py --mode train --steps 4 --warmup 2 --force-dist > $out/r03_train_rccl_world1.json 2> $out/r03_train_rccl.err; cut -c1-1500 $out/r03_train_rccl_world1.json; tail -3 $out/r03_train_rccl.err
timeout 300 python bench.py --mode train --steps 4 --warmup 2 > $out/r03_train.json 2> $out/r03_train.err; cut -c1-300 $out/r03_train.json
prof() {   # name, command...
    local name=$1; shift
    PROF_LINES=40 timeout 400 tools/prof.sh r03q_$name -- "$@" > /dev/null
    cp gpurun_out/r03q_$name/kernel_stats.txt $out/r03_${name}_kernel_stats.txt
}
prof bench python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline
head -12 $out/r03_bench_kernel_stats.txt | cut -c1-150
prof train_step python $R/bench.py --mode train --steps 3 --warmup 1
head -14 $out/r03_train_step_kernel_stats.txt | cut -c1-150
timeout 600 python tools/pmc_traffic.py > $out/r03_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json $out/pmc_traffic.json
tail -5 $out/r03_pmc_traffic.log | cut -c1-300
