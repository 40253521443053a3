#!/bin/bash
# The round's measurement artefacts in one GPU call (from the repo root on the GPU box): everything lands under gpurun_out/final/.
#   tools/final_run.sh [tag]        then copy gpurun_out/final/* into profiles/ (names carry the tag)
set -u
tag=${1:-r06}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/final
mkdir -p $out
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $out/${tag}_pytest_gpu.txt
python __graft_entry__.py --smoke > $out/${tag}_smoke.txt 2>&1
# the contract bench before the profiler / PMC passes
# (`roofline.traffic` then comes from the profiles/pmc_traffic.json of the previous call: null if the kernel sources changed since)
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 600 $out/${tag}_pytest_gpu.txt; tail -2 $out/${tag}_smoke.txt; cut -c1-600 $out/${tag}_bench.json
prof() {   # name, command...
    local name=$1; shift
    PROF_LINES=40 tools/prof.sh final_$name -- "$@" > /dev/null
    cp gpurun_out/final_$name/kernel_stats.txt $out/${tag}_${name}_kernel_stats.txt
    grep -E "ms/call|views/s|renders|samples" gpurun_out/final_$name/run.log | tail -6 > $out/${tag}_${name}.log
}
prof bench python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --graph 0
prof train_step python $R/bench.py --mode train --steps 3 --warmup 1
prof train_scene512 python $R/bench.py --mode train-scene --scene-recompute off --steps 1 --warmup 1      # save-all: the faster mode (the headline of train_step_scene_512)
prof train_scene512_recompute python $R/bench.py --mode train-scene --scene-recompute on --steps 1 --warmup 1
prof raster256_trained python $R/tools/raster_microbench.py --res 256 --regime trained
prof raster256_init python $R/tools/raster_microbench.py --res 256 --regime init
python tools/pmc_traffic.py > $out/${tag}_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json $out/pmc_traffic.json
cp gpurun_out/dit_pmc.txt $out/${tag}_dit_pmc.txt
python tools/raster_det_ab.py > $out/${tag}_raster_deterministic_ab.txt 2>&1
python tools/raster_grad_error.py > $out/${tag}_raster_grad_error_final.txt 2>&1
tools/ubench/exp_ulp_bench > $out/${tag}_exp_ulp_final.txt 2>&1
# the training step under the driver's launch line with forced (world-1 RCCL) collectives: host lead in the backward, verify-wait split
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --mode train --steps 4 --warmup 2 --force-dist --no-cpu-baseline > $out/${tag}_train_rccl_world1.json 2>> $out/${tag}_bench.err
# the driver's multi-GPU launch line with the one GPU there is: RCCL process group of one rank, every collective issued
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 2 --force-dist --no-cpu-baseline > $out/${tag}_bench_torchrun_rccl_world1.json 2>> $out/${tag}_bench.err
cut -c1-400 $out/${tag}_bench_torchrun_rccl_world1.json
# second contract line, now with `traffic` (the PMC JSON above carries this tree's source hashes)
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py --no-extras --no-cpu-baseline > $out/${tag}_bench_after_pmc.json 2>> $out/${tag}_bench.err
cut -c1-900 $out/${tag}_bench_after_pmc.json
