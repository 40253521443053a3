#!/bin/bash
# Round 3, call 28: secondary measurements for the record -- the scene model's TRAINING step at 512^2 (BASELINE configs[4] shape:
# L = 16,386 tokens, P = 1,048,578 Gaussians; 2 samples / GPU, 4 rendered views), and the 256^2 training step with the bf16 gradient
# exchange issued on RCCL (a process group of one rank).
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03g
mkdir -p $out
cd $R
timeout 500 python bench.py --mode train --res 512 --train-batch 2 --train-views 4 --steps 3 --warmup 1 > $out/r03_train_512.json 2> $out/r03_train_512.err; cut -c1-1300 $out/r03_train_512.json; tail -3 $out/r03_train_512.err
timeout 300 python bench.py --mode train --steps 4 --warmup 2 --force-dist --grad-exchange bf16 > $out/r03_train_rccl_world1_bf16.json 2>/dev/null; cut -c1-330 $out/r03_train_rccl_world1_bf16.json
