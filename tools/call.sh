#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 27): the soak of the round's final tree (tools/step_soak.py 40) + one more contract bench line on this box.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 600 python tools/step_soak.py 40 > $out/step_soak.txt 2>&1
grep "soak" $out/step_soak.txt
python bench.py --no-extras --no-cpu-baseline > $out/bench_line.json 2> $out/bench_line.err
cut -c1-250 $out/bench_line.json
