#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 18): the forward's conditioning chain on a side stream beside the token chain (DGS_DIT_PROLOGUE_OVERLAP=0: off):
# DiT / graph / sampler GPU tests, then the contract step A/B, alternating, graph replays and eager.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 1500 python -m pytest tests/test_dit_gpu.py tests/test_graph_gpu.py tests/test_smoke_c1.py tests/test_ref_callers.py tests/test_sampler.py -x -q -m gpu > $out/pytest_dit_gpu.txt 2>&1; tail -3 $out/pytest_dit_gpu.txt
rm -f $out/prologue_overlap_ab.txt
for rep in 1 2 3; do for v in 0 1; do
  DGS_DIT_PROLOGUE_OVERLAP=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$v rep $rep graph ms/step', d['ms_per_step'], 'attention us', d['roofline']['avg_launch_us'])" >> $out/prologue_overlap_ab.txt
done; done
for v in 0 1; do
  DGS_DIT_PROLOGUE_OVERLAP=$v timeout 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline --graph 0 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('overlap=$v eager ms/step', d['ms_per_step'])" >> $out/prologue_overlap_ab.txt
done
cat $out/prologue_overlap_ab.txt
