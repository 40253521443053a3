#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 21): A/B of the forward's prologue fusions (two launches fewer: the timestep sinusoid inside the first Linear,
# the learned tokens' rows read by the input LayerNorm) against the previous commit's dit_forward.hip + dit_elementwise.hip
# (libdgs_hip_base.so), alternating inside one call.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
rm -f $out/prologue_fusion_ab.txt
for rep in 1 2 3 4; do for lib in libdgs_hip_base.so libdgs_hip.so; do
  DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/$lib timeout 300 python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib rep $rep ms/step', d['ms_per_step'], 'attention us', d['roofline']['avg_launch_us'])" >> $out/prologue_fusion_ab.txt
done; done
cat $out/prologue_fusion_ab.txt
