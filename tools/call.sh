#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 44, EXPERIMENT): the K-group 128 x 128 kernel with its DMA pieces spread behind the MFMAs of substep 0
# (lib/libdgs_hip_var.so, -DDGS_KG_SPREAD) against the burst at the top of the iteration (the product library).
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
rm -f $out/kg_spread_ab.txt
for rep in 1 2 3 4; do for lib in libdgs_hip.so libdgs_hip_var.so; do
  DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/$lib python bench.py --steps 40 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib rep $rep ms/step', d['ms_per_step'], 'attention us', d['roofline']['avg_launch_us'])" >> $out/kg_spread_ab.txt
done; done
for lib in libdgs_hip.so libdgs_hip_var.so; do
  echo "== $lib" >> $out/kg_spread_ab.txt
  DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/$lib GEMM_CASES=fc2,proj timeout 120 python tools/gemm_check.py 0 2>&1 | grep -v amdgpu.ids >> $out/kg_spread_ab.txt
done
cat $out/kg_spread_ab.txt
