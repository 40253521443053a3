#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 900 python -m pytest tests/test_raster_forward_gpu.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do for r3 in 1 0; do
  echo "radix3=$r3 init:";    DGS_RASTER_RADIX3=$r3 timeout 120 python tools/raster_microbench.py --res 256 --regime init 2>&1 | grep "async\|sync (" | cut -c1-150
done; done
DGS_RASTER_RADIX3=0 PROF_LINES=14 tools/prof.sh radix1 -- python $R/tools/raster_microbench.py --res 256 --regime init > /dev/null 2>&1
head -14 gpurun_out/radix1/kernel_stats.txt
for det in 0 1; do
  DGS_RASTER_DETERMINISTIC=$det DGS_RASTER_DETERMINISTIC_GIB=64 timeout 300 python bench.py --mode train --steps 3 --warmup 2 > $out/train_det$det.json 2> $out/train_det$det.err
  python - <<PY
import json
d=json.load(open("$out/train_det$det.json"))["train_step"]
print("det $det (64 GiB budget):", d["ms_per_step"], "ms host", d["host_enqueue_ms_per_step"], d["gpu_memory_gib"], d["raster"])
PY
done
timeout 300 python bench.py --mode train --steps 3 --warmup 2 > $out/train_default.json 2> $out/train_default.err
python - <<PY
import json
d=json.load(open("$out/train_default.json"))["train_step"]
print("default:", d["ms_per_step"], "ms host", d["host_enqueue_ms_per_step"], d["gpu_memory_gib"], d["raster"])
PY
timeout 900 python -m pytest tests/test_optim.py tests/test_graph_gpu.py -m gpu -q 2>&1 | grep -v "^  File\|^$" | tail -4 | cut -c1-200
