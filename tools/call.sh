#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 600 python -m pytest tests/test_graph_gpu.py tests/test_raster_forward_gpu.py -m gpu -x -q 2>&1 | tail -2
for reg in init trained; do timeout 120 python tools/raster_microbench.py --res 256 --regime $reg 2>&1 | grep "async\|forward+backward" | cut -c1-120; done
timeout 300 python bench.py --mode train --steps 3 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['train_step']; print('train', d['ms_per_step'], 'host', d['host_enqueue_ms_per_step'])"
