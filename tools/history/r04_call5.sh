#!/bin/bash
# round 4, GPU call 5: memset / memcpy nodes replaced by library kernels -> graph debug script, graph tests, det A/B, full suite, bench
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c5; mkdir -p $out; cd $R
timeout 300 python tools/history/r04_graph_debug.py > $out/graph_debug.txt 2>&1; grep -v amdgpu.ids $out/graph_debug.txt | cut -c1-260
timeout 600 python -m pytest tests/test_graph_gpu.py -m gpu -q 2>&1 | tail -30 > $out/graph_tests.txt; tail -8 $out/graph_tests.txt
if grep -q "failed\|error" $out/graph_tests.txt; then export DGS_GRAPH=0; DESEL="--deselect tests/test_graph_gpu.py"; else DESEL=""; fi
timeout 300 python tools/raster_det_ab.py > $out/raster_det_ab.txt 2>&1; grep -v amdgpu.ids $out/raster_det_ab.txt
DGS_GRAD_PARITY_DUMP=$out/grad_parity timeout 1500 python -m pytest tests -m gpu -q $DESEL 2>&1 | tail -15 > $out/pytest_gpu.txt; tail -6 $out/pytest_gpu.txt
timeout 700 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-3000 $out/bench.json; tail -8 $out/bench.err
for v in "--graph 1" "--graph 0" "--graph 1" "--graph 0"; do
  echo "== $v" >> $out/bench_ab.txt
  timeout 200 python bench.py --no-extras --no-cpu-baseline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_us'], json.dumps(d['timed_region']))" >> $out/bench_ab.txt
done
cat $out/bench_ab.txt
