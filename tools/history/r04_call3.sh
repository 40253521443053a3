#!/bin/bash
# round 4, GPU call 3: diagnose the two failures of call 2, the deterministic raster backward (tests + A/B), the full bench, bench A/Bs
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c3; mkdir -p $out; cd $R
timeout 600 python -m pytest tests/test_graph_gpu.py -m gpu -q 2>&1 | tail -40 > $out/graph_tests.txt; cat $out/graph_tests.txt | tail -25
timeout 600 python -m pytest tests/test_ref_callers.py -m gpu -q 2>&1 | tail -30 > $out/ref_callers.txt; cat $out/ref_callers.txt | tail -12
timeout 900 python -m pytest tests/test_raster_backward_gpu.py tests/test_raster_forward_gpu.py tests/test_ref_glue_gpu.py tests/test_raster_ref_gpu.py -m gpu -q 2>&1 | tail -15 > $out/raster_tests.txt; cat $out/raster_tests.txt
timeout 600 python tools/raster_det_ab.py > $out/raster_det_ab.txt 2>&1; cat $out/raster_det_ab.txt
timeout 700 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-2500 $out/bench.json; tail -12 $out/bench.err
for v in "--graph 1 --preheat-s 2" "--graph 0 --preheat-s 2" "--graph 1 --preheat-s 0" "--graph 0 --preheat-s 0" "--graph 1 --preheat-s 2" "--graph 0 --preheat-s 0"; do
  echo "== $v" >> $out/bench_ab.txt
  timeout 200 python bench.py --no-extras --no-cpu-baseline $v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_us'], json.dumps(d['timed_region']))" >> $out/bench_ab.txt
done
cat $out/bench_ab.txt
