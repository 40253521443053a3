#!/bin/bash
# round 4, GPU call 6: attention tail queries on a second stream -- parity, graph capture with the fork / join, A/B; the PSNR question
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/c6; mkdir -p $out; cd $R
timeout 900 python -m pytest tests/test_dit_gpu.py tests/test_graph_gpu.py tests/test_smoke_c1.py tests/test_ref_callers.py -m gpu -q -x 2>&1 | tail -15 > $out/tests.txt; tail -8 $out/tests.txt
if grep -q "failed\|error" $out/tests.txt; then export DGS_ATTN_TAIL_STREAM=0; echo "SPLIT TAIL DISABLED FOR THE REST" ; fi
timeout 400 python tools/history/r04_psnr_debug.py > $out/psnr_debug.txt 2>&1; grep -v amdgpu.ids $out/psnr_debug.txt | cut -c1-330
for rep in 1 2; do for v in "1 1" "0 1" "1 0" "0 0"; do
  set -- $v
  echo "== DGS_ATTN_TAIL_STREAM=$1 --graph $2" >> $out/bench_ab.txt
  DGS_ATTN_TAIL_STREAM=$1 timeout 200 python bench.py --no-extras --no-cpu-baseline --graph $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['timed_region']['host_enqueue_ms_per_step'], d['timed_region']['gpu_ms_per_step'])" >> $out/bench_ab.txt
done; done
cat $out/bench_ab.txt
timeout 700 python bench.py > $out/bench.json 2> $out/bench.err; cut -c1-1200 $out/bench.json; tail -5 $out/bench.err
PROF_LINES=12 tools/prof.sh c6_prof_bench -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --graph 0
