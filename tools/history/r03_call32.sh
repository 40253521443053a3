#!/bin/bash
# Round 3, call 32: the adaLN GEMV of blocks 1.. on a stream of the library's own beside block 0 -- DiT GPU tests, the contract bench's
# timed region with DGS_DIT_SIDE_STREAM=0 / 1 alternating on one box, kernel statistics with it.
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03k
mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_dit_gpu.py tests/test_smoke_c1.py tests/test_sampler.py -m gpu -q -x 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -3 > $out/pytest_dit.txt; cat $out/pytest_dit.txt
for i in 1 2; do
  for s in 0 1; do
    DGS_DIT_SIDE_STREAM=$s timeout 300 python bench.py --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side stream $s: %.3f ms/step  %.1f renders/s  attention %.1f us' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_us']))" | tee -a $out/side_stream_ab.txt
  done
done
PROF_LINES=40 timeout 400 tools/prof.sh r03k_bench -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > /dev/null
cp gpurun_out/r03k_bench/kernel_stats.txt $out/bench_side_stream_kernel_stats.txt; grep -n "rowlinear\|attention_fwd\|gemm_sliced_kernel<4" $out/bench_side_stream_kernel_stats.txt | cut -c1-150
