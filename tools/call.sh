#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
export PYTHONPATH=$R/open-diffusiongs_amd:$R
timeout 400 python -m pytest tests/test_dit_gpu.py -x -q -k "attention or forward" > $out/pytest_attn.txt 2>&1; tail -3 $out/pytest_attn.txt
timeout 300 python tools/attn_tail_early_ab.py > $out/attn_tail_early_ab.txt 2>&1; cat $out/attn_tail_early_ab.txt
DGS_AMD_LIBRARY=$R/open-diffusiongs_amd/lib/libdgs_hip_instr.so DGS_ATTN_DBG=24 timeout 120 python tools/attn_timeline.py > $out/attn_timeline_early.txt 2>&1; grep "timeline\|attn dbg" $out/attn_timeline_early.txt | head -6 | cut -c1-400
for late in 1 0 1 0; do
  DGS_ATTN_TAIL_LATE=$late timeout 200 python bench.py --no-extras --steps 40 --warmup 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readlines()[-1]); print('late=$late', j['ms_per_step'], j['roofline']['avg_launch_us'])"
done 2>&1 | tee $out/bench_ab.txt
