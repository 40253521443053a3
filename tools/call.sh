#!/bin/bash
# One gpurun call's worth of work; rewritten per call during development.
# This form (round 6, call 36): wall time of the default `python bench.py` (the driver's N = 1 line) on a fresh box, and of smoke().
set -u
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/call
mkdir -p $out
cd $R
SECONDS=0; python __graft_entry__.py --smoke > $out/smoke.txt 2>&1; echo "smoke wall s: $SECONDS" > $out/bench_wall.txt
SECONDS=0; python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default bench.py wall s: $SECONDS" >> $out/bench_wall.txt
SECONDS=0; python bench.py > $out/bench_default2.json 2>> $out/bench_default.err; echo "default bench.py wall s (second run): $SECONDS" >> $out/bench_wall.txt
cat $out/bench_wall.txt; tail -2 $out/smoke.txt; cut -c1-200 $out/bench_default.json; cut -c1-200 $out/bench_default2.json
