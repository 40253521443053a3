#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh NAME -- command...   -> gpurun_out/NAME/{kernel_stats.txt,run.log}
set -u
name=$1; shift; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/prof -- "$@" > $out/run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_stats.py $(find $out/prof -name "*.db" | head -1) $out/kernel_stats.txt > /dev/null
head -${PROF_LINES:-28} $out/kernel_stats.txt
rm -rf $out/prof
