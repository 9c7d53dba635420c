#!/bin/bash
# Kernel trace of the default bench step with and without the matrix path (run via gpurun): tools/mx_prof.sh <tag>
set -e
R=${1:-mx}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
ARGS="--cpu-sample 0 --modes 0 --live-traffic 0 --cxx-host 0 --mx 0 --steps 8"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dot -o run -- python3 $B $ARGS > $OUT/prof_dot.log 2>&1

rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_mx -o run -- python3 $B $ARGS --matrix-path 1 > $OUT/prof_mx.log 2>&1
cd $GRAFT_REPO_ROOT
for d in prof_dot prof_mx; do echo "== $d"; f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); cut -d, -f1-5 $f | head -14; done
tail -1 $OUT/prof_dot.log | cut -c1-200; tail -1 $OUT/prof_mx.log | cut -c1-200
