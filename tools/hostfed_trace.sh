#!/bin/bash
# Kernel + memory-copy trace of the host-fed C++ pipeline (no counters): tools/hostfed_trace.sh <tag> [env...]
R=${1:-hf}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr -o run -- $GRAFT_REPO_ROOT/visualslam_amd/bin/Stream --mode hostfed --batches 14 --warmup 6 > $OUT/stream.log 2>&1
tail -1 $OUT/stream.log | cut -c1-400
cd $GRAFT_REPO_ROOT
python3 tools/hostfed_timeline.py $OUT/tr
