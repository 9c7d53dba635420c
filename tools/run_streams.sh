#!/bin/bash
# One Stream process per GPU of this node (BASELINE config 5): camera stream r on GPU r, the {harris, dog} counts
# all-gathered through RCCL.  Rank 0 prints the job's JSON line.
#   tools/run_streams.sh 8 [--mode device|hostfed] [--frames 256] [--batches 30] ...
set -e
N=${1:-1}; shift || true
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export WORLD_SIZE=$N MASTER_ADDR=${MASTER_ADDR:-127.0.0.1} MASTER_PORT=${MASTER_PORT:-29533} HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for ((r = N - 1; r >= 0; --r)); do
    RANK=$r LOCAL_RANK=$r "$ROOT/visualslam_amd/bin/Stream" "$@" &
    pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
exit $rc
