#!/bin/bash
# Stream (C++ host) at the bench workload: device-resident and host-fed.  One rank, RCCL.
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=${MASTER_PORT:-29611}
B=visualslam_amd/bin/Stream
$B --mode device --batches 30 --warmup 6 | tail -n 1
$B --mode hostfed --batches 40 --warmup 6 | tail -n 1
