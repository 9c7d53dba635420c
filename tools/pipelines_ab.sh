#!/bin/bash
# Two batches in flight (BatchDetector::Options::pipelines = 2: consecutive batches on two contexts / streams, each following
# the other past its octave-0 kernels) against one, C++ host, device-resident, default and matrix path.  On the GPU box.
R=${1:-2}
cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-34s %8.0f frames/s  %6.2f ms/batch' % (sys.argv[1], d['frames_per_sec'], d['ms_per_batch']))" "$1"; }
for i in $(seq $R); do
  for mx in 0 1; do for p in 1 2; do
    VSLAM_MX=$mx ./visualslam_amd/bin/Stream --mode device --batches 30 --warmup 6 --pipelines $p 2>/dev/null | line "matrix_path=$mx pipelines=$p"
  done; done
done
