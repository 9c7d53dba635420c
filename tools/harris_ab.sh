#!/bin/bash
# Harris-only batch timing (BASELINE config 2 shape, 256 frames): bash tools/harris_ab.sh [lib ...]
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  for i in 1 2 3; do
    VSLAM_LIBRARY=$GRAFT_REPO_ROOT/$lib python bench.py --cpu-sample 0 --steps 10 --octaves 0 --kernel k_harris_strip | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value'],1), d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  done
done
