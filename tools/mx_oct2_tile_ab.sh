#!/bin/bash
# A/B build: the matrix path's octave-2 kernel on 256 x 128 tiles (two 128-column strips side by side, eight waves) against 128 x 128.
cd $GRAFT_REPO_ROOT
export VSLAM_MX=1
for i in 1 2; do
for lib in visualslam_amd/lib/libvslam.so visualslam_amd/lib/ab/mx_o2w.so; do
  echo -n "$lib octaves 0-2 alone: "
  VSLAM_LIBRARY=$GRAFT_REPO_ROOT/$lib python3 tools/mx_alone.py --octaves 3 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f ms' % d['octave_kernel_ms_per_step'])"
  echo -n "$lib whole step: "
  VSLAM_LIBRARY=$GRAFT_REPO_ROOT/$lib python3 bench.py --matrix-path 1 --cpu-sample 0 --modes 0 --mx 0 --cxx-host 0 --live-traffic 0 --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.0f frames/s %.3f ms' % (d['value'], d['ms_per_step']))"
done; done
VSLAM_LIBRARY=$GRAFT_REPO_ROOT/visualslam_amd/lib/ab/mx_o2w.so timeout -k 10 300 python3 -m pytest tests/test_gpu_batch.py -m gpu -x -q -k "mx and (1080p or random_shapes or ragged)" 2>&1 | tail -2
