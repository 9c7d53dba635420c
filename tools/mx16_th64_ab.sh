#!/bin/bash
# A/B build: the 16 x 16 x 64 octave-0 kernel on 128 x 64 tiles of four waves (three workgroups per CU) against 128 x 128 / eight waves.
cd $GRAFT_REPO_ROOT
export VSLAM_MX=1 VSLAM_MX_FORM=16
for i in 1 2; do
for lib in visualslam_amd/lib/libvslam_diag.so visualslam_amd/lib/ab/mx16_th64.so; do
  echo -n "$lib octave 0 alone: "
  VSLAM_LIBRARY=$GRAFT_REPO_ROOT/$lib python3 tools/mx_alone.py --octaves 1 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f ms' % d['octave_kernel_ms_per_step'])"
  echo -n "$lib whole step: "
  VSLAM_LIBRARY=$GRAFT_REPO_ROOT/$lib python3 bench.py --matrix-path 1 --cpu-sample 0 --modes 0 --mx 0 --cxx-host 0 --live-traffic 0 --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.0f frames/s %.3f ms' % (d['value'], d['ms_per_step']))"
done; done
VSLAM_LIBRARY=$GRAFT_REPO_ROOT/visualslam_amd/lib/ab/mx16_th64.so timeout -k 10 300 python3 __graft_entry__.py smoke 2>&1 | tail -1
