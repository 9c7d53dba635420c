#!/bin/bash
# tools/orient_alone.py under every library in lib/ab/ and the tree's own, alternating, on one box: tools/lib_ab.sh [orient_alone args]
for round in 1 2; do
  for L in $(ls $GRAFT_REPO_ROOT/visualslam_amd/lib/ab/*.so 2>/dev/null) $GRAFT_REPO_ROOT/visualslam_amd/lib/libvslam.so; do
    echo -n "$(basename $L .so)  "; VSLAM_LIBRARY=$L python3 tools/orient_alone.py "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step %.2f  kernel(%s) %.2f  localize %.2f' % (d['step_ms_orient'], d['kernel'], d['kernel_ms_per_step'], d['step_ms_localize']))"
  done
done
