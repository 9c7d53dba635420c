#!/bin/bash
# Same-box A/B of two builds of libvslam.so (box-to-box spread is +-2.5 %, larger than most
# kernel tweaks).  Build the variants first, e.g.
#   make -C visualslam_amd/csrc                       && cp visualslam_amd/lib/libvslam.so visualslam_amd/lib/libvslam_a.so
#   <edit> && make -C visualslam_amd/csrc             && cp visualslam_amd/lib/libvslam.so visualslam_amd/lib/libvslam_b.so
# then on the GPU box:  bash tools/ab.sh [rounds] [bench args...]
R=${1:-4}; shift
cd $GRAFT_REPO_ROOT
for i in $(seq $R); do
  for v in a b; do
    VSLAM_LIBRARY=$GRAFT_REPO_ROOT/visualslam_amd/lib/libvslam_$v.so python bench.py --cpu-sample 0 --steps 10 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['roofline']['avg_launch_ms'],3), {k:round(m['frames_per_sec']) for k,m in (d.get('modes') or {}).items()})"
  done
done
