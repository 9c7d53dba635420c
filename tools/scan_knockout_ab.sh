#!/bin/bash
# Upper bound of VERDICT r5 item 4 (folding the lattice scan of octaves 0-1 into k_pyr_octave): the default step with those two
# k_extrema_w3 launches LEFT OUT of the timed calls (diagnostics build, VSLAM_DIAG_SKIP_SCAN=3,<warm-up calls>: the warm-up
# calls run them, so the list kernels compact the same flag words).  A fold can only win less than this: it removes the
# launches' 43 MB per frame of DoG re-reads but has to do the min / max work inside the octave kernel.
#   bash tools/scan_knockout_ab.sh [rounds]      (on the GPU box, via gpurun)
R=${1:-3}
cd /tmp && export TMPDIR=/tmp
export VSLAM_LIBRARY=$GRAFT_REPO_ROOT/visualslam_amd/lib/libvslam_diag.so
ARGS="--cpu-sample 0 --modes 0 --mx 0 --cxx-host 0 --live-traffic 0 --steps 10 --warmup 6"
line() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print('%-26s %8.0f frames/s  %6.2f ms/step   k_pyr_octave %.3f ms/launch' % ('$1', d['value'], d['ms_per_step'], r['avg_launch_ms']))"; }
for i in $(seq $R); do
  python3 $GRAFT_REPO_ROOT/bench.py $ARGS 2>/dev/null | line "scans in (default)"
  VSLAM_DIAG_SKIP_SCAN=1,6 python3 $GRAFT_REPO_ROOT/bench.py $ARGS 2>/dev/null | line "octave 0 scan out"
  VSLAM_DIAG_SKIP_SCAN=3,6 python3 $GRAFT_REPO_ROOT/bench.py $ARGS 2>/dev/null | line "octaves 0-1 scans out"
done
