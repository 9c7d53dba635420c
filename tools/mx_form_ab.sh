#!/bin/bash
# The matrix path's octaves 0-1 on the two MFMA shapes, same box: 16 x 16 x 64 (default since round 6) against 32 x 32 x 32
# (VSLAM_MX_FORM=32, diagnostics build).  Kernels alone (pyramid-only batches: octave 0 only, octaves 0-1) and the whole step.
R=${1:-2}
cd $GRAFT_REPO_ROOT
export VSLAM_LIBRARY=$GRAFT_REPO_ROOT/visualslam_amd/lib/libvslam_diag.so VSLAM_MX=1
one() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-22s octave kernels %.3f ms per step, step %.3f ms' % (sys.argv[1], d['octave_kernel_ms_per_step'], d['step_ms']))" "$1"; }
step() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%-22s whole step %.0f frames/s %.3f ms; k_pyr_octave_mx %.3f ms per step' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * d['roofline']['launches'] / d['steps']))" "$1"; }
for i in $(seq $R); do
  for f in 16 32; do
    VSLAM_MX_FORM=$f python3 tools/mx_alone.py --octaves 1 2>/dev/null | one "form $f, octave 0"
    VSLAM_MX_FORM=$f python3 tools/mx_alone.py --octaves 2 2>/dev/null | one "form $f, octaves 0-1"
    VSLAM_MX_FORM=$f python3 bench.py --matrix-path 1 --cpu-sample 0 --modes 0 --mx 0 --cxx-host 0 --live-traffic 0 --steps 10 2>/dev/null | step "form $f"
  done
done
