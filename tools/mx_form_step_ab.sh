#!/bin/bash
# The whole matrix-path step on the two MFMA shapes, alternating (diagnostics build): tools/mx_form_step_ab.sh [rounds]
R=${1:-4}
cd $GRAFT_REPO_ROOT
export VSLAM_LIBRARY=$GRAFT_REPO_ROOT/visualslam_amd/lib/libvslam_diag.so
for i in $(seq $R); do for f in 32 16; do
  VSLAM_MX_FORM=$f python3 bench.py --matrix-path 1 --cpu-sample 0 --modes 0 --mx 0 --cxx-host 0 --live-traffic 0 --steps 20 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('form $f: %.0f frames/s %.3f ms per step (hooked pass %.3f)' % (d['value'], d['ms_per_step'], d['roofline']['hooked_pass_ms_per_step']))"
done; done
