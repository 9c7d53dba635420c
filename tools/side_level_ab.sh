#!/bin/bash
# The batched step with the side streams pinned to each level (0 yielding, 1 same priority = default, 2 none: every kernel on the
# context's stream in order), default and matrix path, same box.
cd $GRAFT_REPO_ROOT
for i in 1 2; do for mx in 0 1; do for lv in 1 2 0; do
  python3 bench.py --matrix-path $mx --side-level $lv --cpu-sample 0 --modes 0 --mx 0 --cxx-host 0 --live-traffic 0 --steps 10 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('matrix_path=%d side level %d: %.0f frames/s %.3f ms' % ($mx, $lv, d['value'], d['ms_per_step']))"
done; done; done
