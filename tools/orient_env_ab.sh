#!/bin/bash
# the orientation stage under several settings of one environment variable, on one box: tools/orient_env_ab.sh VAR v1 v2 ...
VAR=$1; shift
for v in "$@"; do
  echo -n "$VAR=$v  "; env $VAR=$v python3 tools/orient_alone.py $OA_ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step_ms_orient %.2f  kernel_ms %.2f  localize %.2f' % (d['step_ms_orient'], d['kernel_ms_per_step'], d['step_ms_localize']))"
done
