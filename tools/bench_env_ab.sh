#!/bin/bash
# bench.py's main step under settings of one environment variable on one box: tools/bench_env_ab.sh "<bench args>" VAR v1 v2 ...  ("-" = unset)
ARGS=$1; VAR=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "-" ]; then echo -n "$VAR unset  "; python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --modes 0 --mx 0 --cxx-host 0 --live-traffic 0 $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f f/s  %.2f ms' % (d['value'], d['ms_per_step']))"
  else echo -n "$VAR=$v  "; env $VAR=$v python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --modes 0 --mx 0 --cxx-host 0 --live-traffic 0 $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f f/s  %.2f ms' % (d['value'], d['ms_per_step']))"; fi
done
