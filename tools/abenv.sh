#!/bin/bash
# usage: abenv.sh rounds VAR val1 val2 ... -- bench args
R=$1; shift; VAR=$1; shift; VALS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do VALS+=("$1"); shift; done; shift
cd $GRAFT_REPO_ROOT
for i in $(seq $R); do for v in "${VALS[@]}"; do
  export $VAR=$v
  python bench.py --cpu-sample 0 --steps 10 "$@" | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', round(d['value'],1), round(d['roofline']['avg_launch_ms'],3), {k:round(m['frames_per_sec']) for k,m in (d.get('modes') or {}).items()})"
done; done
