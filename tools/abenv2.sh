#!/bin/bash
# usage: abenv2.sh rounds VAR -- bench args : the same bench with VAR unset and VAR=1, alternately
R=$1; shift; VAR=$1; shift; shift
cd $GRAFT_REPO_ROOT
P="import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), {k:round(m['frames_per_sec']) for k,m in (d.get('modes') or {}).items()})"
for i in $(seq $R); do
  unset $VAR; echo -n "unset  "; python bench.py --cpu-sample 0 --steps 10 "$@" | python -c "$P"
  export $VAR=1; echo -n "$VAR=1 "; python bench.py --cpu-sample 0 --steps 10 "$@" | python -c "$P"
done
