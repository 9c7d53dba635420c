#!/bin/bash
# Full GPU test-suite + the default bench line (run via gpurun): bash tools/gpu_check.sh <tag>
R=${1:-check}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; rc=$?
tail -4 $OUT/gputest.log
[ $rc = 0 ] || exit $rc
timeout -k 10 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "roofline.frac", d["roofline"]["frac"])
print("mx_path", json.dumps(d["mx_path"])[:900])
print("wall", d["bench_wall_s"])
PY
