#!/bin/bash
# orientation-stage quick check on the GPU box: parity tests of the stage, then timing + kernel trace of a 64-frame batch
timeout -k 10 300 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -k "orient or describ or sift" 2>&1 | tail -2 || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/oq; mkdir -p $OUT
python3 tools/orient_alone.py 2>/dev/null | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o r -- python3 $GRAFT_REPO_ROOT/tools/orient_alone.py --frames 64 --steps 2 > $OUT/kt.log 2>&1
python3 - <<PY
import csv, glob
tr = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)
d = [(r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in csv.DictReader(open(tr[0])) if "k_orient_survivors" in r["Kernel_Name"]]
pk = [t for n, t in d if n.endswith("_pk")]
print("k_orient_survivors_pk per launch (64 frames) us:", [round(t) for t in pk])
PY
