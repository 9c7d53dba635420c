#!/bin/bash
# A/B of the orientation stage on one box: every library given (default: lib/ab/*.so, then the tree's own) on the same
# 64-frame batch under a kernel trace, three alternating rounds.  tools/orient_ab.sh [lib.so ...]
OUT=$GRAFT_REPO_ROOT/gpurun_out/oab; mkdir -p $OUT
LIBS="$@"; [ -z "$LIBS" ] && LIBS="$(ls $GRAFT_REPO_ROOT/visualslam_amd/lib/ab/*.so 2>/dev/null) $GRAFT_REPO_ROOT/visualslam_amd/lib/libvslam.so"
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  for L in $LIBS; do
    tag=$(basename $L .so)_$round
    VSLAM_LIBRARY=$L rocprofv3 --kernel-trace --output-format csv -d $OUT/$tag -o r -- python3 $GRAFT_REPO_ROOT/tools/orient_alone.py --frames 64 --steps 2 > $OUT/$tag.log 2>&1 || { tail -5 $OUT/$tag.log; exit 1; }
  done
done
python3 - <<PY
import csv, glob, collections, os
res = collections.defaultdict(list)
for d in sorted(glob.glob("$OUT/*_[12]")):
    tr = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr[0])) if "k_orient_survivors" in r["Kernel_Name"]]
    big = sorted(t)[-3:]   # the octave-0 launches of the four batches
    res[os.path.basename(d)[:-2]].append(sum(big) / len(big))
for k, v in res.items():
    print(f"{k:24s} octave-0 launch us per round: {[round(x) for x in v]}  mean {sum(v)/len(v):.0f}")
PY
