#!/bin/bash
# rocprofv3 kernel stats of the bench's main step in two settings, side by side: tools/kstats_two.sh "<args A>" "<args B>"
OUT=$GRAFT_REPO_ROOT/gpurun_out/kst; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--cpu-sample 0 --modes 0 --mx 0 --cxx-host 0 --live-traffic 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/a -o r -- python3 $GRAFT_REPO_ROOT/bench.py $Q $1 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b -o r -- python3 $GRAFT_REPO_ROOT/bench.py $Q $2 > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, json
def load(d):
    f = glob.glob("$OUT/" + d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"].replace("void ", "").replace("vslam::", "")[:60]: float(r["TotalDurationNs"]) / 1e6 / 11 for r in csv.DictReader(open(f)) if "at::native" not in r["Name"]}
a, b = load("a"), load("b")
line = lambda l: json.loads([x for x in open("$OUT/" + l + ".log") if x.startswith("{")][-1])
print("A ms_per_step", round(line("a")["ms_per_step"], 2), " B ms_per_step", round(line("b")["ms_per_step"], 2))
for k in sorted(set(a) | set(b), key=lambda k: -(a.get(k, 0) + b.get(k, 0)))[:16]:
    print(f"{k:60s} A {a.get(k, 0):7.3f}  B {b.get(k, 0):7.3f} ms per batch")
PY
