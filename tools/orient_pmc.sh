#!/bin/bash
# The orientation stage on the bench's mixed content: timing, a kernel trace and one PMC pass of a 64-frame batch.
#   tools/orient_pmc.sh <tag>
R=${1:-orient}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
T=$GRAFT_REPO_ROOT/tools/orient_alone.py
python3 $T > $OUT/alone.json 2> $OUT/alone.err && cat $OUT/alone.json &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o r -- python3 $T --frames 64 --steps 2 > $OUT/kt.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmcA -o r -- python3 $T --frames 64 --steps 1 > $OUT/pmcA.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, json, collections
st = glob.glob("$OUT/kt/**/*kernel_stats.csv", recursive=True)
import re
tr = glob.glob("$OUT/kt/**/*kernel_trace.csv", recursive=True)
for r in [r for r in csv.DictReader(open(tr[0])) if "k_orient_survivors" in r["Kernel_Name"]][:4]:
    print(r["Kernel_Name"].split("(")[0], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us; vgpr", r.get("VGPR_Count", ""), "lds", r.get("LDS_Block_Size", ""))
for row in list(csv.DictReader(open(st[0])))[:6]:
    print(f"{row['Name'][:60]:60s} calls {row['Calls']:>5s} total_ms {float(row['TotalDurationNs'])/1e6:9.3f} avg_us {float(row['AverageNs'])/1e3:9.1f} {row['Percentage']}%")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("$OUT/pmcA/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("vslam::", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "SQ_WAVES": n[k] += 1
alone = json.load(open("$OUT/alone.json"))
for k in ("k_orient_survivors_pk", "k_orient_survivors", "k_edge_flags"):
    v = agg[k]
    if not v: continue
    print(k, "launches", n[k], {c: round(x / 1e6, 2) for c, x in v.items()}, "M; wait_any", round(v["SQ_WAIT_ANY"] / max(1, v["SQ_WAVE_CYCLES"]), 3))
small = json.loads([l for l in open("$OUT/pmcA.log") if l.startswith("{")][-1])
v = collections.defaultdict(float)
for k in ("k_orient_survivors_pk", "k_orient_survivors"):
    for c, x in agg[k].items(): v[c] += x
per = 2 * small["survivors"]  # the warm-up batch and the one step
print("per survivor: VALU", round(v["SQ_INSTS_VALU"] / per), "LDS", round(v["SQ_INSTS_LDS"] / per), "SALU", round(v["SQ_INSTS_SALU"] / per),
      "wave-cycles", round(v["SQ_WAVE_CYCLES"] / per), "busy-cycles(sum over SEs)", round(v["SQ_BUSY_CYCLES"] / per), "survivors / 64 frames", small["survivors"])
print(json.dumps(alone))
PY
