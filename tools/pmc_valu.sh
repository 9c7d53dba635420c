#!/bin/bash
# VALU wave-instructions per frame and kernel for one bench configuration (rocprofv3 --pmc pass, no trace):
#   bash tools/pmc_valu.sh <outdir under gpurun_out> [bench args...]
D=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $D -o r -- python3 $GRAFT_REPO_ROOT/bench.py --frames 64 --steps 1 --warmup 1 --cpu-sample 0 --modes 0 "$@" > $D.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $D <<'P'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "SQ_INSTS_VALU" and "vslam" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vslam::", "")[:60]
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
tot = 0
for k, v in sorted(acc.items(), key=lambda x: -x[1][0]):
    print(f"{k:62s} {v[0]/2/64/1e6:8.3f} M wave-instr/frame  launches {v[1]}")   # warm-up + timed step, 64 frames each
    tot += v[0] / 2 / 64 / 1e6
print("total", round(tot, 3))
P
