#!/bin/bash
# One PMC pass of the descriptor stage (params.describe) on a 64-frame batch of the bench's `modes` content: tools/sift_pmc.sh <tag>
R=${1:-sift}; OUT=$GRAFT_REPO_ROOT/gpurun_out/$R; mkdir -p $OUT
T=$GRAFT_REPO_ROOT/tools/orient_alone.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmcA -o r -- python3 $T --frames 64 --steps 1 --describe 1 > $OUT/pmcA.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmcB -o r -- python3 $T --frames 64 --steps 1 --describe 1 > $OUT/pmcB.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, json, collections
small = json.loads([l for l in open("$OUT/pmcA.log") if l.startswith("{")][-1])
for d in ("pmcA", "pmcB"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("vslam::", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    per = 2 * small["oriented"]
    for k in ("k_sift_descriptors_batch", "k_orient_survivors_pk"):
        print(d, k, "per oriented point" if "sift" in k else "per oriented point (x 2.43 per survivor)", {c: round(x / per, 1) for c, x in agg[k].items()})
print(small)
PY
