#!/bin/bash
# LDS-pipe occupancy per kernel for one bench configuration (one rocprofv3 --pmc pass, no trace):
#   bash tools/pmc_lds.sh <outdir under gpurun_out> [bench args...]
# SQ_ACTIVE_INST_LDS / SQ_BUSY_CU_CYCLES-style ratios are what says whether the LDS unit has room.
D=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $D -o r -- python3 $GRAFT_REPO_ROOT/bench.py --frames 64 --steps 1 --warmup 1 --cpu-sample 0 --modes 0 --cxx-host 0 --live-traffic 0 --mx 0 "$@" > $D.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - $D <<'P'
import csv, glob, collections, sys, json
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vslam" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vslam::", "")[:70]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {}
for k, v in sorted(acc.items(), key=lambda x: -x[1].get("SQ_WAVE_CYCLES", 0)):
    out[k] = dict(v)
    b = v.get("SQ_BUSY_CYCLES", 0) or 1
    print(f"{k:72s} lds_inst {v.get('SQ_INSTS_LDS',0):.3g} active_lds/busy {v.get('SQ_ACTIVE_INST_LDS',0)/b:.3f} idx_active/busy {v.get('SQ_LDS_IDX_ACTIVE',0)/b:.3f} conflict/idx_active {v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',0),1):.3f} active_valu/busy {v.get('SQ_ACTIVE_INST_VALU',0)/b:.3f}")
json.dump(out, open(sys.argv[1] + ".json", "w"), indent=1)
P
