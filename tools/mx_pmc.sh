#!/bin/bash
# PMC passes (separate, no tracing) of the matrix path's kernels on a 64-frame batch: tools/mx_pmc.sh <tag> [extra bench args]
R=${1:-mxpmc}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
ARGS="--frames 64 --steps 1 --warmup 1 --cpu-sample 0 --modes 0 --live-traffic 0 --cxx-host 0 --mx 0 --matrix-path 1 $@"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmcA -o r -- python3 $B $ARGS > $OUT/pmcA.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcB -o r -- python3 $B $ARGS > $OUT/pmcB.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcC -o r -- python3 $B $ARGS > $OUT/pmcC.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py $OUT/pmcA $OUT/pmcB $OUT/pmcC > $OUT/pmc.json
python3 - <<PY
import json
d=json.load(open("$OUT/pmc.json"))
tot=0
for k,v in d.items():
    hb=v.get("hbm_bytes_total",0)/2/64/1e6   # two batches (warm-up + step) of 64 frames
    tot+=hb
    print(f"{k:28s} launches {v['launches']:3d} waves {v.get('SQ_WAVES',0):10.0f} valu/launch {v.get('SQ_INSTS_VALU',0)/1e6:8.2f}M  vmem_wr {v.get('SQ_INSTS_VMEM_WR',0)/1e6:6.2f}M rd {v.get('SQ_INSTS_VMEM_RD',0)/1e6:6.2f}M wait_any {v.get('SQ_WAIT_ANY',0)/max(1,v.get('SQ_WAVE_CYCLES',1)):.2f}  HBM MB/frame {hb:7.2f}")
print("total MB/frame", tot)
PY
