#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/mxdbg
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
ARGS="--frames 64 --steps 1 --warmup 1 --cpu-sample 0 --modes 0 --live-traffic 0 --cxx-host 0 --mx 0 --matrix-path 1"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pA -o r -- python3 $B $ARGS > $OUT/pA.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/pB -o r -- python3 $B $ARGS > $OUT/pB.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("gpurun_out/mxdbg/pA","gpurun_out/mxdbg/pB"):
    for f in glob.glob(d+"/**/*_counter_collection.csv",recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "vslam" not in k: continue
            key=(k[k.index("MxCfg<"):k.index(">(")+1][:60] if "MxCfg<" in k else k.split("(")[0][-40:])
            acc[key][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    wc=max(1,v["SQ_WAVE_CYCLES"]); w=max(1,v["SQ_WAVES"])
    print(f'{k:62s} waves {w:8.0f} valu/w {v["SQ_INSTS_VALU"]/w:6.0f} lds/w {v["SQ_INSTS_LDS"]/w:5.0f} | wait_any {v["SQ_WAIT_ANY"]/wc:.2f} wait_inst {v["SQ_WAIT_INST_ANY"]/wc:.2f} act_valu {v["SQ_ACTIVE_INST_VALU"]/wc:.2f} act_any {v["SQ_ACTIVE_INST_ANY"]/wc:.2f} cyc/w {wc/w*4:8.0f}')
PY
