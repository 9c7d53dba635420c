#!/bin/bash
# Regenerates the evidence behind bench.py's headline line on a GPU box (run via gpurun):
#   default bench line, rocprofv3 kernel stats of the same command, three separate PMC passes.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-refresh}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o runc -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 > $OUT/prof.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmcA -o r -- python3 $GRAFT_REPO_ROOT/bench.py --frames 64 --steps 1 --warmup 1 --cpu-sample 0 > $OUT/pmcA.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmcB -o r -- python3 $GRAFT_REPO_ROOT/bench.py --frames 64 --steps 1 --warmup 1 --cpu-sample 0 > $OUT/pmcB.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmcC -o r -- python3 $GRAFT_REPO_ROOT/bench.py --frames 64 --steps 1 --warmup 1 --cpu-sample 0 > $OUT/pmcC.log 2>&1
cut -c1-400 $OUT/bench.json
