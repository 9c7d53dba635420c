#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/mxdbg
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_large.py -m gpu -x -q -k "matrix_path_switch or matrix_core_octave or batched_4k" > $OUT/test.log 2>&1; rc=$?
tail -15 $OUT/test.log
[ $rc = 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp
VSLAM_AUX_STREAMS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/serial -o run -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-sample 0 --modes 0 --live-traffic 0 --cxx-host 0 --mx 0 --steps 6 --matrix-path 1 > $OUT/serial.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/mxdbg/serial/run_kernel_stats.csv")))
for r in rows[:12]:
    if 'at::native' in r["Name"]: continue
    print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e6:8.3f} ms')
PY
python3 bench.py --cpu-sample 0 --modes 0 --live-traffic 0 --cxx-host 0 --steps 8 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',d['value'],d['ms_per_step']); m=d['mx_path']; print('mx', m['frames_per_sec'], m['ms_per_step'], m['speedup_vs_value'], 'alone', m['k_pyr_octave_mx']['alone'])"
