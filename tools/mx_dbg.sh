#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/mxdbg
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -x -q -k "harris or batch_random_shapes or config2_and_3 or ragged or tiny or small_frames or config4" > $OUT/test.log 2>&1; rc=$?
tail -3 $OUT/test.log
[ $rc = 0 ] || exit $rc
python3 bench.py --cpu-sample 0 --modes 0 --octaves 0 --kernel k_harris_strip --live-traffic 0 --cxx-host 0 --mx 0 --steps 10 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('harris only', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
python3 bench.py --cpu-sample 0 --modes 0 --live-traffic 0 --cxx-host 0 --steps 8 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',d['value'],d['ms_per_step']); m=d['mx_path']; print('mx', m['frames_per_sec'], m['ms_per_step'], m['speedup_vs_value'])"
