#!/usr/bin/env bash
# round-5 GPU session 4: side streams at the main stream's priority from the start vs the lowest priority, per queue count
set -o pipefail
mkdir -p gpurun_out
S=visualslam_amd/bin/Stream
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['frames_per_sec'], d['steady_frames_per_sec'], d['join_watch'], d['gpu_max_hw_queues'])"; }
{
for rep in 1 2; do for q in 2 3 4 6 12; do for lv in 0 1; do
  VSLAM_JOIN_WATCH_LEVEL=$lv timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 8 --hw-queues $q 2>/dev/null | tail -1 | line "device q=$q level=$lv"
done; done; done
for lv in 0 1; do for mx in 0 1; do
  VSLAM_MX=$mx VSLAM_JOIN_WATCH_LEVEL=$lv timeout -k 10 120 $S --mode hostfed --frames 256 --batches 40 --warmup 6 2>/dev/null | tail -1 | line "hostfed mx=$mx level=$lv"
  VSLAM_MX=$mx VSLAM_JOIN_WATCH_LEVEL=$lv timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 8 2>/dev/null | tail -1 | line "device default-env mx=$mx level=$lv"
done; done
} | tee gpurun_out/r05_priority_sweep.txt
for lv in 0 1 0 1; do
  VSLAM_JOIN_WATCH_LEVEL=$lv python bench.py --steps 20 --warmup 5 --modes 0 --cxx-host 0 --cpu-sample 0 --live-traffic 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('python bench level=$lv value', d['value'], 'mx', (d.get('mx_path') or {}).get('frames_per_sec'))"
done | tee gpurun_out/r05_priority_bench.txt
timeout -k 10 400 python tools/graph_try.py > gpurun_out/r05_graph_try.jsonl 2> gpurun_out/r05_graph_try.err; echo "graph_try rc=$?"; python -c "
import json
for l in open('gpurun_out/r05_graph_try.jsonl'):
    d=json.loads(l); print(d['case'], d['exit_code'], d.get('steps_seen'))"
