#!/usr/bin/env bash
# round-5 GPU session 2: watchdog test, capture reproducer with backtrace, queue sweep, host-fed compact A/B, auto octaves
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_batch.py -m gpu -q -x -k "watchdog or tuner or captured" > gpurun_out/r05_t3.log 2>&1; echo "pytest batch rc=$?"; tail -3 gpurun_out/r05_t3.log
python -m pytest tests/test_cxx_mirror.py -m gpu -q -x > gpurun_out/r05_t4.log 2>&1; echo "pytest cxx rc=$?"; tail -3 gpurun_out/r05_t4.log
timeout -k 10 600 python tools/graph_try.py > gpurun_out/r05_graph_try.jsonl 2> gpurun_out/r05_graph_try.err; echo "graph_try rc=$?"
S=visualslam_amd/bin/Stream
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['frames_per_sec'], d['steady_frames_per_sec'], d['join_watch'], d['gpu_max_hw_queues'])"; }
{
for rep in 1 2; do for q in 2 3 4 6 12; do for w in 1 0; do
  VSLAM_JOIN_WATCH=$w timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 8 --hw-queues $q 2>/dev/null | tail -1 | line "device q=$q watch=$w"
done; done; done
for q in 3 4; do timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 8 --hw-queues $q --tuner 2>/dev/null | tail -1 | line "device q=$q tuner"; done
timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 8 2>/dev/null | tail -1 | line "device default-env"
for lv in 1 2; do VSLAM_JOIN_WATCH_LEVEL=$lv timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 8 --hw-queues 12 2>/dev/null | tail -1 | line "device q=12 forced-level=$lv"; done
} | tee gpurun_out/r05_queue_sweep.txt
{
for mx in 0 1; do for c in 0 1 0 1; do
  VSLAM_MX=$mx timeout -k 10 120 $S --mode hostfed --frames 256 --batches 40 --warmup 6 --compact $c 2>/dev/null | tail -1 | line "hostfed mx=$mx compact=$c"
done; done
} | tee gpurun_out/r05_hostfed_compact.txt
timeout -k 10 300 python tools/bench_auto_octaves.py > gpurun_out/r05_auto_octaves.json 2> gpurun_out/r05_auto_octaves.err; echo "auto octaves rc=$?"; cat gpurun_out/r05_auto_octaves.json
