#!/usr/bin/env bash
# round-5 GPU session 3: wide-kernel strips, watchdog v2, capture reproducer with backtraces, queue sweep, auto octaves
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_large.py tests/test_gpu_batch.py -m gpu -q -x -k "large or auto or octave or watchdog or tuner or captured or 4k or portrait or 2050" > gpurun_out/r05_t5.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_t5.log
python -m pytest tests/test_cxx_mirror.py -m gpu -q -x > gpurun_out/r05_t6.log 2>&1; echo "pytest cxx rc=$?"; tail -3 gpurun_out/r05_t6.log
timeout -k 10 600 python tools/graph_try.py > gpurun_out/r05_graph_try.jsonl 2> gpurun_out/r05_graph_try.err; echo "graph_try rc=$?"
S=visualslam_amd/bin/Stream
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['frames_per_sec'], d['steady_frames_per_sec'], d['join_watch'], d['gpu_max_hw_queues'])"; }
{
for rep in 1 2; do for q in 2 3 4 6 12; do
  timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 12 --hw-queues $q 2>/dev/null | tail -1 | line "device q=$q"
done; done
timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 12 2>/dev/null | tail -1 | line "device default-env"
VSLAM_JOIN_WATCH=0 timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 12 2>/dev/null | tail -1 | line "device default-env watch=0"
VSLAM_JOIN_WATCH_LEVEL=1 timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 12 --hw-queues 4 2>/dev/null | tail -1 | line "device q=4 forced-level=1"
VSLAM_MX=1 timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 12 2>/dev/null | tail -1 | line "device mx default-env"
VSLAM_MX=1 timeout -k 10 120 $S --mode device --frames 256 --batches 30 --warmup 12 --hw-queues 12 2>/dev/null | tail -1 | line "device mx q=12"
} | tee gpurun_out/r05_queue_sweep2.txt
timeout -k 10 300 python tools/bench_auto_octaves.py > gpurun_out/r05_auto_octaves.json 2> gpurun_out/r05_auto_octaves.err; echo "auto octaves rc=$?"; cat gpurun_out/r05_auto_octaves.json
