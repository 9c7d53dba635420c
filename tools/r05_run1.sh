#!/usr/bin/env bash
# round-5 GPU session 1: new tests, the capture reproducer, host-fed compact records A/B
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_batch.py -m gpu -q -x -k "pack_points16 or pack_lists or captured" > gpurun_out/r05_t1.log 2>&1; echo "pytest batch rc=$?"; tail -3 gpurun_out/r05_t1.log
python -m pytest tests/test_cxx_mirror.py -m gpu -q -x > gpurun_out/r05_t2.log 2>&1; echo "pytest cxx rc=$?"; tail -3 gpurun_out/r05_t2.log
timeout -k 10 500 python tools/graph_try.py > gpurun_out/r05_graph_try.jsonl 2> gpurun_out/r05_graph_try.err; echo "graph_try rc=$?"; cat gpurun_out/r05_graph_try.jsonl
S=visualslam_amd/bin/Stream
for mx in 0 1; do for c in 0 1 0 1; do
  VSLAM_MX=$mx timeout -k 10 120 $S --mode hostfed --frames 256 --batches 40 --warmup 6 --compact $c 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('mx=$mx compact=$c', d['frames_per_sec'], d['steady_frames_per_sec'])"
done; done | tee gpurun_out/r05_hostfed_compact.txt
