#!/usr/bin/env bash
# round-5 GPU session 8: the capture crash with both ends of the stack; the stand-alone reproducer's modes
set -o pipefail
mkdir -p gpurun_out
{
for m in 2 3 4 5; do
  echo "== stand-alone reproducer, /opt/rocm runtime, mode $m"
  LD_PRELOAD=$PWD/tools/segv_bt.so timeout -k 5 60 tools/capture_cycle_repro $m 2>&1 | tail -12
  echo "exit code ${PIPESTATUS[0]}"
done
} > gpurun_out/r05_capture_cycle_repro.txt 2>&1
grep -n "exit code\|== stand\|kernels run\|SEGV" gpurun_out/r05_capture_cycle_repro.txt
timeout -k 10 300 python tools/graph_try.py --case raw 3 > gpurun_out/r05_graph_try_bt.out 2> gpurun_out/r05_graph_try_bt.err
echo "graph_try single case rc=$?"
grep -n "SEGV_BT" -A48 gpurun_out/r05_graph_try_bt.err | cut -c1-170 | head -80
