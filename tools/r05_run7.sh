#!/usr/bin/env bash
# round-5 GPU session 7: stand-alone capture reproducer on both runtimes, parity soaks (standard, deep, matrix path)
set -o pipefail
mkdir -p gpurun_out
{
for rt in rocm72 torch70; do
  for m in 0 1 2; do
    echo "== runtime $rt mode $m"
    if [ $rt = torch70 ]; then export LD_LIBRARY_PATH=/usr/local/lib/python3.10/dist-packages/torch/lib:$LD_LIBRARY_PATH; fi
    LD_PRELOAD=$PWD/tools/segv_bt.so timeout -k 5 60 tools/capture_cycle_repro $m 2>&1 | awk '{ if ($0 == last) { n++ } else { if (n > 0) print "    ... the same line " n " more times"; print; n = 0 } last = $0 } END { if (n > 0) print "    ... the same line " n " more times" }' | tail -25
    echo "exit code ${PIPESTATUS[0]}"
  done
done
} > gpurun_out/r05_capture_cycle_repro.txt 2>&1
tail -60 gpurun_out/r05_capture_cycle_repro.txt
unset LD_LIBRARY_PATH
timeout -k 10 260 python tools/soak_batch.py 501 200 deep > gpurun_out/r05_soak_deep.txt 2>&1; echo "soak deep rc=$?"; tail -2 gpurun_out/r05_soak_deep.txt
timeout -k 10 200 python tools/soak_batch.py 502 150 > gpurun_out/r05_soak_std.txt 2>&1; echo "soak std rc=$?"; tail -2 gpurun_out/r05_soak_std.txt
VSLAM_MX=1 timeout -k 10 200 python tools/soak_batch.py 503 150 > gpurun_out/r05_soak_mx.txt 2>&1; echo "soak mx rc=$?"; tail -2 gpurun_out/r05_soak_mx.txt
VSLAM_MX=1 timeout -k 10 200 python tools/soak_batch.py 504 120 deep > gpurun_out/r05_soak_mx_deep.txt 2>&1; echo "soak mx deep rc=$?"; tail -2 gpurun_out/r05_soak_mx_deep.txt
