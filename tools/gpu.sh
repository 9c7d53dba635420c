#!/bin/bash
# Build everything, and only if that succeeds send the tree to a GPU box: tools/gpu.sh <timeout> <command...>
cd /root/repo
make -C visualslam_amd/csrc 2>&1 | grep -E "error|Error" -A4 | head -30
[ "${PIPESTATUS[0]}" = 0 ] || { echo "BUILD FAILED (csrc) - not sending"; exit 1; }
make -C visualslam_amd/cxx 2>&1 | grep -E "error|Error" -A4 | head -30
[ "${PIPESTATUS[0]}" = 0 ] || { echo "BUILD FAILED (cxx) - not sending"; exit 1; }
T=$1; shift
/usr/local/graft/bin/gpurun --timeout $T -- "$@"
