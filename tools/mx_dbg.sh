#!/bin/bash
# EXPERIMENT: the matrix-core octave kernels alone (pyramid-only batch) + parity of the switch
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/mxdbg
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_batch.py -m gpu -x -q -k "matrix_path_switch" > $OUT/test.log 2>&1 || { tail -30 $OUT/test.log; exit 1; }
tail -2 $OUT/test.log
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  VSLAM_MX=$1 VSLAM_MX_DBG=$2 python3 tools/mx_alone.py > $OUT/alone_mx$1_dbg$2.txt 2>&1 || true
  echo "mx=$1 dbg=$2: $(tail -1 $OUT/alone_mx$1_dbg$2.txt)"
done
