#!/usr/bin/env bash
# round-5 GPU session 6: fused upsample in the default octave-0 kernel: parity, A/B, mx pack-on-main experiment
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_batch.py tests/test_gpu_ref_images.py tests/test_gpu_large.py -m gpu -q -x > gpurun_out/r05_t7.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r05_t7.log
for rep in 1 2; do for f in 1 0; do
  VSLAM_FUSE_UP2=$f python bench.py --steps 20 --warmup 5 --modes 0 --cxx-host 0 --cpu-sample 0 --live-traffic 0 --mx 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('fuse_up2=$f value', round(d['value'],1), 'ms', round(d['ms_per_step'],3), 'k_pyr launches', r['launches'], 'avg ms', round(r['avg_launch_ms'],3), 'frac', round(r['frac'],4))"
done; done | tee gpurun_out/r05_fuse_up2_ab.txt
for rep in 1 2; do for pm in 0 1; do
  VSLAM_MX_PACK_MAIN=$pm python bench.py --steps 20 --warmup 5 --modes 0 --cxx-host 0 --cpu-sample 0 --live-traffic 0 --matrix-path 1 --mx 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('mx pack_main=$pm value', round(d['value'],1), 'ms', round(d['ms_per_step'],3))"
done; done | tee gpurun_out/r05_mx_pack_main_ab.txt
