#!/bin/bash
# Knock-out builds of the 16 x 16 x 64 octave-0 kernel of the diagnostics build (timing only; results wrong by construction;
# masks: 1 no HBM stores, 2 no tile staging, 8 no arithmetic - 10 = the stores alone):
#   tools/mx16_knockout.sh build        here (no GPU): visualslam_amd/lib/ab/ko16_<mask>.so for masks 0 1 2 3
#   tools/mx16_knockout.sh run          on the GPU box: tools/mx_alone.py --octaves 1 under each
cd "$(dirname "$0")/.."
ROOT=$(pwd); C=$ROOT/visualslam_amd/csrc; O=$ROOT/visualslam_amd/lib/obj; AB=$ROOT/visualslam_amd/lib/ab
mkdir -p $AB
if [ "$1" = build ]; then
  for k in 0 16 2 1; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form -DVSLAM_DIAGNOSTICS -DVSLAM_MX16_KO=$k -c -o $AB/vslam_mx_ko$k.o $C/vslam_mx.hip || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $AB/ko16_$k.so $O/vslam_hip_diag.o $O/vslam_sched_diag.o $O/vslam_params.o $O/vslam_mx0.o $AB/vslam_mx_ko$k.o || exit 1
  done
  rm -f $AB/vslam_mx_ko*.o
  ls -la $AB/ko16_*.so
else
  for k in 0 16 2 1; do
    echo -n "mask $k (1 = no HBM stores, 2 = no staging, 8 = no arithmetic, 16 = no LDS source rectangle): "
    VSLAM_MX=1 VSLAM_MX_FORM=16 VSLAM_LIBRARY=$AB/ko16_$k.so python3 tools/mx_alone.py --octaves 1 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('%.3f ms per step' % d['octave_kernel_ms_per_step'])"
  done
fi
