#!/bin/bash
# Round 6's soak of the FINAL kernels: random sizes / batch sizes / octave counts / contents / list modes, every output against the
# oracle (tools/soak_batch.py), on the default path, with the f32 stages fused, on the matrix path in both MFMA shapes.
cd $GRAFT_REPO_ROOT
D=$GRAFT_REPO_ROOT/visualslam_amd/lib/libvslam_diag.so
python3 tools/soak_batch.py 601 150 && VSLAM_SOAK_F32_FUSED=1 python3 tools/soak_batch.py 602 100 && python3 tools/soak_batch.py 603 80 deep && \
VSLAM_MX=1 python3 tools/soak_batch.py 604 120 && VSLAM_MX=1 VSLAM_SOAK_F32_FUSED=1 python3 tools/soak_batch.py 605 80 big && \
VSLAM_LIBRARY=$D VSLAM_MX=1 VSLAM_MX_FORM=16 python3 tools/soak_batch.py 606 150 && VSLAM_LIBRARY=$D VSLAM_MX=1 VSLAM_MX_FORM=16 python3 tools/soak_batch.py 607 90 big && \
VSLAM_LIBRARY=$D VSLAM_MX=1 VSLAM_MX_FORM=16 python3 tools/soak_batch.py 608 60 deep
