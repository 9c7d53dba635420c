"""One frame of a given size through the per-image API (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import capi, synth
rows, cols = int(sys.argv[1]), int(sys.argv[2])
ctx = capi.Context(0)
img = synth.frame_np(rows, cols)
for _ in range(3):
    p = ctx.pyramid(img, 4, 1.6)
    for o in range(4):
        kp, n = p.keypoints(o, 3)
        p.filter_keypoints(o, kp)
    p.close()
    ctx.harris_keypoints(img)
