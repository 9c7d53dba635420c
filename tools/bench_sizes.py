"""Per-image pyramid-build + Harris latency across frame sizes (fast paths need cols % 8 == 0 at
every octave; other sizes take the generic kernels).  Needs a GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualslam_amd import capi, synth

ctx = capi.Context(0)
for shape in ((1240, 1754), (1240, 1752), (600, 868), (600, 864), (384, 512), (1080, 1920), (1080, 1924)):
    img = synth.frame_np(*shape)
    for _ in range(2):
        p = ctx.pyramid(img, 4, 1.6); p.close()
    t = time.perf_counter()
    for _ in range(10):
        p = ctx.pyramid(img, 4, 1.6); p.close()
    tb = (time.perf_counter() - t) / 10 * 1e3
    ctx.harris_keypoints(img)
    t = time.perf_counter()
    for _ in range(10):
        ctx.harris_keypoints(img)
    th = (time.perf_counter() - t) / 10 * 1e3
    print(shape, "pyramid %.2f ms  harris_keypoints %.2f ms" % (tb, th), flush=True)
