"""Experiment: two batches in flight (two contexts, two streams, two output sets) against one.
Does the tail of batch k (small octaves, lists) hide under the head of batch k + 1?  FOLLOW=1 orders the two with
vslam_ctx_follow.  NOT part of the product or of bench.py's `value`."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualslam_amd import capi, synth

capi.build()
n, rows, cols = int(os.environ.get("N", 256)), 1080, 1920
dev = "cuda:0"
P = 2
frames = [synth.frames_torch(n, rows, cols, stream_id=i, device=dev) for i in range(P)]
p = capi.default_params(rows, cols)
L = capi.batch_layout(p)

def outs():
    return dict(response=torch.empty((n, rows, cols), dtype=torch.float32, device=dev),
                nms_mask=torch.empty((n, rows, cols), dtype=torch.uint8, device=dev),
                harris_kps=torch.empty((n, p.harris_cap, 3), dtype=torch.int32, device=dev),
                harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                extrema_bits=torch.empty((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                dog_points=torch.empty((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
                dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))

streams = [torch.cuda.Stream() for _ in range(P)]
ctxs = [capi.Context(0, s.cuda_stream) for s in streams]
o = [outs() for _ in range(P)]
for pipes in (1, 2):
    for k in range(12):  # six calls per context: the library's stream tuner has decided before the timing
        ctxs[k % pipes].detect_batch(p, frames[k % pipes], **o[k % pipes])
    torch.cuda.synchronize()
    K = 12
    t0 = time.perf_counter()
    for k in range(K):
        if pipes > 1 and os.environ.get("FOLLOW"):
            ctxs[k % pipes].follow(ctxs[(k - 1) % pipes])
        ctxs[k % pipes].detect_batch(p, frames[k % pipes], **o[k % pipes])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"pipes={pipes}: {n * K / dt:.0f} frames/s, {dt / K * 1e3:.2f} ms per batch", flush=True)
print([int(x["dog_counts"].sum()) for x in o])
