"""Worst case for the batched filterKeypoints stage: uniform-noise frames, where every octave-0
keypoint survives the edge test (~21 k survivors, ~29 k oriented points per 1080p frame)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visualslam_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = "cuda:0"
ctx = capi.Context(0)
host = np.stack([synth.frame_np(1080, 1920, frame=f, kind="noise") for f in range(min(n, 4))])
frames = torch.from_numpy(np.concatenate([host] * ((n + len(host) - 1) // len(host)))[:n].copy()).to(dev)
for orient in (0, 1):
    p = capi.default_params(1080, 1920, localize=1, orient=orient, oriented_cap=1 << 16)
    L = capi.batch_layout(p)
    o = dict(response=torch.empty((n, 1080, 1920), dtype=torch.float32, device=dev), nms_mask=torch.empty((n, 1080, 1920), dtype=torch.uint8, device=dev),
             harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev), harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
             pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev), extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
             dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))
    if orient:
        o["oriented_points"] = torch.zeros((n, p.oriented_cap, 6), dtype=torch.int32, device=dev)
        o["oriented_counts"] = torch.zeros(n, dtype=torch.int32, device=dev)
    for _ in range(2):
        ctx.detect_batch(p, frames, **o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 5
    for _ in range(K):
        ctx.detect_batch(p, frames, **o)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    extra = " oriented/frame %.0f" % (o["oriented_counts"].float().mean().item()) if orient else ""
    print("noise frames, orient=%d: %.2f ms per %d-frame batch = %.0f frames/s, dog points/frame %.0f%s" %
          (orient, dt * 1e3, n, n / dt, o["dog_counts"].float().mean().item(), extra))
