"""Host-side enqueue time of one vslam_detect_batch_dev call vs its GPU time (single frame)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualslam_amd import capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda:0"
ctx = capi.Context(0, torch.cuda.current_stream().cuda_stream)
p = capi.default_params(1080, 1920)
L = capi.batch_layout(p)
frames = synth.frames_torch(n, 1080, 1920, device=dev)
o = dict(response=torch.empty((n, 1080, 1920), dtype=torch.float32, device=dev), nms_mask=torch.empty((n, 1080, 1920), dtype=torch.uint8, device=dev),
         harris_kps=torch.zeros((n, p.harris_cap, 3), dtype=torch.int32, device=dev), harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
         pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev), extrema_bits=torch.zeros((n, L.bits_frame_words), dtype=torch.int64, device=dev),
         dog_points=torch.zeros((n, p.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))
for _ in range(10):
    ctx.detect_batch(p, frames, **o)
torch.cuda.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    ctx.detect_batch(p, frames, **o)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("frames %d: host enqueue %.1f us per call, total %.1f us per call" % (n, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
