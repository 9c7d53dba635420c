"""One-off randomized parity soak of the batched path (not part of the test suite)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_batch import run_batch, check_frame
from visualslam_amd import capi, synth
capi.build()
ctx = capi.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); it = 0
while time.time() - t0 < (float(sys.argv[2]) if len(sys.argv) > 2 else 120):
    rows, cols = int(rng.integers(17, 140)), int(rng.integers(17, 200))
    n = int(rng.choice([1, 2, 5, 31, 32, 33, 63, 64, 65, 90]))
    n_oct = int(rng.integers(1, max(1, min(4, capi.auto_num_octaves(rows, cols))) + 1))
    mode = int(rng.integers(0, 3))
    kinds = ["checker", "noise"]
    frames = np.stack([synth.frame_np(rows, cols, f, int(rng.integers(0, 50)), kinds[int(rng.integers(0, 2))]) for f in range(n)])
    p, L, out = run_batch(ctx, torch, frames, n_octaves=n_oct, harris_cap=4096, dog_cap=16384, localize=int(mode >= 1), orient=int(mode == 2), with_nms2=bool(rng.integers(0, 2)))
    for f in sorted(set([0, n // 2 - 1 if n > 1 else 0, n // 2, n - 1])):
        check_frame(p, L, out, f, frames[f], n_oct)
    it += 1
print("soak ok", it, "cases")
