"""One-off randomized parity soak of the batched path (not part of the test suite).

    [VSLAM_MX=1] [VSLAM_SOAK_F32_FUSED=1] python tools/soak_batch.py <seed> <seconds> [big|deep]
VSLAM_SOAK_F32_FUSED=1 (round 6): the f32 stages with fused multiply-adds on both sides (vslam_ctx_set_f32_fused against
oracle.fma_variant); with VSLAM_LIBRARY=lib/libvslam_diag.so VSLAM_MX=1 VSLAM_MX_FORM=16 the matrix path runs its 16 x 16 x 64 kernels.
`deep` (round 5): 1..6 octaves whatever the frame size (the reference's constructor takes any count) - octaves 4 and 5 run
kernels of hundreds of taps on images of a few pixels (strip kernels up to 2047 taps, repeated BORDER_REFLECT_101).
`big`: frames up to 400 x 700 in batches of 1-9 (octave 0 up to 800 x 1400: several seams and straddling lattice rows of the
matrix path's fused scan per frame); with VSLAM_MX=1 the whole sweep runs on the opt-in matrix path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_batch import run_batch, check_frame
from visualslam_amd import capi, synth
capi.build()
ctx = capi.Context(0)
fused = os.environ.get("VSLAM_SOAK_F32_FUSED") == "1"
if fused:
    import oracle as _o
    ctx.set_f32_fused(True)
    _o.lib().vo_set_fma_variant(3)
last_print = time.time()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); it = 0
while time.time() - t0 < (float(sys.argv[2]) if len(sys.argv) > 2 else 120):
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    rows, cols = (int(rng.integers(17, 400)), int(rng.integers(17, 700))) if big else (int(rng.integers(17, 140)), int(rng.integers(17, 200)))
    n = int(rng.choice([1, 2, 3, 9])) if big else int(rng.choice([1, 2, 5, 31, 32, 33, 63, 64, 65, 90]))
    deep = len(sys.argv) > 3 and sys.argv[3] == "deep"
    n_oct = int(rng.integers(1, 7)) if deep else int(rng.integers(1, max(1, min(4, capi.auto_num_octaves(rows, cols))) + 1))
    if deep:  # (the batches stay small: the oracle's 977-tap blurs are what takes the time)
        n = int(rng.choice([1, 2, 5, 33]))
    mode = int(rng.integers(0, 3))
    kinds = ["checker", "noise"]
    frames = np.stack([synth.frame_np(rows, cols, f, int(rng.integers(0, 50)), kinds[int(rng.integers(0, 2))]) for f in range(n)])
    p, L, out = run_batch(ctx, torch, frames, n_octaves=n_oct, harris_cap=4096, dog_cap=16384, localize=int(mode >= 1), orient=int(mode == 2), with_nms2=bool(rng.integers(0, 2)))
    for f in sorted(set([0, n // 2 - 1 if n > 1 else 0, n // 2, n - 1])):
        check_frame(p, L, out, f, frames[f], n_oct)
    if it % 4 == 0:  # the packed form of the DoG list (vslam_pack_lists_dev) = the per-frame lists back to back
        dev = "cuda:0"
        lists, counts = torch.from_numpy(out["dog_points"]).to(dev), torch.from_numpy(out["dog_counts"]).to(dev)
        m = np.minimum(out["dog_counts"], p.dog_cap).astype(np.int64)
        packed = torch.zeros((int(m.sum()) + 1, 6), dtype=torch.int32, device=dev)
        off = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        ctx.pack_lists(lists, counts, packed, off)
        torch.cuda.synchronize()
        assert (off.cpu().numpy() == np.concatenate([[0], np.cumsum(m)])).all()
        want = np.concatenate([out["dog_points"][f][: m[f]] for f in range(n)])
        assert (packed.cpu().numpy()[: len(want)] == want).all()
        p16 = torch.zeros((int(m.sum()) + 1, 4), dtype=torch.int32, device=dev)  # the 16-byte form and its host-side inverse (round 5)
        ctx.pack_points16(lists, counts, p16, off)
        torch.cuda.synchronize()
        assert capi.points16_expand(p16.cpu().numpy()[: len(want)]).tobytes() == np.ascontiguousarray(want).astype(np.int32).tobytes()
    if it % 5 == 0 and mode == 0:  # the dense 3x3x3 extension in the batch, two frames against the oracle
        import oracle
        pd = capi.default_params(rows, cols, n_octaves=n_oct, extrema_dense=1, dog_cap=1 << 17)
        Ld = capi.batch_layout(pd)
        dev = "cuda:0"
        o = dict(pyramid=torch.empty((n, Ld.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                 extrema_bits=torch.zeros((n, Ld.bits_frame_words), dtype=torch.int64, device=dev),
                 dog_points=torch.zeros((n, pd.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))
        ctx.detect_batch(pd, torch.from_numpy(frames).to(dev), **o)
        torch.cuda.synchronize()
        for f in sorted(set([0, n - 1])):
            w = oracle.Pyramid(frames[f], n_oct)
            allp = np.concatenate([w.extrema_dense(oc, 8)[1] for oc in range(n_oct)])
            w.close()
            assert int(o["dog_counts"][f]) == len(allp)
            k = min(len(allp), pd.dog_cap)
            assert o["dog_points"][f][:k].cpu().numpy().view(capi.POINT_DTYPE).reshape(-1).tobytes() == allp[:k].tobytes()
    if it % 7 == 0 and mode <= 1:  # the host-memory entry point (vslam_detect_batch_host): the same lists, packed
        res = ctx.detect_batch_host(capi.default_params(rows, cols, n_octaves=n_oct, harris_cap=4096, dog_cap=16384, localize=int(mode >= 1)), frames)
        for name, key, cnt, cap in (("harris", "harris_kps", "harris_counts", p.harris_cap), ("dog", "dog_points", "dog_counts", p.dog_cap)):
            rec, off, c = res[name]
            m = np.minimum(out[cnt], cap).astype(np.int64)
            assert (c == out[cnt].astype(np.uint32)).all() and (off == np.concatenate([[0], np.cumsum(m)]).astype(np.uint64)).all()
            want = np.concatenate([out[key][f][: m[f]] for f in range(n)]).reshape(-1)
            assert rec.view(np.int32).reshape(-1).tobytes() == want.astype(np.int32).tobytes()
    it += 1
    if time.time() - last_print > 45:
        print(f"... {it} cases, {time.time() - t0:.0f} s", flush=True)  # gpurun takes a silent run for a hung one
        last_print = time.time()
print("soak ok", it, "cases", "(matrix path" + (", form " + os.environ.get("VSLAM_MX_FORM", "32") if ctx.matrix_path() else "") + ")" if ctx.matrix_path() else "(default path)",
      "f32 fused" if fused else "f32 rounded", sys.argv[3] if len(sys.argv) > 3 else "")
