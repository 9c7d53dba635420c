#!/usr/bin/env python3
"""the reference's SECOND constructor, GaussPyramid(img, sigma) (GaussPyramid.cpp:150-152), derives
the octave count floor(log2(min(rows, cols))) - 4 from the 2x-upsampled image: 6 octaves at 1080p, 7 at 4K.  Per-image build
latency (PCIe included) and batched frames/s for 4 / 5 / 6 octaves at 1080p, and the per-octave kernel time of the
deep octaves (the library's HIP-event hook around every kernel family that serves them).  Needs a GPU.

    python tools/bench_auto_octaves.py [--frames 64] [--steps 5]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    args = ap.parse_args()
    import torch

    from visualslam_amd import capi, synth

    capi.build()
    rows, cols, n = args.rows, args.cols, args.frames
    auto = capi.auto_num_octaves(rows, cols)  # calculateNumOctaves(img) runs on the image as given (GaussPyramid.hpp:18-19), before the x2 upsample
    st = torch.cuda.Stream()
    torch.cuda.set_stream(st)
    ctx = capi.Context(0, st.cuda_stream)
    img = synth.frame_np(rows, cols)
    res = {"rows": rows, "cols": cols, "auto_octaves": auto, "frames_per_batch": n, "per_image_ms": {}, "batched": {}}
    for n_oct in sorted({4, 5, auto}):
        for _ in range(3):
            ctx.pyramid(img, n_oct, 1.6).close()
        t = time.perf_counter()
        for _ in range(10):
            ctx.pyramid(img, n_oct, 1.6).close()
        res["per_image_ms"][str(n_oct)] = (time.perf_counter() - t) / 10 * 1e3
    frames = synth.frames_torch(n, rows, cols, stream_id=0, device="cuda:0")
    for n_oct in sorted({4, 5, auto}):
        p = capi.default_params(rows, cols, n_octaves=n_oct)
        L = capi.batch_layout(p)
        dev = "cuda:0"
        o = dict(response=torch.empty((n, rows, cols), dtype=torch.float32, device=dev), nms_mask=torch.empty((n, rows, cols), dtype=torch.uint8, device=dev),
                 harris_kps=torch.empty((n, p.harris_cap, 3), dtype=torch.int32, device=dev), harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                 pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                 extrema_bits=torch.empty((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                 dog_points=torch.empty((n, p.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))
        for _ in range(3):
            ctx.detect_batch(p, frames, **o)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(args.steps):
            ctx.detect_batch(p, frames, **o)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / args.steps
        b = {"ms_per_step": dt * 1e3, "frames_per_sec": n / dt, "dog_points_per_step": int(o["dog_counts"].sum()),
             "octave_sizes": [[L.rows[k], L.cols[k]] for k in range(n_oct)], "per_octave_kernel_ms": {}}
        for k in range(2, n_oct):  # which kernel family serves the deep octaves, and what it costs in the step
            tot = {}
            for name in ("k_gauss_v_strip", "k_gauss_h_strip", "k_blur_h_generic", "k_blur_v_generic", "k_dog5", "k_extrema_w3"):
                ctx.kernel_timing_enable(f"{name}@{k}")
                ctx.detect_batch(p, frames, **o)
                torch.cuda.synchronize()
                nl, ms = ctx.kernel_timing_read()
                ctx.kernel_timing_enable(None)
                if nl:
                    tot[name] = {"launches": nl, "ms": ms}
            b["per_octave_kernel_ms"][str(k)] = tot
        res["batched"][str(n_oct)] = b
        del o
        torch.cuda.empty_cache()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
