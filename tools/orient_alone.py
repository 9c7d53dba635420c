"""The orientation stage (filterKeypoints inside the batch, params.orient) on the bench's `modes` content: every second frame
uniform noise.  Prints the step time with and without the stage, k_orient_survivors' time by HIP events and the survivor /
oriented-point totals, so that instructions per survivor can be read off a PMC pass of the same command.

    [VSLAM_MX=1] python tools/orient_alone.py [--frames 256] [--steps 2]
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from visualslam_amd import capi, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--kernel", default="k_orient_survivors")
    ap.add_argument("--describe", type=int, default=0, help="1: the SIFT descriptors of every oriented point as well (params.describe)")
    a = ap.parse_args()
    capi.build()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    ctx = capi.Context(0, st.cuda_stream)
    n = a.frames
    frames = synth.frames_torch(n, a.rows, a.cols, stream_id=0, device=dev, noise_every=2)
    res = {"frames": n, "matrix_path": bool(ctx.matrix_path())}
    for orient in (0, 1):
        p = capi.default_params(a.rows, a.cols, n_octaves=4, localize=1, orient=orient)
        L = capi.batch_layout(p)
        out = dict(pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                   extrema_bits=torch.empty((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                   dog_points=torch.empty((n, p.dog_cap, 6), dtype=torch.int32, device=dev), dog_counts=torch.zeros(n, dtype=torch.int32, device=dev))
        if orient and a.describe:
            out.update(descriptors=torch.empty((n, p.oriented_cap, 128), dtype=torch.float32, device=dev),
                       descriptor_defined=torch.zeros((n, p.oriented_cap), dtype=torch.uint8, device=dev))
        if orient:
            out.update(oriented_points=torch.empty((n, p.oriented_cap, 6), dtype=torch.int32, device=dev),
                       oriented_counts=torch.zeros(n, dtype=torch.int32, device=dev), oriented_survivors=torch.zeros(n, dtype=torch.int32, device=dev))
        ctx.detect_batch(p, frames, **out)
        torch.cuda.synchronize()
        if orient:
            ctx.kernel_timing_enable(a.kernel)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ctx.detect_batch(p, frames, **out)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.steps
        if orient:
            launches, ms = ctx.kernel_timing_read()
            ctx.kernel_timing_enable(None)
            res.update(step_ms_orient=dt * 1e3, kernel=a.kernel, kernel_launches_per_step=launches / a.steps, kernel_ms_per_step=ms / a.steps,
                       keypoints=int(out["dog_counts"].sum()), survivors=int(out["oriented_survivors"].sum()), oriented=int(out["oriented_counts"].sum()))
        else:
            res.update(step_ms_localize=dt * 1e3)
        del out
    print(json.dumps(res))
    ctx.close()


if __name__ == "__main__":
    main()
