"""The LDS-tiled octave kernels with the chip to themselves: pyramid-only batches (no Harris chain, no scan, no lists),
dot kernels or - with VSLAM_MX=1 - the opt-in matrix-core kernels.  Prints the per-launch time of the octave kernel
(HIP events on the launch stream) and the whole step.

    [VSLAM_MX=1] python tools/mx_alone.py [--frames 256] [--steps 5] [--octaves 2]
"""
import argparse, json, os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from visualslam_amd import capi, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--octaves", type=int, default=2)
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    a = ap.parse_args()
    capi.build()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(st)
    ctx = capi.Context(0, st.cuda_stream)
    frames = synth.frames_torch(a.frames, a.rows, a.cols, stream_id=0, device=dev)
    p = capi.default_params(a.rows, a.cols, n_octaves=a.octaves)
    L = capi.batch_layout(p)
    pyr = torch.empty((a.frames, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev)
    kname = "k_pyr_octave_mx" if ctx.matrix_path() else "k_pyr_octave"
    for _ in range(2):
        ctx.detect_batch(p, frames, pyramid=pyr)
    torch.cuda.synchronize()
    ctx.kernel_timing_enable(kname)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ctx.detect_batch(p, frames, pyramid=pyr)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    launches, ms = ctx.kernel_timing_read()
    ctx.kernel_timing_enable(None)
    alg = sum(11 * L.rows[o] * L.pitch[o] for o in range(min(2, a.octaves))) * a.frames
    print(json.dumps({"kernel": kname, "dbg": os.environ.get("VSLAM_MX_DBG", "0"), "frames": a.frames, "octaves": a.octaves, "launches_per_step": launches / a.steps,
                      "octave_kernel_ms_per_step": ms / a.steps, "step_ms": dt * 1e3, "alg_TBps": alg / (ms / a.steps) / 1e9}))


if __name__ == "__main__":
    main()
