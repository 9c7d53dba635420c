"""PCIe-inclusive throughput of the batched path from Python: frames start in pinned host memory and the
keypoint lists end there.  Double-buffered: the upload of batch k+1 and the download of batch
k-1's lists overlap the kernels of batch k (three streams).  Since round 3 the lists are packed on the
device (vslam_pack_lists_dev) and only the records that exist are downloaded.  Prints one JSON line.
The C++ form of this pipeline is visualslam_amd/cxx/batch_detector.hpp (`Stream --mode hostfed`).

    python tools/bench_hostfed.py [--frames 256 --steps 40]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualslam_amd import capi, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    a = ap.parse_args()
    n, rows, cols = a.frames, a.rows, a.cols
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    compute = torch.cuda.Stream()
    up, down = torch.cuda.Stream(), torch.cuda.Stream()
    ctx = capi.Context(0, compute.cuda_stream)
    p = capi.default_params(rows, cols)
    L = capi.batch_layout(p)
    host_frames = synth.frames_torch(n, rows, cols).pin_memory()
    # the lists come back packed: sum over frames of min(count, cap) records, against a per-batch budget
    hk_budget, dp_budget = n * (1 << 17), n * (1 << 17)
    bufs = []
    for _ in range(2):
        bufs.append(dict(
            frames=torch.empty((n, rows, cols), dtype=torch.uint8, device=dev),
            harris_kps=torch.empty((n, p.harris_cap, 3), dtype=torch.int32, device=dev),
            harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
            dog_points=torch.empty((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
            dog_counts=torch.zeros(n, dtype=torch.int32, device=dev),
            d_kps=torch.empty((hk_budget, 3), dtype=torch.int32, device=dev), d_pts=torch.empty((dp_budget, 6), dtype=torch.int32, device=dev),
            d_off=torch.zeros((2, n + 1), dtype=torch.int64, device=dev),
            h_kps=torch.empty((hk_budget, 3), dtype=torch.int32).pin_memory(),
            h_pts=torch.empty((dp_budget, 6), dtype=torch.int32).pin_memory(),
            h_off=torch.zeros((2, n + 1), dtype=torch.int64).pin_memory(),
            up_done=torch.cuda.Event(), comp_done=torch.cuda.Event(), down_done=torch.cuda.Event()))
    shared = dict(response=torch.empty((n, rows, cols), dtype=torch.float32, device=dev),
                  nms_mask=torch.empty((n, rows, cols), dtype=torch.uint8, device=dev),
                  pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                  extrema_bits=torch.empty((n, L.bits_frame_words), dtype=torch.int64, device=dev))

    def upload(b):
        with torch.cuda.stream(up):
            up.wait_event(b["comp_done"])  # the previous use of this device frame buffer has been consumed
            b["frames"].copy_(host_frames, non_blocking=True)
            b["up_done"].record(up)

    def run(b):
        compute.wait_event(b["up_done"])
        compute.wait_event(b["down_done"])  # its list buffers have been downloaded
        with torch.cuda.stream(compute):
            ctx.detect_batch(p, b["frames"], harris_kps=b["harris_kps"], harris_counts=b["harris_counts"],
                             dog_points=b["dog_points"], dog_counts=b["dog_counts"], **shared)
            ctx.pack_lists(b["harris_kps"], b["harris_counts"], b["d_kps"], b["d_off"][0])
            ctx.pack_lists(b["dog_points"], b["dog_counts"], b["d_pts"], b["d_off"][1])
            b["h_off"].copy_(b["d_off"], non_blocking=True)
            b["comp_done"].record(compute)

    def download(b):
        b["comp_done"].synchronize()  # the offsets are on the host: only the records that exist travel
        nh, nd = min(int(b["h_off"][0, n]), hk_budget), min(int(b["h_off"][1, n]), dp_budget)
        with torch.cuda.stream(down):
            b["h_kps"][:nh].copy_(b["d_kps"][:nh], non_blocking=True)
            b["h_pts"][:nd].copy_(b["d_pts"][:nd], non_blocking=True)
            b["down_done"].record(down)
        return nh, nd

    for b in bufs:
        b["comp_done"].record(compute)
        b["down_done"].record(down)
    # warm-up + steady state: batch s+1 is uploaded and enqueued before the host waits for batch s
    for phase, steps in (("warm", 3), ("timed", a.steps)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        upload(bufs[0])
        run(bufs[0])
        for s in range(steps):
            cur, nxt = bufs[s % 2], bufs[(s + 1) % 2]
            if s + 1 < steps:
                nxt["down_done"].synchronize()
                upload(nxt)
                run(nxt)
            nh, nd = download(cur)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    up_bytes = n * rows * cols
    down_bytes = nh * 12 + nd * 24 + 16 * (n + 1)
    print(json.dumps({"mode": "host-fed from Python, double-buffered, packed lists downloaded", "frames_per_batch": n, "steps": a.steps,
                      "frames_per_sec": n * a.steps / dt, "ms_per_batch": dt / a.steps * 1e3,
                      "upload_MB_per_batch": up_bytes / 1e6, "download_MB_per_batch": down_bytes / 1e6,
                      "harris_records_per_batch": nh, "dog_records_per_batch": nd}))


if __name__ == "__main__":
    main()
