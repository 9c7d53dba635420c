#!/usr/bin/env python3
"""Headline benchmark: Harris + DoG keypoint detection on 1920x1080 frames (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the fused hot path (vslam_detect_batch_dev) over one batch of F
synthetic 1080p frames per GPU (BASELINE config 4: F = 256).  Frames are generated on the
device and are resident in HBM before the timed region.  Frames shard across ranks (one
camera stream per GPU, weak scaling); the only collective is the RCCL all-gather of the
per-rank keypoint counts.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s achievable

# Algorithmic HBM bytes per frame attributed to each kernel of the batch (DESIGN.md
# "Kernels"): the API-contract bytes the kernel is the one to read/write:
# inputs read once, every returned image written once, intermediates not counted.  The
# entries sum to 6N + 11*sumP = 133,617,600 B per 1080p frame.
def kernel_algorithmic_bytes(L, rows, cols):
    N = rows * cols
    P = [L.rows[o] * L.cols[o] for o in range(L.n_octaves)]
    return {
        "k_harris_strip": 6 * N,                 # u8 frame in, f32 response + u8 NMS mask out (one pass)
        "k_resize_linear2x_slide": N,               # DoG path's read of the frame
        "k_pyr_octave": 11 * sum(P[:2]),         # 6 Gaussian + 5 DoG images of octaves 0-1 (LDS-tiled)
        "k_gauss_h_strip": 11 * sum(P[2:]),      # the same for the coarse octaves (strip kernels)
    }


def _cpu_frames(rows, cols, n_oct, n, stream_id):
    """Oracle on n frames of one synthetic stream (worker of the all-cores baseline)."""
    import oracle
    from visualslam_amd import synth

    kp = 0
    for f in range(n):
        img = synth.frame_np(rows, cols, f, stream_id)
        R = oracle.harris_response(img)
        oracle.nms_strict(oracle.convert_scale_abs(R), 3)
        kp += len(oracle.harris_keypoints(oracle.nms2(R, 5)[0]))
        p = oracle.Pyramid(img, n_oct, 1.6)
        for o in range(n_oct):
            kp += len(p.extrema(o, 3, 8)[1])
        p.close()
    return kp


def cpu_baseline(rows, cols, n_oct, sample_frames):
    """Time the CPU oracle (single thread, like the reference) on a bounded sample."""
    import numpy as np

    import oracle
    from visualslam_amd import synth

    oracle.build()
    frames = synth.frames_np(sample_frames, rows, cols, stream_id=0)
    kp = 0
    t0 = time.perf_counter()
    for f in range(sample_frames):
        R = oracle.harris_response(frames[f])
        oracle.nms_strict(oracle.convert_scale_abs(R), 3)
        kp += len(oracle.harris_keypoints(oracle.nms2(R, 5)[0]))
        p = oracle.Pyramid(frames[f], n_oct, 1.6)
        for o in range(n_oct):
            kp += len(p.extrema(o, 3, 8)[1])
        p.close()
    dt = time.perf_counter() - t0
    # secondary figure (BASELINE.md section 3 ii): all host cores, independent worker processes
    # (plain subprocesses with a hard timeout; the parent holds a HIP context, so no fork)
    allcores = None
    try:
        import subprocess

        workers = min(os.cpu_count() or 1, 64)
        if workers > 1:
            per = 2
            code = ("import sys; sys.path.insert(0, %r); import bench; "
                    "print(bench._cpu_frames(%d, %d, %d, %d, int(sys.argv[1])))" % (ROOT, rows, cols, n_oct, per))
            t1 = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, "-c", code, str(w)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
                     for w in range(workers)]
            ok = 0
            for pr in procs:
                try:
                    pr.communicate(timeout=max(5.0, 180.0 - (time.perf_counter() - t1)))
                    ok += pr.returncode == 0
                except subprocess.TimeoutExpired:
                    pr.kill()
            d2 = time.perf_counter() - t1
            if ok == workers:
                allcores = {"value": workers * per / d2, "unit": "frames/s", "cores": workers,
                            "sample": f"{workers * per} frames, {workers} processes x {per} frames (includes interpreter start-up)"}
            else:
                allcores = {"error": f"{workers - ok} of {workers} workers failed or timed out"}
    except Exception as e:  # the single-thread figure is the reported baseline
        allcores = {"error": repr(e)}
    return {
        "all_cores": allcores,
        "value": sample_frames / dt,
        "unit": "frames/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{sample_frames} synthetic {cols}x{rows} frames, Harris+NMS+DoG pyramid+extrema, oracle/vslam_oracle.c -O2, 1 thread",
        "keypoints_per_sec": kp / dt,
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step (BASELINE config 4: 256)")
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--octaves", type=int, default=4)
    ap.add_argument("--kernel", default=None, help="kernel to time with HIP events for the roofline object")
    ap.add_argument("--localize", type=int, default=0,
                    help="1: DoG list = FeaturePointLocalization survivors (SURVEY 8f row 2) instead of the contrast-8 candidate list")
    ap.add_argument("--orient", type=int, default=0,
                    help="1: also run filterKeypoints on every frame's keypoint list (SURVEY 8f row 3); implies --localize 1")
    ap.add_argument("--cpu-sample", type=int, default=6, help="frames in the CPU baseline sample (0 = skip)")
    ap.add_argument("--stream", choices=["side", "null"], default="side",
                    help="stream of the whole job: a torch side stream (default) or torch's default (NULL) stream")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from visualslam_amd import capi, sharding, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU fallback")
    # Rehearsal switches (not used by the driver): VSLAM_BENCH_BACKEND=gloo runs the collectives of
    # the N > 1 path over gloo on host copies, VSLAM_BENCH_SHARE_GPU=1 puts every rank on cuda:0 -
    # together they exercise the multi-rank code on a one-GPU box (RCCL refuses two ranks per GPU).
    backend = os.environ.get("VSLAM_BENCH_BACKEND", "nccl")
    if os.environ.get("VSLAM_BENCH_SHARE_GPU") == "1":
        local_rank_dev = 0
    else:
        local_rank_dev = local_rank
    torch.cuda.set_device(local_rank_dev)
    dev = torch.device("cuda", local_rank_dev)
    use_dist = world > 1 or "RANK" in os.environ  # under torchrun even a single rank goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        print(f"note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    # one rank per node compiles (a no-op when the in-tree .so is current); the others wait
    if local_rank == 0:
        capi.build()
    if use_dist:
        dist.barrier()
    capi.build()
    rows, cols, n = args.rows, args.cols, args.frames
    # One stream for the whole job: frame synthesis, the detection kernels, the count sums and the
    # collective's input all run on it, so the stream alone orders them (capi.Context launches on
    # the handle it is given; handle 0 = the NULL stream, passed on as VSLAM_STREAM_LEGACY).
    job_stream = torch.cuda.Stream(device=dev) if args.stream == "side" else torch.cuda.default_stream(dev)
    torch.cuda.set_stream(job_stream)
    ctx = capi.Context(local_rank_dev, torch.cuda.current_stream().cuda_stream)
    p = capi.default_params(rows, cols, n_octaves=args.octaves, localize=1 if args.orient else args.localize, orient=args.orient)
    L = capi.batch_layout(p)

    # one camera stream per GPU: stream_id = rank
    frames = synth.frames_torch(n, rows, cols, stream_id=rank, device=dev)
    out = dict(
        response=torch.empty((n, rows, cols), dtype=torch.float32, device=dev),
        nms_mask=torch.empty((n, rows, cols), dtype=torch.uint8, device=dev),
        harris_kps=torch.empty((n, p.harris_cap, 3), dtype=torch.int32, device=dev),
        harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
        pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
        extrema_bits=torch.empty((n, L.bits_frame_words), dtype=torch.int64, device=dev),
        dog_points=torch.empty((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
        dog_counts=torch.zeros(n, dtype=torch.int32, device=dev),
    )
    if args.orient:
        out["oriented_points"] = torch.empty((n, p.oriented_cap, 6), dtype=torch.int32, device=dev)
        out["oriented_counts"] = torch.zeros(n, dtype=torch.int32, device=dev)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live
    counts_local = torch.zeros(2, dtype=torch.int64, device=dev)
    counts_all = torch.zeros((world, 2), dtype=torch.int64, device=cdev)

    def step():
        ctx.detect_batch(p, frames, **out)
        counts_local[0] = out["harris_counts"].sum()
        counts_local[1] = out["dog_counts"].sum()
        sharding.gather_counts(counts_local.to(cdev), counts_all)  # the one collective of the path: 16 B per rank over RCCL

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    algo = kernel_algorithmic_bytes(L, rows, cols)
    kname = args.kernel or "k_pyr_octave"
    fence()
    ctx.kernel_timing_enable(kname)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    launches, kms = ctx.kernel_timing_read()
    ctx.kernel_timing_enable(None)

    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    totals = counts_all.sum(0).tolist()
    overflow = bool((out["harris_counts"] > p.harris_cap).any() or (out["dog_counts"] > p.dog_cap).any())

    if rank == 0:
        total_frames = n * world * args.steps
        fps = total_frames / dt
        kp_per_step = totals[0] + totals[1]
        bytes_frame = L.algorithmic_bytes_harris + L.algorithmic_bytes_dog - rows * cols  # fused: input counted once
        roof = None
        if launches and kms > 0:
            ach = algo.get(kname, 0) * n * args.steps / (kms * 1e-3) / 1e9
            # measured HBM bytes per launch from the committed PMC passes (profiles/traffic.json,
            # made by tools/pmc_summary.py), scaled to this run's frames per launch
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                t = json.load(open(tpath)).get(kname)
                if t:
                    traffic = t["hbm_bytes_per_frame"] * n * args.steps / launches
            roof = {
                "kernel": kname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS, "traffic": traffic,
                "launches": launches, "avg_launch_ms": kms / launches,
                "algorithmic_bytes_per_frame": algo.get(kname, 0),
                "algorithmic_bytes_per_launch": algo.get(kname, 0) * n * args.steps / launches,
            }
        # secondary view of the same kernel: it is bound by VALU issue of the packed dot
        # instructions, not by HBM.  Algorithmic dot instructions per pixel =
        # sum over levels of n/4 (v_dot4_u32_u8, vertical) + n/2 (v_dot2_u32_u16, horizontal) with
        # the zero-trimmed kernel widths n; peak = 32.8e12 lane-instr/s measured by
        # tools/ubench_valu.hip (4.3 cycles per wave64 dot instruction per SIMD).
        valu = None
        if roof and kname == "k_pyr_octave" and args.octaves >= 2 and (rows, cols) == (1080, 1920):
            widths = [[9, 13, 15, 19, 23, 29], [19, 23, 29, 37, 45, 57]]
            per_frame = sum(0.75 * sum(w) * L.rows[o] * L.cols[o] for o, w in enumerate(widths))
            ach = per_frame * n * args.steps / (kms * 1e-3)
            valu = {"bound": "valu-dot", "achieved": ach / 1e12, "peak": 32.8, "unit": "T lane-instr/s",
                    "frac": ach / 32.8e12, "algorithmic_dot_instr_per_frame": per_frame}
        line = {
            "metric": "frames/sec @1080p (Harris + DoG keypoint detection)",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"batch of {n} synthetic {cols}x{rows} frames per GPU, Harris(k=0.04)+NMS and DoG pyramid "
                            f"{args.octaves} octaves x (6 Gaussian, 5 DoG) + extrema, fused (BASELINE config 4)",
                "frames_per_gpu": n, "rows": rows, "cols": cols, "octaves": args.octaves, "localize": p.localize, "orient": args.orient,
                "parallelism": f"frames sharded 1 stream/GPU x{world}; RCCL all-gather of counts only",
            },
            "keypoints_per_sec": kp_per_step * args.steps / dt,
            "keypoints_per_step": {"harris": totals[0], "dog": totals[1], "list_overflow": overflow,
                                   **({"oriented_rank0": int(out["oriented_counts"].sum())} if args.orient else {})},
            "pipeline_hbm": {
                "algorithmic_bytes_per_frame": bytes_frame,
                "achieved_GBps": bytes_frame * fps / world / 1e9,
                "frac_of_peak": bytes_frame * fps / world / 1e9 / HBM_PEAK_GBPS,
            },
            "roofline": roof,
            "roofline_valu": valu,
            "cpu_baseline": cpu_baseline(rows, cols, args.octaves, args.cpu_sample) if (world == 1 and args.cpu_sample > 0) else None,
        }
        print(json.dumps(line))
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
