#!/usr/bin/env python3
"""Headline benchmark: Harris + DoG keypoint detection on 1920x1080 frames (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames F]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Outside torchrun `--gpus N` with N > 1 starts the N ranks itself (a torch.distributed.run child job, launch_ranks) and
relays rank 0's line: both forms measure BASELINE config 5.

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

# Algorithmic HBM bytes per frame attributed to each kernel of the batch (DESIGN.md "Kernels"): the
# API-contract bytes the kernel is the one to read / write: inputs read once, every
# returned image written once, intermediates not counted.  The DoG path's read of the frame belongs to
# the upsample kernel, so the entries sum to 6N + N + 11*sumP = 135,691,200 B per 1080p frame; the
# fused figure of BASELINE config 4 counts the frame once (6N + 11*sumP = 133,617,600 B) and is what
# `pipeline_hbm` reports.
def kernel_algorithmic_bytes(L, rows, cols):
    N = rows * cols
    P = [L.rows[o] * L.cols[o] for o in range(L.n_octaves)]
    return {
        "k_harris_strip": 6 * N,                 # u8 frame in, f32 response + u8 NMS mask out (one pass)
        "k_resize_linear2x_slide": N,            # DoG path's read of the frame
        "k_pyr_octave": 11 * sum(P[:2]),         # 6 Gaussian + 5 DoG images of octaves 0-1 (LDS-tiled)
        "k_pyr_octave_mx": 11 * sum(P[:4]),      # OPT-IN matrix path: the same images of octaves 0-3, one kernel family
        "k_gauss_band": 11 * sum(P[2:]),         # the same for the coarse octaves (fused band kernel)
        "k_gauss_h_strip": 11 * sum(P[2:]),      # ... or the two strip kernels, where a band does not fit the LDS
    }


def gauss_trimmed_width(capi, sigma):
    """Width of the Gaussian kernel of one pyramid level after dropping its zero outer taps."""
    t = capi.gauss_taps_q8(capi.gauss_ksize_u8(sigma), sigma)
    nz = t.nonzero()[0]
    return int(nz[-1] - nz[0] + 1)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cpu_baseline(rows, cols, n_oct, sample_frames, gpu_keypoints=None):
    """Time the CPU oracle on a bounded sample of the same workload: 1 thread (the reference is
    single-threaded: the reported baseline) and all host CPUs of this box's share with OpenMP over
    frames inside the oracle (SURVEY 8d ii)."""
    import oracle
    from visualslam_amd import synth

    oracle.build()
    frames = synth.frames_np(sample_frames, rows, cols, stream_id=0)
    t0 = time.perf_counter()
    kp = oracle.baseline_frames(frames, n_oct, threads=1)
    dt = time.perf_counter() - t0
    # ALL host cores.  The box may give this process fewer CPUs than it shows (affinity mask, cgroup
    # quota): both are reported, the run uses one thread per CPU of the affinity mask, and the 64-thread figure the
    # earlier rounds reported stays beside it.
    allcores = None
    try:
        affinity = len(os.sched_getaffinity(0))
        quota = None
        try:
            q = open("/sys/fs/cgroup/cpu.max").read().split()
            if q and q[0] != "max":
                quota = float(q[0]) / float(q[1])
        except (OSError, ValueError, IndexError):
            pass
        if affinity > 1:
            base_n = min(2 * 64, 2 * affinity)
            many = synth.frames_np(base_n, rows, cols, stream_id=0)

            def run(threads, nfr):
                fr = many if nfr == len(many) else np.concatenate([many] * ((nfr + len(many) - 1) // len(many)))[:nfr]
                t1 = time.perf_counter()
                oracle.baseline_frames(fr, n_oct, threads=threads)
                d2 = time.perf_counter() - t1
                return {"value": nfr / d2, "unit": "frames/s", "cores": threads,
                        "sample": f"{nfr} frames, OpenMP over frames in oracle/vslam_oracle.c ({threads} threads)"}

            import numpy as np

            # the CPUs this process can really keep busy: its affinity mask, cut down to the cgroup's CPU quota where one is
            # set (the GPU box shows 256 CPUs and grants 16 CPUs' worth of time: 256 threads then ran SLOWER than 64,
            # 7.1 against 12.2 frames/s, and took 55 s)
            usable = affinity if not quota else max(1, min(affinity, int(quota + 0.999)))
            allcores = run(usable, 2 * usable if usable <= 64 else usable)  # > 64 threads: one frame each (bounded run time)
            allcores["affinity_cpus"] = affinity
            allcores["cgroup_cpu_quota"] = quota
            allcores["what"] = "threads = CPUs usable by this process (affinity mask, limited to the cgroup CPU quota)"
            if usable != 64 and affinity >= 64:
                allcores["threads_64"] = run(64, 128)  # the figure rounds 2-4 reported
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
        "threads_used": {"value": 1, "all_cores": (allcores or {}).get("cores")},
        "cpu_model": cpu_model(),
        # the same frames give the same keypoint counts on both paths (Harris list + DoG list
        # with value >= 8 of the sample's frames, which are the first frames of rank 0's batch)
        "keypoints": int(kp),
        "keypoints_gpu_same_frames": gpu_keypoints,
        "keypoints_equal_gpu": (int(kp) == int(gpu_keypoints)) if gpu_keypoints is not None else None,
    }


def live_traffic(kname, rows, cols, octaves, frames=64, matrix_path=0):
    """HBM bytes per frame of kernel `kname`, MEASURED for this run's build on this box: two child runs of this
    script under rocprofv3 (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, nothing else traced), a small
    batch of the same frames; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 (it counts 64 B of every
    128-B request), both reported in KiB.  None when rocprofv3 is missing or a pass fails (the caller then falls back
    to the committed profile and says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not prof:
        return None
    sums, launches, step_sums = {}, 0, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                cmd = [prof, "--pmc", counter, "--output-format", "csv", "-d", td, "-o", "r", "--", sys.executable, os.path.join(ROOT, "bench.py"),
                       "--frames", str(frames), "--steps", "1", "--warmup", "1", "--cpu-sample", "0", "--modes", "0", "--cxx-host", "0", "--live-traffic", "0", "--mx", "0", "--roofline-pass", "0",
                       "--matrix-path", str(int(matrix_path)), "--rows", str(rows), "--cols", str(cols), "--octaves", str(octaves)]
                env = dict(os.environ, TMPDIR="/tmp")
                for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                    env.pop(k, None)
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd="/tmp", env=env)
                files = glob.glob(td + "/**/*_counter_collection.csv", recursive=True)
                if r.returncode != 0 or not files:
                    return None
                tot, n_l, tot_all = 0.0, 0, 0.0
                for f in files:
                    for row in csv.DictReader(open(f)):
                        if row["Counter_Name"] != counter:
                            continue
                        kn = row["Kernel_Name"]
                        if kname in kn:
                            tot += float(row["Counter_Value"])
                            n_l += 1
                        # every kernel of the library (all vslam::k_*; torch's own kernels - frame synthesis, fills - are not the step)
                        if "vslam::k_" in kn or kn.startswith("k_"):
                            tot_all += float(row["Counter_Value"])
                if n_l == 0:
                    return None
                sums[counter], launches, step_sums[counter] = tot, n_l, tot_all
    except Exception:
        return None
    batches = 2  # warm-up + the step
    bytes_total = (2.0 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024.0
    step_total = (2.0 * step_sums["FETCH_SIZE"] + step_sums["WRITE_SIZE"]) * 1024.0
    return {"hbm_bytes_per_frame": bytes_total / (frames * batches), "frames_per_batch": frames, "batches": batches, "launches": launches,
            "fetch_KiB": sums["FETCH_SIZE"], "write_KiB": sums["WRITE_SIZE"],
            "step_hbm_bytes_per_frame": step_total / (frames * batches)}  # all k_* kernels of the step together


def cxx_host_runs(rows, cols, n, octaves, rank=0, world=1, local_rank=0):
    """frames/s of the C++ throughput host on this workload: visualslam_amd/bin/Stream, one process per rank, the
    counts all-gathered through RCCL's C API.  With world > 1 every rank of this job starts one Stream process with
    its own RANK / LOCAL_RANK (so the C++ processes form their own RCCL communicator over the same GPUs); rank 0's
    child prints the job's line.  Device-resident mode always; host-fed only at N = 1 (it needs the host's memory
    bandwidth to itself to mean anything)."""
    import subprocess

    exe = os.path.join(ROOT, "visualslam_amd", "bin", "Stream")
    if not os.path.exists(exe):
        r = subprocess.run(["make", "-C", os.path.join(ROOT, "visualslam_amd", "cxx")], capture_output=True, text=True)
        if r.returncode != 0:
            return {"error": "building the C++ host failed: " + (r.stdout + r.stderr)[-400:]}
    port = int(os.environ.get("MASTER_PORT", "29533")) + 7  # the rendezvous of the C++ ranks (MASTER_PORT itself is torch's store)
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank), MASTER_ADDR="127.0.0.1", VSLAM_RDV_PORT=str(port))
    if world > 1:  # a scaling run must not hang on a secondary figure: 20 s to meet, 45 s per child, one mode
        env.setdefault("VSLAM_RDV_TIMEOUT_MS", "20000")
    env.pop("VSLAM_COUNT_BACKEND", None)
    if os.environ.get("VSLAM_BENCH_SHARE_GPU") == "1":  # rehearsal on one GPU (tests/test_bench_ranks.py): RCCL refuses two ranks per device
        env["VSLAM_COUNT_BACKEND"] = "tcp"
    res = {}
    # hostfed: the lists reach the host as packed 16-byte records (vslam_point16); hostfed_expanded: the same run with 8 host
    # threads rebuilding every frame's 24-byte SLAM::point records inside the timed loop (what a consumer of the
    # reference's std::vector<SLAM::point> gets)
    for key in (("device", "hostfed", "hostfed_expanded") if world == 1 else ("device",)):
        mode = "hostfed" if key.startswith("hostfed") else key
        extra = ["--expand", "8"] if key == "hostfed_expanded" else []
        try:
            r = subprocess.run([exe, *extra, "--mode", mode, "--frames", str(n), "--batches", ("30" if mode == "device" else "40") if world == 1 else "12", "--warmup", "6", "--rows", str(rows), "--cols", str(cols),
                                "--octaves", str(octaves)], capture_output=True, text=True, timeout=600 if world == 1 else 45, env=env)
            if r.returncode != 0:
                res[key] = {"error": (r.stdout + r.stderr)[-400:]}
            elif rank == 0:
                res[key] = json.loads(r.stdout.strip().splitlines()[-1])
            else:
                res[key] = {"rank": rank, "ok": True}
        except Exception as e:
            res[key] = {"error": repr(e)}
    return res


def launch_ranks(n, argv):
    """BASELINE config 5 from the plain command: `python bench.py --gpus N ...` outside torchrun starts
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same args>` as a CHILD process (never an
    exec: this process has not touched the GPU and never will), relays rank 0's one JSON line and the child's exit code,
    and fails when the line does not show N ranks gathered."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("OMP_NUM_THREADS", "1")  # what torchrun sets itself, without its warning
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"bench.py: --gpus {n} outside torchrun: starting {n} ranks as a child job (port {port})", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    line = None
    for out in child.stdout:  # stderr goes straight through; stdout is relayed as it comes, the JSON line kept
        sys.stdout.write(out)
        sys.stdout.flush()
        if out.startswith("{"):
            line = out
    rc = child.wait()
    if rc != 0:
        print(f"bench.py: the {n}-rank child job exited with {rc}", file=sys.stderr)
        return rc
    try:
        d = json.loads(line)
        got = d["distributed"]["ranks_gathered"]
    except Exception:
        print("bench.py: the child job printed no JSON line", file=sys.stderr)
        return 3
    if got != n or d.get("n_gpus") != n:
        print(f"bench.py: asked for {n} ranks, the line shows ranks_gathered={got} n_gpus={d.get('n_gpus')}", file=sys.stderr)
        return 4
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=6)  # >= 5: the library times its 2nd..5th batch call on candidate side streams
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
    ap.add_argument("--modes", type=int, default=1, help="1: also time the localize / orient list modes (the `modes` object)")
    ap.add_argument("--live-traffic", type=int, default=1,
                    help="1: measure roofline.traffic in this run (two rocprofv3 --pmc child passes on a 64-frame batch, N = 1 only); 0: the committed profile's figure")
    ap.add_argument("--roofline-pass", type=int, default=1, help="0: skip the hooked pass behind `value` (the PMC child passes: exactly warm-up + one step)")
    ap.add_argument("--cxx-host", type=int, default=1, help="1: also run the C++ Stream executable on the same workload (one process per rank, after the measurement)")
    ap.add_argument("--matrix-path", type=int, default=0,
                    help="PROFILING ONLY: 1 runs the timed region itself on the opt-in matrix-core kernels (vslam_ctx_set_matrix_path); the line then says so "
                         "in config.matrix_path and is not the headline.  The default run reports that path beside `value` as `mx_path`.")
    ap.add_argument("--mx", type=int, default=1, help="1: also time the opt-in matrix-core path (the `mx_path` object; never `value`)")
    ap.add_argument("--secondary-at-scale", type=int, default=0, help="1: run `modes` / `two_in_flight` / `mx_path` also at N > 1 (default: N = 1 only)")
    ap.add_argument("--side-level", type=int, default=-1,
                    help="A/B: pin the library's side-stream level (vslam_ctx_pin_side_streams: 0 yielding, 1 the context stream's priority, 2 no side streams); -1 = the library's default + watchdog")
    ap.add_argument("--stream", choices=["side", "null"], default="side",
                    help="stream of the whole job: a torch side stream (default) or torch's default (NULL) stream")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # the plain form `python bench.py --gpus N`: this process becomes the launcher (nothing here has touched the GPU)
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    from visualslam_amd import capi, sharding, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # N = 1, not under torchrun: the C++ host's two runs come FIRST, before this process has a GPU context - with the
    # Python process's idle context and queues beside it the host-fed child ran 4 % slower than the same command alone
    # (12.9 k against 13.4 k steady).  Secondary figures either way; the order of the legs is in bench_wall_s.
    cxx_early, cxx_early_s = None, 0.0
    if world == 1 and "RANK" not in os.environ and args.cxx_host:
        capi.build()
        t_cxx = time.perf_counter()
        cxx_early = cxx_host_runs(args.rows, args.cols, args.frames, args.octaves)
        cxx_early_s = time.perf_counter() - t_cxx
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
    # who is really in the job: every rank's id and the identity of its GPU, gathered through the job's own backend
    # (ranks_gathered / n_gpus of the line are COUNTED from this, not copied from WORLD_SIZE)
    try:
        pr = torch.cuda.get_device_properties(dev)
        my_gpu = {"rank": rank, "device_index": local_rank_dev, "name": pr.name, "uuid": str(getattr(pr, "uuid", "")),
                  "pci_bus_id": getattr(pr, "pci_bus_id", None), "pci_device_id": getattr(pr, "pci_device_id", None)}
    except Exception as e:
        my_gpu = {"rank": rank, "device_index": local_rank_dev, "error": repr(e)}
    members = [my_gpu]
    if use_dist:
        members = [None] * world
        dist.all_gather_object(members, my_gpu)
    ranks_gathered = len({m["rank"] for m in members if m})

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
    ctx.set_matrix_path(bool(args.matrix_path))  # the headline runs on the default (MFMA-free) kernels whatever VSLAM_MX says
    if args.side_level >= 0:
        ctx.pin_side_streams(args.side_level)
    wall = {}  # seconds per leg of this script (rank 0's clock)
    t_leg = [time.perf_counter()]

    def leg(name):
        now = time.perf_counter()
        wall[name] = wall.get(name, 0.0) + now - t_leg[0]
        t_leg[0] = now

    secondary = world == 1 or bool(args.secondary_at_scale)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the collectives' tensors live
    # one camera stream per GPU: stream_id = rank
    frames = synth.frames_torch(n, rows, cols, stream_id=rank, device=dev)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    shared = {}  # the big output buffers are shared by the modes (the pyramids alone are 31 GB)

    def run_mode(localize, orient, steps, warmup, kname, describe=0, dense=0, frames_t=None):
        """K timed steps of one list mode; returns per-mode results (times: max over ranks)."""
        frames_t = frames if frames_t is None else frames_t
        p = capi.default_params(rows, cols, n_octaves=args.octaves, localize=1 if orient else localize, orient=orient, extrema_dense=dense)
        if dense:  # every pixel of three levels is a site: a noise frame has ~4x the candidates of the lattice test's worst case
            p.dog_cap = 4 * p.dog_cap
        L = capi.batch_layout(p)
        if dense:  # the dense scan has its own (larger) bitmask; every other buffer is shared
            if "dense_bits" not in shared:
                shared["dense_bits"] = torch.empty((n, L.bits_frame_words), dtype=torch.int64, device=dev)
        if "response" not in shared:
            shared.update(
                response=torch.empty((n, rows, cols), dtype=torch.float32, device=dev),
                nms_mask=torch.empty((n, rows, cols), dtype=torch.uint8, device=dev),
                harris_kps=torch.empty((n, p.harris_cap, 3), dtype=torch.int32, device=dev),
                harris_counts=torch.zeros(n, dtype=torch.int32, device=dev),
                pyramid=torch.empty((n, L.pyramid_frame_bytes), dtype=torch.uint8, device=dev),
                extrema_bits=torch.empty((n, L.bits_frame_words), dtype=torch.int64, device=dev),
                dog_points=torch.empty((n, p.dog_cap, 6), dtype=torch.int32, device=dev),
                dog_counts=torch.zeros(n, dtype=torch.int32, device=dev),
            )
        out = {k: v for k, v in shared.items() if k != "dense_bits"}
        if dense:
            out["extrema_bits"] = shared["dense_bits"]
            out["dog_points"] = torch.empty((n, p.dog_cap, 6), dtype=torch.int32, device=dev)
        if orient:
            out["oriented_points"] = torch.empty((n, p.oriented_cap, 6), dtype=torch.int32, device=dev)
            out["oriented_counts"] = torch.zeros(n, dtype=torch.int32, device=dev)
            out["oriented_survivors"] = torch.zeros(n, dtype=torch.int32, device=dev)
        if describe:  # 128 floats per oriented point at the full per-frame capacity (8.6 GB for 256 x 65536)
            out["descriptors"] = torch.empty((n, p.oriented_cap, 128), dtype=torch.float32, device=dev)
            out["descriptor_defined"] = torch.zeros((n, p.oriented_cap), dtype=torch.uint8, device=dev)
        counts_local = torch.zeros(2, dtype=torch.int64, device=dev)
        counts_all = torch.zeros((world, 2), dtype=torch.int64, device=cdev)

        def step():
            ctx.detect_batch(p, frames_t, **out)
            ctx.count_totals(out["harris_counts"], out["dog_counts"], counts_local)  # {harris, dog} of this rank, on the device (one small kernel)
            sharding.gather_counts(counts_local.to(cdev), counts_all)  # the one collective of the path: 16 B per rank over RCCL

        for _ in range(warmup):
            step()
        fence()
        if kname:
            ctx.kernel_timing_enable(kname)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        launches, kms = ctx.kernel_timing_read() if kname else (0, 0.0)
        ctx.kernel_timing_enable(None)
        # max over ranks of the time; sums over ranks of the list flags
        red = torch.tensor([dt], dtype=torch.float64, device=cdev)
        flags = torch.tensor([int((out["harris_counts"] > p.harris_cap).any() or (out["dog_counts"] > p.dog_cap).any()),
                              int(out["oriented_counts"].sum()) if orient else 0,
                              int((out["oriented_survivors"] > p.oriented_cap).any()) if orient else 0], dtype=torch.int64, device=cdev)
        if use_dist:
            dist.all_reduce(red, op=dist.ReduceOp.MAX)
            dist.all_reduce(flags, op=dist.ReduceOp.SUM)
        totals = counts_all.sum(0).tolist()
        fl = flags.tolist()
        if describe:
            del out["descriptors"], out["descriptor_defined"]
        return {"p": p, "L": L, "dt": float(red.item()), "launches": launches, "kms": kms, "harris": totals[0], "dog": totals[1],
                "list_overflow": bool(fl[0]), "oriented": fl[1], "oriented_truncated": bool(fl[2])}

    def kernel_alone(kname, steps):
        """The dominant kernel with nothing beside it: the same batch, pyramid output only (no Harris
        chain, no extrema scan, no lists).  Not part of the timed region of `value`."""
        p = capi.default_params(rows, cols, n_octaves=args.octaves)
        ctx.detect_batch(p, frames, pyramid=shared["pyramid"])
        fence()
        ctx.kernel_timing_enable(kname)
        for _ in range(steps):
            ctx.detect_batch(p, frames, pyramid=shared["pyramid"])
        fence()
        launches, kms = ctx.kernel_timing_read()
        ctx.kernel_timing_enable(None)
        return launches, kms

    def by_kernel(p, L, out, matrix_path, steps=2):
        """every major kernel of the step priced against the HBM roof in the driver's own line.  One extra
        pass of `steps` full steps PER KERNEL outside the timed region, the library's HIP-event hook (vslam_kernel_timing_*)
        around that kernel's launches only ("name@o" = the launches of octave o), on the stream each launch goes to: the
        durations are the in-step ones (the kernel beside whatever the schedule runs with it).  Algorithmic bytes per frame
        as SURVEY 8d attributes them (kernel_algorithmic_bytes); a scan's are the mask words it is there to write."""
        N = rows * cols
        P = [L.rows[o] * L.cols[o] for o in range(L.n_octaves)]
        mask_b = [3 * L.lat_rows[o] * L.lat_words[o] * 8 for o in range(L.n_octaves)]
        items = [("k_harris_strip", None, 6 * N, "frame in, f32 response + u8 NMS mask out (6N)")]
        if matrix_path:
            for o in range(min(L.n_octaves, 4)):
                items.append(("k_pyr_octave_mx", o, 11 * P[o] + (N if o == 0 else 0), "6 Gaussian + 5 DoG planes of the octave" + (" + the frame (fused x2 upsample, lattice scan inside)" if o == 0 else "")))
            for o in range(min(L.n_octaves, 2)):
                items.append(("k_extrema_pack", o, mask_b[o], "site bytes of the fused scan -> candidate mask words"))
            for o in range(L.n_octaves):
                items.append(("k_extrema_w3", o, mask_b[o] if o >= 2 else mask_b[o] // 32, "lattice scan: whole octave (o >= 2) or the straddling rows only"))
        else:
            items.append(("k_resize_linear2x_slide", None, N, "the DoG path's read of the frame (x2 bilinear upsample into the octave-0 base)"))
            for o in range(min(L.n_octaves, 2)):
                items.append(("k_pyr_octave", o, 11 * P[o], "6 Gaussian + 5 DoG planes of the octave"))
            for o in range(2, L.n_octaves):
                items.append(("k_gauss_v_strip", o, None, "vertical pass into the u16 scratch (intermediate: priced with the horizontal pass)"))
                items.append(("k_gauss_h_strip", o, 11 * P[o], "horizontal pass + DoG: 6 Gaussian + 5 DoG planes of the octave (frac is for v + h together)"))
            for o in range(L.n_octaves):
                items.append(("k_extrema_w3", o, mask_b[o], "lattice scan: candidate mask words (re-reads the DoG planes: traffic, not algorithmic bytes)"))
        items.append(("k_flag_scatter", None, None, "list records (Harris + DoG lists)"))
        res, last_v = [], {}
        for name, o, ab, what in items:
            ctx.kernel_timing_enable(name if o is None else f"{name}@{o}")
            for _ in range(steps):
                ctx.detect_batch(p, frames, **out)
            fence()
            nl, ms = ctx.kernel_timing_read()
            ctx.kernel_timing_enable(None)
            if nl == 0:
                continue
            e = {"kernel": name, "octave": o, "launches_per_step": nl / steps, "avg_launch_ms": ms / nl, "ms_per_step": ms / steps,
                 "algorithmic_bytes_per_frame": ab, "what": what}
            if name == "k_gauss_v_strip":
                last_v[o] = ms / steps
            t = ms / steps + (last_v.get(o, 0.0) if name == "k_gauss_h_strip" else 0.0)
            if ab and t > 0:
                e["achieved_GBps"] = ab * n / (t * 1e-3) / 1e9
                e["frac"] = e["achieved_GBps"] / HBM_PEAK_GBPS
            res.append(e)
        return res

    def config_legs(steps=3):
        """BASELINE configs 2 and 3 as batches of the same frames (the driver line's `value` is config 4, both fused):
        Harris + NMS only, and DoG pyramid + extrema only."""
        res = {}
        N = rows * cols
        try:
            p2 = capi.default_params(rows, cols, n_octaves=0)
            o2 = {k: shared[k] for k in ("response", "nms_mask", "harris_kps", "harris_counts")}
            ctx.detect_batch(p2, frames, **o2)
            fence()
            ctx.kernel_timing_enable("k_harris_strip")
            t0 = time.perf_counter()
            for _ in range(steps):
                ctx.detect_batch(p2, frames, **o2)
            fence()
            d = time.perf_counter() - t0
            nl, ms = ctx.kernel_timing_read()
            ctx.kernel_timing_enable(None)
            res["config2_harris_only"] = {
                "what": "BASELINE config 2 as a batch: Harris response + NMS mask + keypoint list, no pyramid", "steps": steps,
                "frames_per_sec": n * world * steps / d, "ms_per_step": d / steps * 1e3, "algorithmic_bytes_per_frame": 6 * N,
                "pipeline_frac_of_peak": 6 * N * (n * steps / d) / 1e9 / HBM_PEAK_GBPS,
                "k_harris_strip": {"avg_launch_ms": ms / max(nl, 1), "achieved_GBps": 6 * N * n * steps / (ms * 1e-3) / 1e9 if ms > 0 else None,
                                   "frac": 6 * N * n * steps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if ms > 0 else None, "bound": "hbm"}}
        except Exception as e:
            res["config2_harris_only"] = {"error": str(e)[:200]}
        try:
            p3 = capi.default_params(rows, cols, n_octaves=args.octaves)
            L3 = capi.batch_layout(p3)
            o3 = {k: shared[k] for k in ("pyramid", "extrema_bits", "dog_points", "dog_counts")}
            ctx.detect_batch(p3, frames, **o3)
            fence()
            ctx.kernel_timing_enable("k_pyr_octave")
            t0 = time.perf_counter()
            for _ in range(steps):
                ctx.detect_batch(p3, frames, **o3)
            fence()
            d = time.perf_counter() - t0
            nl, ms = ctx.kernel_timing_read()
            ctx.kernel_timing_enable(None)
            ab = L3.algorithmic_bytes_dog
            a01 = 11 * sum(L3.rows[o] * L3.cols[o] for o in range(min(2, L3.n_octaves)))
            res["config3_dog_only"] = {
                "what": f"BASELINE config 3 as a batch: DoG pyramid {args.octaves} octaves x (6 Gaussian, 5 DoG) + extrema masks + candidate list, no Harris chain",
                "steps": steps, "frames_per_sec": n * world * steps / d, "ms_per_step": d / steps * 1e3, "algorithmic_bytes_per_frame": ab,
                "pipeline_frac_of_peak": ab * (n * steps / d) / 1e9 / HBM_PEAK_GBPS,
                "k_pyr_octave": {"launches_per_step": nl / steps, "kernel_ms_per_step": ms / steps,
                                 "achieved_GBps": a01 * n * steps / (ms * 1e-3) / 1e9 if ms > 0 else None,
                                 "frac": a01 * n * steps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS if ms > 0 else None, "bound": "hbm (limited by VALU issue)"}}
        except Exception as e:
            res["config3_dog_only"] = {"error": str(e)[:200]}
        return res

    kname = args.kernel or ("k_pyr_octave_mx" if args.matrix_path else "k_pyr_octave")
    leg("setup")
    # `value`: K steps with NO per-launch event hook; the roofline's kernel durations come from a
    # second pass of the same K steps with the hook around the dominant kernel's launches, outside `value`
    main = run_mode(args.localize, args.orient, args.steps, args.warmup, None)
    leg("value")
    hooked = run_mode(args.localize, args.orient, args.steps, 0, kname) if args.roofline_pass else {"launches": 0, "kms": 0.0, "dt": 0.0}
    leg("roofline_pass")
    p, L, dt, launches, kms = main["p"], main["L"], main["dt"], hooked["launches"], hooked["kms"]
    # keypoints of the first frames of this rank's batch, for the CPU baseline's count check
    cs = min(args.cpu_sample, n)
    gpu_kp_sample = None
    if cs > 0 and not (args.localize or args.orient) and args.octaves > 0:
        gpu_kp_sample = int(shared["harris_counts"][:cs].sum().item() + shared["dog_counts"][:cs].sum().item())
    alone = kernel_alone(kname, 3) if (args.modes and kname in ("k_pyr_octave", "k_pyr_octave_mx") and args.octaves >= 2) else None
    leg("alone")
    per_kernel = cfg_legs = None
    if args.modes and secondary and args.octaves >= 1 and not (args.localize or args.orient):
        try:
            per_kernel = by_kernel(p, L, {k: v for k, v in shared.items() if k != "dense_bits"}, bool(args.matrix_path))
        except Exception as e:  # a secondary figure must never take the headline down
            per_kernel = {"error": str(e)[:200]}
        cfg_legs = config_legs()
    leg("by_kernel")
    # OPT-IN matrix-core path: the same steps with vslam_ctx_set_matrix_path(1) - octaves 0..3 as
    # chained i8 MFMA band products instead of packed dots.  The north star rules MFMA out of this path, so this is
    # reported BESIDE the headline, never as `value` / `roofline`.
    mxp = None
    if args.mx and secondary and not args.matrix_path and args.octaves >= 1 and not (args.localize or args.orient):
        try:
            ctx.set_matrix_path(True)
            mm = run_mode(0, 0, args.steps, 2, None)
            mh = run_mode(0, 0, args.steps, 0, "k_pyr_octave_mx")  # the hooked pass, as for `value`
            mm["launches"], mm["kms"] = mh["launches"], mh["kms"]
            ma = kernel_alone("k_pyr_octave_mx", 3) if args.octaves >= 2 else None
            try:
                mk = by_kernel(mm["p"], mm["L"], {k: v for k, v in shared.items() if k != "dense_bits"}, True) if args.modes else None
            except Exception as e:
                mk = {"error": str(e)[:200]}
            ctx.set_matrix_path(False)
            mxp = {"dt": mm["dt"], "launches": mm["launches"], "kms": mm["kms"], "harris": mm["harris"], "dog": mm["dog"], "alone": ma, "by_kernel": mk}
        except Exception as e:
            ctx.set_matrix_path(False)
            mxp = {"error": str(e)[:200]}
    leg("mx_path")
    # the list modes the reference's own functions produce (initialKeypointDetection appends the
    # FeaturePointLocalization survivors, Diff_of_Gauss.cpp:290; filterKeypoints the oriented points,
    # :787), measured in the same process on the same frames with fewer steps
    modes = None
    if args.modes and secondary and args.octaves > 0 and not (args.localize or args.orient):
        # on the checkerboard batch the orientation stage sees ~36 points per frame, so the list
        # modes were timed on near-empty work.  They run on a batch whose every second frame is the uniform-
        # noise frame of SURVEY 8d (tens of thousands of oriented points each); `candidates` is the default
        # mode on that same batch, for reference.  Full list capacities, nothing clamped.
        mixed = synth.frames_torch(n, rows, cols, stream_id=rank, device=dev, noise_every=2)
        modes = {"content": "the bench's camera stream with every second frame replaced by a uniform-noise frame (synth kind=noise)"}
        for name, (lz, orr, de, dn) in (("candidates", (0, 0, 0, 0)), ("localize", (1, 0, 0, 0)), ("orient", (1, 1, 0, 0)), ("describe", (1, 1, 1, 0)),
                                        ("dense", (0, 0, 0, 1))):
            ms = max(2, args.steps // 2)
            m = run_mode(lz, orr, ms, 1, None, de, dn, frames_t=mixed)
            modes[name] = {"frames_per_sec": n * world * ms / m["dt"], "ms_per_step": m["dt"] / ms * 1e3, "steps": ms,
                           "harris_points_per_step": m["harris"], "dog_points_per_step": m["dog"], "list_overflow": m["list_overflow"],
                           **({"oriented_points_per_step": m["oriented"], "oriented_truncated": m["oriented_truncated"],
                               "oriented_cap": int(m["p"].oriented_cap)} if orr else {}),
                           **({"what": "the whole DoG executable: pyramid, initialKeypointDetection, filterKeypoints, SIFT descriptors of every oriented point"} if de else {}),
                           **({"what": "extension: dense 3x3x3 scale-space test on every pixel of levels 1..3 (params.extrema_dense) instead of the reference's lattice test"} if dn else {})}
        # the same list modes with the OPT-IN matrix path on (reported inside `mx_path`, never beside `value`)
        if mxp and "error" not in mxp:
            try:
                ctx.set_matrix_path(True)
                mxp["modes"] = {}
                for name, (lz, orr, de) in (("localize", (1, 0, 0)), ("orient", (1, 1, 0)), ("describe", (1, 1, 1))):
                    ms = max(2, args.steps // 2)
                    m = run_mode(lz, orr, ms, 1, None, de, 0, frames_t=mixed)
                    mxp["modes"][name] = {"frames_per_sec": n * world * ms / m["dt"], "ms_per_step": m["dt"] / ms * 1e3,
                                          "same_counts_as_default_path": bool(m["dog"] == modes[name]["dog_points_per_step"] and
                                                                             (not orr or m["oriented"] == modes[name]["oriented_points_per_step"]))}
            except Exception as e:
                mxp["modes"] = {"error": str(e)[:200]}
            finally:
                ctx.set_matrix_path(False)
        del mixed
    leg("modes")

    # Two batches in flight: a second context on a second stream with output buffers of its own, the
    # K steps alternating between the two.  Reported beside `value`, never as it: the pair's kernels run side by side, so a
    # per-launch roofline of such a run says little, and how much the pair gains depends on the hardware-queue layout.
    two = None
    if args.modes and secondary and args.octaves > 0 and not (args.localize or args.orient) and n * rows * cols <= 256 * 1080 * 1920:
        try:
            s2 = torch.cuda.Stream()
            ctx2 = capi.Context(local_rank_dev, s2.cuda_stream)
            out1 = {k: v for k, v in shared.items() if k != "dense_bits"}
            out2 = {k: torch.empty_like(v) for k, v in out1.items()}
            cur = torch.cuda.current_stream()
            pair = ((ctx, cur, out1), (ctx2, s2, out2))

            def step2(k):
                c, st, o = pair[k & 1]
                with torch.cuda.stream(st):
                    c.detect_batch(p, frames, **o)

            s2.wait_stream(cur)
            for k in range(12):  # six calls per context: the second context's stream tuner has decided before the timing starts
                step2(k)
            fence()
            t0 = time.perf_counter()
            for k in range(args.steps):
                step2(k)
            fence()
            dt2 = time.perf_counter() - t0
            same = bool((out2["dog_counts"] == out1["dog_counts"]).all() and (out2["harris_counts"] == out1["harris_counts"]).all())
            two = {"frames_per_sec": n * world * args.steps / dt2, "ms_per_step": dt2 / args.steps * 1e3, "steps": args.steps,
                   "same_counts_on_both_contexts": same,
                   "what": "K steps alternating between two contexts / streams / output sets (per rank); not `value`: see DESIGN section 5.4"}
            ctx2.close()
            del out2
        except Exception as e:  # a secondary figure must never take the headline down
            two = {"error": str(e)[:200]}
    leg("two_in_flight")

    if rank == 0:
        algo = kernel_algorithmic_bytes(L, rows, cols)
        total_frames = n * world * args.steps
        fps = total_frames / dt
        kp_per_step = main["harris"] + main["dog"]
        bytes_frame = L.algorithmic_bytes_harris + L.algorithmic_bytes_dog - rows * cols  # fused: input counted once
        roof = None
        step_traffic = None
        if launches and kms > 0:
            ach = algo.get(kname, 0) * n * args.steps / (kms * 1e-3) / 1e9
            # HBM bytes per launch measured by rocprofv3 PMC passes of this same command (separate
            # --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per MI355X_MICROARCH.md) and
            # committed as profiles/traffic.json; counters cannot be read from inside this process,
            # so the figure is the committed profile's, scaled to this run's frames per launch
            traffic = tsrc = tprof = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            # (not inside a profiler run of this script itself: tools/refresh_profiles.sh wraps it in rocprofv3 with --modes 0)
            profiled = any("rocprof" in os.environ.get(k, "") for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD"))
            lt = live_traffic(kname, rows, cols, args.octaves, matrix_path=args.matrix_path) if (args.live_traffic and args.modes and world == 1 and not profiled) else None
            leg("live_traffic")
            if lt:
                traffic = lt["hbm_bytes_per_frame"] * n * args.steps / launches
                tsrc = "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes (separate), 2*FETCH_SIZE + WRITE_SIZE KiB"
                tprof = dict(lt, live=True, scaled_to_frames_per_launch=n * args.steps / launches)
                step_traffic = lt.get("step_hbm_bytes_per_frame")
            elif os.path.exists(tpath):
                tj = json.load(open(tpath))
                t = tj.get(kname)
                if t:
                    traffic = t["hbm_bytes_per_frame"] * n * args.steps / launches
                    step_traffic = tj.get("_total_bytes_per_frame")
                    tsrc = "profiles/traffic.json: " + tj.get("_round", "rocprofv3 --pmc passes")
                    # NOT measured in this run: the committed profile's per-frame figure times this run's frames per launch
                    tprof = {"live": False, "hbm_bytes_per_frame": t["hbm_bytes_per_frame"], "profiled_frames_per_batch": tj.get("_frames"),
                             "profile": tj.get("_round"), "scaled_to_frames_per_launch": n * args.steps / launches}
            roof = {
                "kernel": kname, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": tsrc, "traffic_profiled": tprof,
                "timed_in": "a second pass of the same K steps with the HIP-event hook on (the `value` pass runs with the hook off)",
                "hooked_pass_ms_per_step": hooked["dt"] / args.steps * 1e3,
                "limited_by": "valu-dot issue rate, see roofline_valu" if kname == "k_pyr_octave" else ("hbm (write-dominated mix: DESIGN 5.6)" if kname == "k_pyr_octave_mx" else None),
                "launches": launches, "avg_launch_ms": kms / launches,
                "algorithmic_bytes_per_frame": algo.get(kname, 0),
                "algorithmic_bytes_per_launch": algo.get(kname, 0) * n * args.steps / launches,
            }
            if alone and alone[0] and alone[1] > 0:
                # `achieved` / `frac` above are from the timed region, where octave 0's extrema scan runs
                # beside octave 1's launch; this is the same kernel with the chip to itself
                a_ms = alone[1] / alone[0]
                a_ach = roof["algorithmic_bytes_per_launch"] / (a_ms * 1e-3) / 1e9
                roof["alone"] = {"avg_launch_ms": a_ms, "achieved": a_ach, "frac": a_ach / HBM_PEAK_GBPS,
                                 "what": "pyramid-only batches (no Harris chain, no extrema scan), 3 steps, outside the timed region"}
        # second roof of the same kernel: it is bound by VALU issue of the packed dot instructions,
        # not by HBM.  Algorithmic dot instructions per pixel = sum over levels of
        # n/4 (v_dot4_u32_u8, vertical) + n/2 (v_dot2_u32_u16, horizontal) with the zero-trimmed
        # kernel widths n; peak = measured issue rate of a wave64 dot instruction, 2.0 ns per SIMD on
        # 1024 SIMDs = 32.8e12 lane-instr/s (tools/ubench_valu.hip, tools/ubench_valu2.hip).
        valu = None
        if roof and kname == "k_pyr_octave" and args.octaves >= 2:
            widths = [[gauss_trimmed_width(capi, capi.sigma_at(p.sigma0, o, l)) for l in range(6)] for o in range(2)]
            per_frame = sum(0.75 * sum(w) * L.rows[o] * L.cols[o] for o, w in enumerate(widths))
            ach = per_frame * n * args.steps / (kms * 1e-3)
            valu = {"bound": "valu-dot", "achieved": ach / 1e12, "peak": 32.8, "unit": "T lane-instr/s",
                    "frac": ach / 32.8e12, "algorithmic_dot_instr_per_frame": per_frame, "trimmed_widths": widths}
        mx_obj = None
        if mxp and "error" in mxp:
            mx_obj = mxp
        elif mxp:
            a_bytes = algo["k_pyr_octave_mx"]
            mx_obj = {
                "what": "OPT-IN (VSLAM_MX=1 / vslam_ctx_set_matrix_path): octaves 0-3 as chained v_mfma_i32_32x32x32_i8 band products, bit-identical outputs; "
                        "NOT `value`: BASELINE's north star rules MFMA out of this path (DESIGN section 5.5)",
                "frames_per_sec": n * world * args.steps / mxp["dt"], "ms_per_step": mxp["dt"] / args.steps * 1e3, "steps": args.steps,
                "speedup_vs_value": (n * world * args.steps / mxp["dt"]) / fps,
                "same_keypoint_counts_as_value": bool(mxp["harris"] == main["harris"] and mxp["dog"] == main["dog"]),
                "k_pyr_octave_mx": {
                    "launches_per_step": mxp["launches"] / args.steps, "kernel_ms_per_step": mxp["kms"] / args.steps,
                    "avg_launch_ms": mxp["kms"] / max(1, mxp["launches"]),
                    "algorithmic_bytes_per_frame": a_bytes,
                    "achieved": a_bytes * n * args.steps / (mxp["kms"] * 1e-3) / 1e9 if mxp["kms"] > 0 else None, "unit": "GB/s",
                    "frac": a_bytes * n * args.steps / (mxp["kms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS if mxp["kms"] > 0 else None,
                },
                "pipeline_hbm_frac_of_peak": bytes_frame * (n * args.steps / mxp["dt"]) / 1e9 / HBM_PEAK_GBPS,
            }
            if mxp.get("by_kernel"):
                mx_obj["roofline_by_kernel"] = mxp["by_kernel"]
            # live roofline of the path's dominant kernel, like the default path's: achieved from the HIP events of the step,
            # traffic from two rocprofv3 --pmc child passes of this script with the switch on
            if mxp["kms"] > 0 and mxp["launches"]:
                k = mx_obj["k_pyr_octave_mx"]
                profiled = any("rocprof" in os.environ.get(kk, "") for kk in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD"))
                lt = live_traffic("k_pyr_octave_mx", rows, cols, args.octaves, matrix_path=1) if (args.live_traffic and args.modes and world == 1 and not profiled) else None
                mx_obj["roofline"] = {
                    "kernel": "k_pyr_octave_mx", "bound": "hbm", "achieved": k["achieved"], "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": k["frac"],
                    "launches": mxp["launches"], "avg_launch_ms": k["avg_launch_ms"],
                    "algorithmic_bytes_per_launch": a_bytes * n * args.steps / mxp["launches"],
                    "traffic": (lt["hbm_bytes_per_frame"] * n * args.steps / mxp["launches"]) if lt else None,
                    "traffic_bytes_per_frame": lt["hbm_bytes_per_frame"] if lt else None,
                    "traffic_source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes with --matrix-path 1 (separate), 2*FETCH_SIZE + WRITE_SIZE KiB" if lt else None,
                    "traffic_profiled": dict(lt, live=True) if lt else None}
                leg("mx_live_traffic")
            if mxp.get("modes"):
                mx_obj["modes"] = dict(mxp["modes"], content="as `modes`: every second frame uniform noise")
            if mxp["alone"] and mxp["alone"][0] and mxp["alone"][1] > 0:
                a_ms_step = mxp["alone"][1] / 3
                mx_obj["k_pyr_octave_mx"]["alone"] = {"kernel_ms_per_step": a_ms_step, "achieved": a_bytes * n / (a_ms_step * 1e-3) / 1e9,
                                                      "frac": a_bytes * n / (a_ms_step * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                                      "what": "pyramid-only batches, 3 steps"}
        line = {
            "metric": "frames/sec @1080p (Harris + DoG keypoint detection)",
            "value": fps,
            "unit": "frames/s",
            "n_gpus": ranks_gathered,
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
                "matrix_path": bool(args.matrix_path), "side_level_pinned": args.side_level if args.side_level >= 0 else None,
                "parallelism": f"frames sharded 1 stream/GPU x{world}; RCCL all-gather of counts only",
            },
            "distributed": {"initialized": bool(use_dist), "world_size": dist.get_world_size() if use_dist else 1,
                            "backend": dist.get_backend() if use_dist else None, "ranks_gathered": int(ranks_gathered),
                            "distinct_gpus": len({(m.get("uuid"), m.get("pci_bus_id"), m.get("device_index")) for m in members if m}),
                            "members": members if world > 1 else None},
            "join_watch": dict(zip(("level", "done", "last_lag_fraction"), ctx.join_watch_report())),  # 0 = low-priority side streams kept
            "side_streams": dict(zip(("pair", "tuner_state"), ctx.side_stream_report())),  # which candidate pair the library's stream tuner kept (0 = the first), 2 = decided: DESIGN section 5.4
            "keypoints_per_sec": kp_per_step * args.steps / dt,
            "keypoints_per_step": {"harris": main["harris"], "dog": main["dog"], "list_overflow": main["list_overflow"],
                                   **({"oriented": main["oriented"], "oriented_truncated": main["oriented_truncated"]} if args.orient else {})},
            "modes": modes,
            "two_in_flight": two,
            "mx_path": mx_obj,
            "pipeline_hbm": {
                "algorithmic_bytes_per_frame": bytes_frame,
                "achieved_GBps": bytes_frame * fps / world / 1e9,
                "frac_of_peak": bytes_frame * fps / world / 1e9 / HBM_PEAK_GBPS,
                # HBM bytes ALL kernels of the step moved per frame (the same two PMC child passes as roofline.traffic) over the algorithmic bytes
                "traffic_bytes_per_frame": step_traffic,
                "traffic_ratio": (step_traffic / bytes_frame) if step_traffic else None,
            },
            "roofline": roof,
            "roofline_valu": valu,
            "roofline_by_kernel": per_kernel,
            **(cfg_legs or {}),
            "cpu_baseline": None,  # filled in below, after the C++ host's runs: 18 s of 64 busy OpenMP threads right in front of the
                                   # host-fed pipeline cost it 10 % (11.3 k against 13.0 k stand-alone on the same box)
        }
    # The same workload driven by the C++ host (visualslam_amd/cxx: BatchDetector + Stream, RCCL from librccl,
    # no torch in that process): device-resident, and host-fed (pinned frames in, packed lists out).  Child
    # processes, after this process has released the GPU memory; N = 1 only; never part of `value`.
    # (under the rehearsal switches of tests/test_bench_ranks.py the ranks share one GPU, which RCCL refuses: the C++
    # processes then exchange their counts over the TCP rehearsal backend)
    cxx_wanted = bool(args.cxx_host) and (world == 1 or backend == "nccl" or os.environ.get("VSLAM_BENCH_SHARE_GPU") == "1")
    ctx.close()
    if cxx_wanted and cxx_early is not None:
        line["cxx_host"] = dict(cxx_early, order="before this process touched the GPU")
    elif cxx_wanted:
        shared.clear()
        del frames
        torch.cuda.empty_cache()
        if use_dist:
            dist.barrier()  # every rank has released its buffers and starts its C++ process now
        cxx = cxx_host_runs(rows, cols, n, args.octaves, rank, world, local_rank_dev)
        if use_dist:
            dist.barrier()
        if rank == 0:
            line["cxx_host"] = cxx
    leg("cxx_host")
    wall["cxx_host"] = wall.get("cxx_host", 0.0) + cxx_early_s
    if rank == 0 and world == 1 and args.cpu_sample > 0:
        line["cpu_baseline"] = cpu_baseline(rows, cols, args.octaves, cs, gpu_kp_sample)
    leg("cpu_baseline")
    if rank == 0:
        line["bench_wall_s"] = {k: round(v, 2) for k, v in wall.items()}
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
