"""bench.py's N > 1 path end to end on the one-GPU box: two ranks under torch.distributed.run share
cuda:0 and exchange their keypoint counts over gloo (RCCL refuses two ranks on one GPU; the
collective calls, the barrier / max-over-ranks timing and the rank-0 JSON line are the same)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, VSLAM_BENCH_BACKEND="gloo", VSLAM_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "4", "--rows", "240", "--cols", "320",
           "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--live-traffic", "0"]
    import time

    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    wall = time.time() - t0
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 prints the one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    # a scaling run must not spend minutes in secondary figures.  At N > 1 the list modes, two-in-flight and
    # the matrix-path leg are skipped, the C++ children are bounded (20 s rendezvous, 45 s per child) and the line says
    # where the time went
    assert wall < 120, wall
    assert d["modes"] is None and d["two_in_flight"] is None and d["mx_path"] is None
    bw = d["bench_wall_s"]
    assert {"value", "modes", "two_in_flight", "mx_path", "live_traffic", "cxx_host", "cpu_baseline"} <= set(bw), bw
    assert bw["cxx_host"] < 60 and bw["modes"] < 1 and bw["two_in_flight"] < 1
    # the C++ host after the measurement: one Stream process per rank, their own communicator (here the TCP rehearsal
    # exchange), rank 0's child reports the job: two ranks, two different camera streams, the same totals as the Python ranks
    cx = d["cxx_host"]["device"]
    assert cx["n_gpus"] == 2 and cx["frames_per_sec"] > 0 and len(cx["counts_by_rank"]) == 2, cx
    assert cx["keypoints_per_batch"] == {"harris": d["keypoints_per_step"]["harris"], "dog": d["keypoints_per_step"]["dog"]}
    # both ranks' streams are counted: the per-step totals are the sum over two different streams
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "4", "--rows", "240", "--cols", "320", "--steps", "2",
                          "--warmup", "1", "--cpu-sample", "0", "--live-traffic", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d["keypoints_per_step"]["harris"] > d1["keypoints_per_step"]["harris"]
    assert d["keypoints_per_step"]["dog"] > d1["keypoints_per_step"]["dog"]


@pytest.mark.gpu
def test_plain_bench_command_forms_its_ranks_itself():
    # `python bench.py --gpus 2` WITHOUT torchrun around it (the form the driver uses for --gpus 1) must
    # measure two ranks: bench.py starts the torch.distributed.run job as a child, relays rank 0's line, and fails unless
    # the line shows two ranks gathered.  Same rehearsal switches as above (two ranks on the one GPU, gloo).
    env = dict(os.environ, VSLAM_BENCH_BACKEND="gloo", VSLAM_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "4", "--rows", "240", "--cols", "320", "--steps", "2",
                        "--warmup", "1", "--cpu-sample", "0", "--live-traffic", "0", "--cxx-host", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    dd = d["distributed"]
    assert dd["initialized"] is True and dd["world_size"] == 2 and dd["ranks_gathered"] == 2 and dd["backend"] == "gloo"
    assert sorted(m["rank"] for m in dd["members"]) == [0, 1]
    assert dd["distinct_gpus"] == 1  # the rehearsal's two ranks share cuda:0, and the line says so
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "4", "--rows", "240", "--cols", "320", "--steps", "2", "--warmup", "1",
                          "--cpu-sample", "0", "--live-traffic", "0", "--modes", "0", "--mx", "0", "--cxx-host", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_gpus"] == 1 and d1["distributed"]["initialized"] is False
    # two different camera streams in the totals
    assert d["keypoints_per_step"]["harris"] > d1["keypoints_per_step"]["harris"] and d["keypoints_per_step"]["dog"] > d1["keypoints_per_step"]["dog"]
    assert d["keypoints_per_step"]["dog"] != 2 * d1["keypoints_per_step"]["dog"] or d["keypoints_per_step"]["harris"] != 2 * d1["keypoints_per_step"]["harris"]


@pytest.mark.gpu
def test_plain_bench_command_with_four_ranks():
    # the same launcher at N = 4 (four ranks on the one GPU of the test box: within the box's six-process limit): every rank
    # is counted, every rank's camera stream is in the totals, the C++ children of the four ranks meet too
    env = dict(os.environ, VSLAM_BENCH_BACKEND="gloo", VSLAM_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--frames", "2", "--rows", "120", "--cols", "160", "--steps", "2",
                        "--warmup", "1", "--cpu-sample", "0", "--live-traffic", "0", "--cxx-host", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 4 and d["distributed"]["ranks_gathered"] == 4 and sorted(m["rank"] for m in d["distributed"]["members"]) == [0, 1, 2, 3]
    assert d["config"]["frames_per_gpu"] == 2 and d["value"] > 0
    assert "x4" in d["config"]["parallelism"]


def test_plain_bench_command_with_ranks_fails_loudly_without_gpus():
    # the launcher itself on the CPU box: the child job starts (two ranks under torch.distributed.run), every rank refuses to
    # run without a GPU, the launcher hands the failure on instead of printing a one-GPU line
    import torch

    if torch.cuda.is_available():
        pytest.skip("CPU-box behaviour")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "2", "--rows", "64", "--cols", "64", "--steps", "1", "--warmup", "0",
                        "--cpu-sample", "0", "--cxx-host", "0"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "needs a GPU" in r.stderr, r.stderr[-1500:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bench_one_rank_goes_through_rccl():
    # the driver launches N ranks with torch.distributed.run over RCCL; with one rank on the one GPU of
    # the test box the same code path runs: init_process_group("nccl"), the barrier / max-over-ranks
    # timing, the all-gather of the counts on device tensors and the flag reductions
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("VSLAM_BENCH_BACKEND", None)
    env.pop("VSLAM_BENCH_SHARE_GPU", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--frames", "8", "--rows", "240", "--cols", "320",
           "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--live-traffic", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["distributed"] == {"initialized": True, "world_size": 1, "backend": "nccl", "ranks_gathered": 1, "distinct_gpus": 1, "members": None}
    assert "hook off" in d["roofline"]["timed_in"] and d["roofline"]["launches"] > 0  # the kernel durations come from a pass of their own
    assert d["two_in_flight"]["frames_per_sec"] > 0 and d["two_in_flight"]["same_counts_on_both_contexts"] is True  # secondary figure, DESIGN 5.4
    # the opt-in matrix path beside the headline: same counts, its own kernel figures, never `value`
    assert d["config"]["matrix_path"] is False and d["mx_path"]["same_keypoint_counts_as_value"] is True and d["mx_path"]["frames_per_sec"] > 0
    assert d["mx_path"]["k_pyr_octave_mx"]["launches_per_step"] >= 2
    for name in ("localize", "orient", "describe"):  # the list modes with the matrix path on: the default path's counts
        assert d["mx_path"]["modes"][name]["same_counts_as_default_path"] is True and d["mx_path"]["modes"][name]["frames_per_sec"] > 0
    assert d["cxx_host"]["device"]["frames_per_sec"] > 0 and "RCCL" in d["cxx_host"]["device"]["host"]
    assert d["cxx_host"]["device"]["keypoints_per_batch"] == {"harris": d["keypoints_per_step"]["harris"], "dog": d["keypoints_per_step"]["dog"]}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "8", "--rows", "240", "--cols", "320", "--steps", "2",
                          "--warmup", "1", "--cpu-sample", "0", "--modes", "0", "--live-traffic", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d["keypoints_per_step"] == d1["keypoints_per_step"]  # the gathered counts are the rank's own


@pytest.mark.gpu
def test_bench_measures_its_hbm_traffic_live():
    # roofline.traffic was a committed constant.  The default run now measures it: two rocprofv3 --pmc child
    # passes (FETCH_SIZE, WRITE_SIZE) of a small batch; the line says so, and the figure is at least the algorithmic bytes
    import shutil

    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        pytest.skip("no rocprofv3 on this box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "16", "--rows", "540", "--cols", "960", "--steps", "2", "--warmup", "1",
                        "--cpu-sample", "0", "--cxx-host", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    roof = d["roofline"]
    assert roof["traffic_profiled"]["live"] is True and "measured in this run" in roof["traffic_source"], roof
    assert roof["traffic"] >= 0.95 * roof["algorithmic_bytes_per_launch"]
    assert roof["traffic"] < 3 * roof["algorithmic_bytes_per_launch"]
