"""The C++ host mirror (visualslam_amd/cxx) builds against the C ABI; its headless executables
(same target names as the reference: Harris, DoG, Pyramid_Test) run on the GPU box and fail
loudly without a GPU."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "visualslam_amd", "bin")


@pytest.fixture(scope="module")
def built():
    from visualslam_amd import capi

    capi.build()
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "visualslam_amd", "cxx")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return BIN


def test_cxx_mirror_builds(built):
    for exe in ("Harris", "DoG", "Pyramid_Test", "RotateImgTest", "Stream", "BatchDetector_Test"):
        assert os.access(os.path.join(built, exe), os.X_OK)


def test_rotate_img_test_runs_without_a_gpu(built):
    # the Rotation members on the descriptor path are host arithmetic (rotation.cpp:5-27,112-130)
    r = subprocess.run([os.path.join(built, "RotateImgTest")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["exe"] == "RotateImgTest" and out["points"] == 289 and out["failures"] == 0


def _cmake_project(built, tmp_path):
    # CMake >= 3.21 + CTest with the reference's target names (KeyPointDetection/CMakeLists.txt:7-13,
    # include/CMakeLists.txt:1-5, tests/CMakeLists.txt:1-5); the HIP library is taken prebuilt here
    from visualslam_amd import capi

    b = str(tmp_path / "build")
    src = os.path.join(ROOT, "visualslam_amd", "cxx")
    r = subprocess.run(["cmake", "-S", src, "-B", b, "-DVSLAM_PREBUILT_LIB=" + capi.LIB_PATH], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["cmake", "--build", b, "-j", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    for exe in ("Harris", "DoG", "Pyramid_Test", "RotateImgTest", "Stream", "BatchDetector_Test"):
        assert os.access(os.path.join(b, exe), os.X_OK)
    r = subprocess.run(["ctest", "-N"], capture_output=True, text=True, timeout=120, cwd=b)
    for name in ("Harris", "DoG", "Pyramid_Test", "RotateImgTest", "Stream", "Stream_hostfed", "BatchDetector_Test"):
        assert ": " + name in r.stdout, r.stdout
    r = subprocess.run(["ctest", "-R", "RotateImgTest", "--output-on-failure"], capture_output=True, text=True, timeout=120, cwd=b)
    assert r.returncode == 0, r.stdout + r.stderr
    return b


def test_cmake_build_registers_the_reference_targets(built, tmp_path):
    _cmake_project(built, tmp_path)


@pytest.mark.gpu
def test_ctest_project_passes_on_the_gpu(built, tmp_path):
    b = _cmake_project(built, tmp_path)
    r = subprocess.run(["ctest", "--output-on-failure"], capture_output=True, text=True, timeout=600, cwd=b)
    assert r.returncode == 0 and "100% tests passed" in r.stdout, r.stdout + r.stderr


def test_executables_fail_loudly_without_gpu(built):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([os.path.join(built, "Harris"), "64x48"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no CPU fallback" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("exe,arg", [("Harris", "640x480"), ("Harris", "1754x1240"), ("DoG", "512x384"), ("Pyramid_Test", None)])
def test_executables_run(built, exe, arg, tmp_path):
    cmd = [os.path.join(built, exe)] + ([arg] if arg else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=tmp_path)  # DoG writes featureDescriptors.dat into its cwd
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["exe"] == exe
    if exe == "Harris":
        assert out["keypoints_stagewise"] == out["keypoints_fused"] > 0 and out["structure_matrix_mismatch"] == 0
    if exe == "DoG":
        assert len(out["octaves"]) == 4 and out["keypoints"] > 0 and out["per_point_mismatch"] == 0
        assert out["descriptors"] == out["oriented"] and os.path.exists(tmp_path / "featureDescriptors.dat")
    if exe == "Pyramid_Test":
        assert out["failures"] == 0


@pytest.mark.gpu
def test_cxx_results_match_oracle(built, tmp_path):
    # a PGM on disk through the C++ executables gives the oracle's counts
    import numpy as np

    import oracle
    from visualslam_amd import synth

    img = synth.frame_np(96, 160, 0, 0, "checker")  # same generator as imgio::synthetic(96,160)
    p = tmp_path / "f.pgm"
    with open(p, "wb") as f:
        f.write(b"P5\n160 96\n255\n" + img.tobytes())
    r = subprocess.run([os.path.join(built, "Harris"), str(p)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want = len(oracle.harris_keypoints(oracle.nms2(oracle.harris_response(img), 5)[0]))
    assert json.loads(r.stdout.strip().splitlines()[-1])["keypoints_fused"] == want
    r = subprocess.run([os.path.join(built, "Harris"), "160x96"], capture_output=True, text=True, timeout=300)
    assert json.loads(r.stdout.strip().splitlines()[-1])["keypoints_fused"] == want  # imgio::synthetic == synth.py
    r = subprocess.run([os.path.join(built, "DoG"), str(p), str(tmp_path / "fd.dat")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    pyr = oracle.Pyramid(img, 4, 1.6)
    want_oct = [len(pyr.extrema(o, 3, 8)[1]) for o in range(4)]
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert [o["candidates"] for o in got["octaves"]] == want_oct
    assert [o["keypoints"] for o in got["octaves"]] == [len(pyr.keypoints(o, 3)) for o in range(4)]
    assert [o["oriented"] for o in got["octaves"]] == [len(pyr.filter_keypoints(o, pyr.keypoints(o, 3))) for o in range(4)]
    assert [o["dense_3x3x3"] for o in got["octaves"]] == [len(pyr.extrema_dense(o, 8)[1]) for o in range(4)]
    assert got["per_point_mismatch"] == 0


@pytest.mark.gpu
def test_dog_executable_writes_the_reference_descriptor_file(built, tmp_path):
    # the one persisted artefact of the reference's DoG executable (Diff_of_Gauss.cpp:837-863): int32
    # {n, 128, sizeof(std::vector<float>) = 24} then n x 128 float32 -- on the reference's square image
    # (blox.jpg: every rotated window is defined), equal to the oracle's SIFT() output octave by octave
    import numpy as np

    import oracle
    from tests import refimg

    img = refimg.load("blox")
    p = tmp_path / "blox.pgm"
    with open(p, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]) + img.tobytes())
    dat = tmp_path / "featureDescriptors.dat"
    r = subprocess.run([os.path.join(built, "DoG"), str(p), str(dat)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    pyr = oracle.Pyramid(img, 4, 1.6)
    want = []
    for o in range(4):
        d, ok = pyr.sift_descriptors(o, pyr.filter_keypoints(o, pyr.keypoints(o, 3)))
        assert ok.all()
        want.append(d)
    want = np.concatenate(want)
    raw = open(dat, "rb").read()
    head = np.frombuffer(raw[:12], "<i4")
    assert head.tolist() == [len(want), 128, 24] and len(raw) == 12 + 4 * 128 * len(want)
    body = np.frombuffer(raw[12:], "<f4").reshape(-1, 128)
    assert np.array_equal(body, want, equal_nan=True)
    assert got["descriptors"] == len(want) > 0 and got["undefined_windows"] == 0


# ---- the C++ host of the throughput path: BatchDetector, Stream, RCCL count all-gather

def _rank_env(rank, world, port):
    return dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))


@pytest.mark.parametrize("world", [2, 4, 8])  # 8 = the ranks of BASELINE config 5 (one node)
def test_stream_rendezvous_of_the_rccl_id_without_a_gpu(built, world):
    # the N > 1 bootstrap of the RCCL communicator (rank 0's ncclUniqueId over TCP to every rank) on its own:
    # `world` processes on the CPU, every rank must end up with rank 0's 128 bytes
    port = 29840 + world
    procs = [subprocess.Popen([os.path.join(built, "Stream"), "--rdv-selftest"], env=_rank_env(r, world, port), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in reversed(range(world))]  # rank 0 last: the others must retry
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    got = [json.loads(o[0].strip().splitlines()[-1]) for o in outs]
    assert sorted(g["rank"] for g in got) == list(range(world)) and all(g["world"] == world for g in got)
    assert len({g["id_hash"] for g in got}) == 1


def test_stream_rendezvous_ignores_foreign_connections(built):
    # rank 0 used to abort the whole job on the first stray connection and to hand the id to anything that sent a
    # plausible rank.  A port scanner (connects, sends nothing), a peer with a well-formed hello of ANOTHER job (wrong token)
    # and a garbage sender all get dropped; the real rank 1, arriving last, is served
    import socket
    import struct
    import time

    port = 29877
    env0, env1 = _rank_env(0, 2, port), _rank_env(1, 2, port)
    p0 = subprocess.Popen([os.path.join(built, "Stream"), "--rdv-selftest"], env=env0, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    strays = []
    try:
        deadline = time.time() + 30
        while True:  # wait for rank 0 to listen
            try:
                strays.append(socket.create_connection(("127.0.0.1", port + 1), timeout=1))
                break
            except OSError:
                assert time.time() < deadline and p0.poll() is None
                time.sleep(0.05)
        for payload in (struct.pack("<IIi", 0x56534C4D, 0xDEADBEEF, 1), b"GET / HTTP/1.0\r\n\r\n"):
            c = socket.create_connection(("127.0.0.1", port + 1), timeout=1)
            c.sendall(payload)
            strays.append(c)
        p1 = subprocess.Popen([os.path.join(built, "Stream"), "--rdv-selftest"], env=env1, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        o1 = p1.communicate(timeout=120)
        o0 = p0.communicate(timeout=120)
    finally:
        for c in strays:
            c.close()
        p0.kill()
    assert p0.returncode == 0 and p1.returncode == 0, (o0, o1)
    a, b = json.loads(o0[0].strip().splitlines()[-1]), json.loads(o1[0].strip().splitlines()[-1])
    assert a["id_hash"] == b["id_hash"] and {a["rank"], b["rank"]} == {0, 1}


def test_stream_rendezvous_times_out_loudly(built):
    # a rank that never finds rank 0 fails with a message instead of hanging (60 s by default, shortened here)
    env = dict(_rank_env(1, 2, 29870), VSLAM_RDV_PORT="1", VSLAM_RDV_TIMEOUT_MS="1500")  # nothing listens on port 1
    p = subprocess.Popen([os.path.join(built, "Stream"), "--rdv-selftest"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out = p.communicate(timeout=100)
    finally:
        p.kill()
    assert p.returncode == 1 and "could not reach rank 0" in out[1]


@pytest.mark.gpu
def test_batch_detector_test_passes(built):
    r = subprocess.run([os.path.join(built, "BatchDetector_Test")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["failures"] == 0 and out["harris"] > 0 and out["dog"] > 0


def _read_dump(path):
    import numpy as np

    from visualslam_amd import capi

    raw = open(path, "rb").read()
    magic, n, rows, cols = np.frombuffer(raw[:16], "<u4")
    assert magic == 0x504B5356
    off, frames = 16, []
    for _ in range(n):
        nh, nd, th, td = np.frombuffer(raw[off: off + 16], "<u4")
        off += 16
        kp = np.frombuffer(raw[off: off + 12 * nh], capi.KP_DTYPE)
        off += 12 * nh
        pt = np.frombuffer(raw[off: off + 24 * nd], capi.POINT_DTYPE)
        off += 24 * nd
        frames.append((kp, pt, int(th), int(td)))
    assert off == len(raw)
    return int(rows), int(cols), frames


@pytest.mark.gpu
@pytest.mark.parametrize("mode,pipelines", [("device", 1), ("hostfed", 1), ("device", 2), ("hostfed", 2)])
def test_stream_one_rank_over_rccl_matches_the_oracle(built, tmp_path, mode, pipelines):
    # (pipelines = 2: consecutive batches alternate between two contexts / streams, the collectives on a stream of their own)
    # the C++ host end to end: one process = one rank, RCCL communicator of one rank (ncclCommInitRank +
    # ncclAllGather from librccl, no torch), BatchDetector driving vslam_detect_batch_dev; the per-frame lists
    # of the last batch against the oracle on four frames of the camera stream (stream id = rank = 0)
    import numpy as np

    import oracle
    from visualslam_amd import synth

    rows, cols, n = 270, 480, 6
    dump = tmp_path / "lists.bin"
    r = subprocess.run([os.path.join(built, "Stream"), "--mode", mode, "--frames", str(n), "--batches", "4", "--warmup", "1", "--rows", str(rows),
                        "--cols", str(cols), "--dump", str(dump), "--pipelines", str(pipelines)], capture_output=True, text=True, timeout=600,
                       env=_rank_env(0, 1, 29890))
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["exe"] == "Stream" and line["mode"] == mode and line["n_gpus"] == 1 and line["frames_per_sec"] > 0 and line["pipelines"] == pipelines
    gr, gc, frames = _read_dump(dump)
    assert (gr, gc, len(frames)) == (rows, cols, n)
    tot_h = tot_d = 0
    for f, (kp, pt, th, td) in enumerate(frames):
        tot_h, tot_d = tot_h + th, tot_d + td
        if f in (0, 1, 3, n - 1):
            img = synth.frame_np(rows, cols, f, 0)
            want_k = oracle.harris_keypoints(oracle.nms2(oracle.harris_response(img), 5)[0])
            assert th == len(want_k) and kp.tobytes() == want_k.tobytes(), f
            pyr = oracle.Pyramid(img, 4, 1.6)
            want_p = np.concatenate([pyr.extrema(o, 3, 8)[1] for o in range(4)])
            pyr.close()
            assert td == len(want_p) and pt.tobytes() == want_p.tobytes(), f
    # the all-gathered totals (one rank: its own) are the sums of the per-frame counts
    assert line["keypoints_per_batch"] == {"harris": tot_h, "dog": tot_d} and line["rank0_counts"] == [tot_h, tot_d]


@pytest.mark.gpu
def test_stream_results_do_not_depend_on_the_hardware_queue_layout(built):
    # GPU_MAX_HW_QUEUES changes which hardware queue every stream lands on.  Two mechanisms may then move the side work
    # (DESIGN section 5.4): the join watchdog (always on: steps the side streams down when the join at the end of a call lags) and
    # the opt-in stream tuner (Stream --tuner: the 2nd to 5th batch on different pairs of side streams, the fastest kept).
    # Whatever either decides, the counts must not move.
    totals = set()
    for q, extra in (("1", []), ("3", []), ("4", []), ("12", []), ("3", ["--tuner"]), ("4", ["--tuner"])):
        env = dict(_rank_env(0, 1, 29893), GPU_MAX_HW_QUEUES=q)
        r = subprocess.run([os.path.join(built, "Stream"), "--mode", "device", "--frames", "32", "--batches", "4", "--warmup", "6", "--rows", "270", "--cols", "480"] + extra,
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["gpu_max_hw_queues"] == q and line["join_watch"]["level"] in (0, 1, 2)
        if extra:
            assert 0 <= line["side_stream_pair"] <= 2 and line["side_stream_tuner"] == 2
        else:
            assert line["side_stream_tuner"] in (0, 2)  # 2 only when the watchdog stepped down (it ends the tuner's candidates)
        totals.add((line["keypoints_per_batch"]["harris"], line["keypoints_per_batch"]["dog"]))
    assert len(totals) == 1 and min(next(iter(totals))) > 0, totals


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["device", "hostfed"])
def test_stream_two_ranks_on_one_gpu_with_the_rehearsal_exchange(built, tmp_path, mode):
    # N > 1 in the C++ host: RCCL refuses two ranks per device, so the rehearsal backend (VSLAM_COUNT_BACKEND=tcp: the
    # same collectives through the rendezvous sockets) carries the counts of two processes that share the GPU - like
    # tests/test_bench_ranks.py does for the Python host over gloo.  Rank r reads camera stream r: the all-gathered
    # table must hold each rank's own totals, rank 0's line the whole-job sums, and rank 1's frames the oracle's lists.
    import numpy as np

    import oracle
    from visualslam_amd import synth

    rows, cols, n, world = 135, 240, 5, 2
    procs = []
    for r in reversed(range(world)):
        env = dict(_rank_env(r, world, 29930 + (mode == "hostfed")), VSLAM_COUNT_BACKEND="tcp")
        procs.append((r, subprocess.Popen([os.path.join(built, "Stream"), "--mode", mode, "--frames", str(n), "--batches", "3", "--warmup", "1", "--rows", str(rows),
                                           "--cols", str(cols), "--dump", str(tmp_path / f"lists{r}.bin")], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.PIPE, text=True)))
    outs = {r: p.communicate(timeout=600) for r, p in procs}
    assert all(p.returncode == 0 for _, p in procs), outs
    line = json.loads(outs[0][0].strip().splitlines()[-1])
    assert line["n_gpus"] == world and "rehearsal" in line["host"] and line["mode"] == mode
    assert outs[1][0].strip() == "" or "frames_per_sec" not in outs[1][0]  # only rank 0 prints the job's line
    by_rank = line["counts_by_rank"]
    assert len(by_rank) == world and by_rank[0] != by_rank[1]  # different camera streams
    assert line["keypoints_per_batch"] == {"harris": by_rank[0][0] + by_rank[1][0], "dog": by_rank[0][1] + by_rank[1][1]}
    # every rank prints where it put itself - {rank, gpu, PCI address, NUMA node, CPUs} - before it
    # allocates its pinned staging buffers (gpu_locality.hpp; no flag needed); rank 0's is also in the job's line
    for r in range(world):
        pl = [json.loads(l.split("placement ", 1)[1]) for l in outs[r][1].splitlines() if l.startswith("Stream: placement ")]
        assert len(pl) == 1 and pl[0]["rank"] == r and pl[0]["gpu"] == 0 and pl[0]["n_cpus"] >= 1 and len(pl[0]["pci_bus_id"]) >= 7, outs[r][1]
        assert pl[0]["bound"] == (pl[0]["numa_node"] >= 0 and "stay on the GPU's node" in pl[0]["note"])
    assert line["placement"]["rank"] == 0 and line["placement"]["pci_bus_id"]
    for r in range(world):
        gr, gc, frames = _read_dump(tmp_path / f"lists{r}.bin")
        assert (gr, gc, len(frames)) == (rows, cols, n)
        assert [sum(f[2] for f in frames), sum(f[3] for f in frames)] == by_rank[r]
        img = synth.frame_np(rows, cols, 2, r)  # frame 2 of camera stream r
        kp, pt, th, td = frames[2]
        want_k = oracle.harris_keypoints(oracle.nms2(oracle.harris_response(img), 5)[0])
        pyr = oracle.Pyramid(img, 4, 1.6)
        want_p = np.concatenate([pyr.extrema(o, 3, 8)[1] for o in range(4)])
        pyr.close()
        assert kp.tobytes() == want_k.tobytes() and pt.tobytes() == want_p.tobytes(), r


@pytest.mark.gpu
def test_stream_reads_a_raw_frame_file(built, tmp_path):
    # --source <file>: a camera stream recorded as raw 8-bit frames (rows x cols bytes each), read cyclically
    import numpy as np

    import oracle
    from visualslam_amd import synth

    rows, cols = 96, 160
    rec = np.stack([synth.frame_np(rows, cols, f, 3, "noise" if f == 1 else "checker") for f in range(3)])
    raw = tmp_path / "camera.y8"
    rec.tofile(raw)
    dump = tmp_path / "lists.bin"
    r = subprocess.run([os.path.join(built, "Stream"), "--mode", "hostfed", "--frames", "5", "--batches", "2", "--warmup", "0", "--rows", str(rows), "--cols", str(cols),
                        "--octaves", "3", "--source", str(raw), "--dump", str(dump)], capture_output=True, text=True, timeout=600, env=_rank_env(0, 1, 29950))
    assert r.returncode == 0, r.stdout + r.stderr
    _, _, frames = _read_dump(dump)
    assert len(frames) == 5
    for f in range(5):  # frame f of the batch is frame f % 3 of the file
        img = rec[f % 3]
        kp, pt, th, td = frames[f]
        want_k = oracle.harris_keypoints(oracle.nms2(oracle.harris_response(img), 5)[0])
        pyr = oracle.Pyramid(img, 3, 1.6)
        want_p = np.concatenate([pyr.extrema(o, 3, 8)[1] for o in range(3)])
        pyr.close()
        assert th == len(want_k) and kp.tobytes() == want_k.tobytes(), f
        assert td == len(want_p) and pt.tobytes() == want_p.tobytes(), f


def test_stream_rendezvous_under_torchrun(built):
    # torch.distributed.run --no-python sets RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* for a native program: the C++
    # ranks find each other on MASTER_PORT + 1 (MASTER_PORT itself is torch's store)
    import sys

    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--no-python", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", "29871", os.path.join(built, "Stream"), "--rdv-selftest"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert sorted(g["rank"] for g in got) == [0, 1, 2] and len({g["id_hash"] for g in got}) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["device", "hostfed"])
def test_stream_list_modes_match_the_oracle_counts(built, mode):
    # --lists orient: FeaturePointLocalization + filterKeypoints for every frame of the batch from the C++ host; a
    # ragged frame size (pitched pyramid planes) and a camera file with a noise frame so that the stages have work
    import numpy as np

    import oracle
    from visualslam_amd import synth

    rows, cols, n = 150, 217, 4  # building.jpg's coarsest octave size: odd width, planes pitched to 224
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        rec = np.stack([synth.frame_np(rows, cols, f, 4, "noise" if f == 2 else "checker") for f in range(n)])
        raw = os.path.join(td, "cam.y8")
        rec.tofile(raw)
        dump = os.path.join(td, "l.bin")
        r = subprocess.run([os.path.join(built, "Stream"), "--mode", mode, "--lists", "orient", "--frames", str(n), "--batches", "2", "--warmup", "0", "--rows", str(rows),
                            "--cols", str(cols), "--octaves", "3", "--source", raw, "--dump", dump], capture_output=True, text=True, timeout=600, env=_rank_env(0, 1, 29960))
        assert r.returncode == 0, r.stdout + r.stderr
        line = json.loads(r.stdout.strip().splitlines()[-1])
        _, _, frames = _read_dump(dump)
    want_oriented = 0
    for f in range(n):
        pyr = oracle.Pyramid(rec[f], 3, 1.6)
        kps = [pyr.keypoints(o, 3) for o in range(3)]
        want_oriented += sum(len(pyr.filter_keypoints(o, kps[o])) for o in range(3))
        pyr.close()
        allk = np.concatenate(kps)
        assert frames[f][3] == len(allk) and frames[f][1].tobytes() == allk.tobytes(), f  # the DoG list = the localized keypoints
    assert line["lists"] == "orient" and line["oriented_points_rank0_last_batch"] == want_oriented > 100
