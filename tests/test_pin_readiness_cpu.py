"""The pin stays one command away.  No OpenCV exists here, so what CAN be kept from rotting
is checked on the CPU: the pin script answers --help, the OpenCV-free half of the C++ harness compiles and behaves, the
oracle's fused-multiply-add variant of the f32 stages exists and the committed report of what it changes is reproducible."""
import json
import os
import subprocess
import sys

import numpy as np

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_pin_script_answers_help():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "pin_with_opencv.sh"), "--help"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "--regen" in r.stdout and "OpenCV" in r.stdout


def test_opencv_free_half_of_the_harness_compiles_and_behaves(tmp_path):
    src = os.path.join(ROOT, "tools", "opencv_pin")
    exe = str(tmp_path / "pin_io_check")
    r = subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-Wextra", "-Werror", os.path.join(src, "pin_io_check.cpp"), "-o", exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    # the harness proper includes exactly two OpenCV headers and the shared half; nothing in the tree stands in for OpenCV
    text = open(os.path.join(src, "pin_harness.cpp")).read()
    incs = [l.split("<", 1)[1].split(">")[0] for l in text.splitlines() if l.startswith("#include <")]
    assert [i for i in incs if i.startswith("opencv2/")] == ["opencv2/core.hpp", "opencv2/imgproc.hpp"] and '#include "pin_io.hpp"' in text
    assert not any("opencv2" in d for d, _, _ in os.walk(ROOT) if ".git" not in d)


def test_fma_variant_touches_the_three_f32_primitives_only():
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (64, 80), dtype=np.uint8)
    g = oracle.gaussian_blur_u8(img, 0, 1.6)
    base = oracle.level_gradients(g)
    with oracle.fma_variant(oracle.fma_variant.ATAN):
        atan = oracle.level_gradients(g)
    with oracle.fma_variant(oracle.fma_variant.FILTER):
        filt = oracle.level_gradients(g)
    for k in range(3):  # Sobel and magnitude: no variant touches them
        assert (base[k] == atan[k]).all() and (base[k] == filt[k]).all()
    assert (base[3] == filt[3]).all()  # the filter bit leaves the arctangent alone
    d = base[3] != atan[3]
    assert 0 < d.sum() < 0.02 * d.size and np.abs(base[3] - atan[3]).max() <= 2 ** -15  # last-bit differences, at most one ulp at 256..360 degrees
    mag = base[2]
    a = oracle.blur_f32_roi(mag, 20, 10, 16, 16, 3.0)
    with oracle.fma_variant(oracle.fma_variant.ATAN):
        assert (oracle.blur_f32_roi(mag, 20, 10, 16, 16, 3.0) == a).all()
    with oracle.fma_variant(True):
        b = oracle.blur_f32_roi(mag, 20, 10, 16, 16, 3.0)
    assert (a != b).any() and np.abs(a - b).max() <= 4 * np.spacing(np.float32(np.abs(a).max()))
    assert oracle.lib().vo_get_fma_variant() == 0  # the context managers restored the default
    # the integer rows do not depend on it
    with oracle.fma_variant(True):
        p1 = oracle.Pyramid(img, 2, 1.6)
        r1 = oracle.harris_response(img)
    p0 = oracle.Pyramid(img, 2, 1.6)
    assert (oracle.harris_response(img) == r1).all()
    for o in range(2):
        for l in range(5):
            assert (p0.dog(o, l) == p1.dog(o, l)).all()
        assert p0.keypoints(o, 3).tobytes() == p1.keypoints(o, 3).tobytes()
    p0.close()
    p1.close()


def test_no_histogram_bin_depends_on_the_arctangent_variant():
    # every integer gradient pair a Sobel(ksize 1) of u8 data can produce, both variants of fastAtan32f's polynomial: the
    # values differ in the last bit for ~2 % of the pairs, the 36-bin (Diff_of_Gauss.cpp:126, :352) and 8-bin (:631) indices
    # never - which is why only the kernel that RETURNS the angle (k_level_gradients) carries the switch
    L = oracle.lib()
    v = np.arange(-255, 256, dtype=np.float32)

    def table(mask):
        L.vo_set_fma_variant(mask)
        try:
            return np.array([[L.vo_fast_atan2_deg(float(y), float(x)) for x in v] for y in v], np.float32)
        finally:
            L.vo_set_fma_variant(0)

    a, b = table(0), table(1)
    assert 0 < (a != b).sum() < 0.03 * a.size and np.abs(a - b).max() <= 2 ** -15
    for nb in (36, 8):
        k = np.float32(nb / 360.0)
        assert ((a * k).astype(np.int32) == (b * k).astype(np.int32)).all()


def test_committed_fma_report_is_reproducible_and_backs_the_stated_tolerance(tmp_path):
    # profiles/r06_fma_risk.json (tools/fma_risk_report.py, the four reference images + a 1080p noise frame): the numbers
    # BASELINE.md section 5 / DESIGN section 3 quote.  The quick form (blox + a small noise frame) is re-run here.
    rep = json.load(open(os.path.join(ROOT, "profiles", "r06_fma_risk.json")))
    s = rep["summary"]
    assert [im["image"] for im in rep["images"]][:4] == ["blox", "home", "building", "chessboard"] and "1920x1080" in rep["images"][4]["image"]
    assert s["f1_bin36_differ_fraction"] == 0.0 and s["f1_bin8_differ_fraction"] == 0.0 and s["f3_oriented_points_changed"] == 0
    assert s["f1_max_abs_diff_deg"] <= 2 ** -15 and s["f4_max_abs_diff"] <= 5e-7 and s["f3_oriented_points_baseline"] > 30000
    out = tmp_path / "quick.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fma_risk_report.py"), "--quick", "--out", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    q = json.load(open(out))
    blox_full = rep["images"][0]
    assert q["images"][0] == blox_full  # the same image gives the same counts
    assert q["summary"]["f3_oriented_points_changed"] == 0 and q["summary"]["f1_bin36_differ_fraction"] == 0.0
