#!/usr/bin/env python3
"""How much of rows (f)1 / (f)3 / (f)4 depends on ONE unpinned arithmetic choice: fused multiply-add in the f32 stages.

The reference's f32 arithmetic runs inside OpenCV: `cv::phase` (GaussPyramid.cpp:96) -> hal::fastAtan32f, and
`GaussianBlur` on CV_32F ROIs (Diff_of_Gauss.cpp:348, :616-618) -> the separable f32 filter.  A stock x86-64 OpenCV
dispatches AVX2 + FMA3 variants of both, whose multiply-adds are fused; its SSE2 baseline rounds each multiply and add.
The oracle and the GPU kernels implement the baseline (VERDICT r5 missing #3).  The oracle carries the fused form as a
switch (oracle.fma_variant; vslam_oracle.c g_fma_variant: exactly those three places, nothing else); this script runs
both over the reference's four images and one synthetic 1080p uniform-noise frame and counts what changes:

  (f)1  octaveGradOrient values (processGradients), and how many of them land in another 36-bin / 8-bin histogram bin
  (f)3  filterKeypoints: oriented points (Diff_of_Gauss.cpp:362-366) present in one variant's list and not the other's
  (f)4  SIFT descriptor entries (Diff_of_Gauss.cpp:629-675) of the points both lists share: differing entries, largest
        difference, descriptors whose `defined` flag differs

    python tools/fma_risk_report.py [--out profiles/r06_fma_risk.json] [--quick]

CPU only (the oracle); about two minutes.  Test infrastructure: nothing here is part of the product path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from tests import refimg  # noqa: E402
from visualslam_amd import synth  # noqa: E402


def keyset(pts):
    return set(map(tuple, np.stack([pts[n] for n in ("row", "col", "value", "padding", "octave", "level")], 1).tolist())) if len(pts) else set()


def one_image(name, img, n_oct):
    pyr = oracle.Pyramid(img, n_oct, 1.6)
    rep = {"image": name, "rows": int(img.shape[0]), "cols": int(img.shape[1]), "octaves": n_oct}
    g = {"values": 0, "differ": 0, "max_abs_diff_deg": 0.0, "bin36_differ": 0, "bin8_differ": 0}
    f3 = {"keypoints_in": 0, "oriented_baseline": 0, "oriented_fma": 0, "only_in_baseline": 0, "only_in_fma": 0}
    f4 = {"descriptors_compared": 0, "entries": 0, "entries_differ": 0, "max_abs_diff": 0.0, "descriptors_with_a_difference": 0,
          "defined_flag_differs": 0, "entries_differ_beyond_1e-6": 0}
    for o in range(n_oct):
        # (f)1: orientation planes of all six levels
        for l in range(6):
            a = oracle.level_gradients(pyr.gauss(o, l))[3]
            with oracle.fma_variant(True):
                b = oracle.level_gradients(pyr.gauss(o, l))[3]
            d = a != b
            g["values"] += int(a.size)
            g["differ"] += int(d.sum())
            if d.any():
                g["max_abs_diff_deg"] = max(g["max_abs_diff_deg"], float(np.abs(a[d].astype(np.float64) - b[d]).max()))
            f36, f8 = np.float32(36 / 360.0), np.float32(8 / 360.0)
            g["bin36_differ"] += int(((a * f36).astype(np.int32) != (b * f36).astype(np.int32)).sum())
            g["bin8_differ"] += int(((a * f8).astype(np.int32) != (b * f8).astype(np.int32)).sum())
        # (f)3: the oriented list of this octave's keypoints
        kps = pyr.keypoints(o, 3)
        f3["keypoints_in"] += len(kps)
        base = pyr.filter_keypoints(o, kps)
        with oracle.fma_variant(True):
            fma = pyr.filter_keypoints(o, kps)
        sb, sf = keyset(base), keyset(fma)
        f3["oriented_baseline"] += len(base)
        f3["oriented_fma"] += len(fma)
        f3["only_in_baseline"] += len(sb - sf)
        f3["only_in_fma"] += len(sf - sb)
        # (f)4: descriptors of the points both lists hold, computed under each variant
        common = np.array([p for p in base.tolist() if tuple(p) in sf], dtype=oracle.POINT_DTYPE) if len(base) else base
        if len(common):
            da, oka = pyr.sift_descriptors(o, common)
            with oracle.fma_variant(True):
                db, okb = pyr.sift_descriptors(o, common)
            both = oka & okb
            f4["descriptors_compared"] += int(both.sum())
            f4["defined_flag_differs"] += int((oka != okb).sum())
            a, b = da[both], db[both]
            same = (a == b) | (np.isnan(a) & np.isnan(b))
            f4["entries"] += int(a.size)
            f4["entries_differ"] += int((~same).sum())
            f4["descriptors_with_a_difference"] += int((~same).any(1).sum())
            if (~same).any():
                diff = np.abs(a.astype(np.float64) - b)[~same]
                diff = diff[np.isfinite(diff)]
                if diff.size:
                    f4["max_abs_diff"] = max(f4["max_abs_diff"], float(diff.max()))
                    f4["entries_differ_beyond_1e-6"] += int((diff > 1e-6).sum())
    pyr.close()
    rep["f1_gradient_orientation"] = g
    rep["f3_oriented_points"] = f3
    rep["f4_descriptors"] = f4
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true", help="blox + a 270 x 480 noise frame only (the CPU test's form)")
    args = ap.parse_args()
    oracle.build()
    t0 = time.time()
    images = []
    if args.quick:
        images.append(("blox", refimg.load("blox"), 4))
        images.append(("synthetic uniform noise 480x270", synth.frame_np(270, 480, kind="noise"), 3))
    else:
        for n in refimg.NAMES:
            images.append((n, refimg.load(n), 4))
        images.append(("synthetic uniform noise 1920x1080", synth.frame_np(1080, 1920, kind="noise"), 4))
    reps = []
    for name, img, n_oct in images:
        r = one_image(name, img, n_oct)
        reps.append(r)
        print(f"{name}: orient values {r['f1_gradient_orientation']['differ']} / {r['f1_gradient_orientation']['values']} differ, "
              f"oriented points {r['f3_oriented_points']['oriented_baseline']} vs {r['f3_oriented_points']['oriented_fma']} "
              f"(-{r['f3_oriented_points']['only_in_baseline']} +{r['f3_oriented_points']['only_in_fma']}), "
              f"descriptor entries {r['f4_descriptors']['entries_differ']} / {r['f4_descriptors']['entries']} differ "
              f"(max {r['f4_descriptors']['max_abs_diff']:.3g})", file=sys.stderr, flush=True)

    def tot(sec, key):
        return sum(r[sec][key] for r in reps)

    summary = {
        "f1_orientation_values_differ_fraction": tot("f1_gradient_orientation", "differ") / max(1, tot("f1_gradient_orientation", "values")),
        "f1_max_abs_diff_deg": max(r["f1_gradient_orientation"]["max_abs_diff_deg"] for r in reps),
        "f1_bin36_differ_fraction": tot("f1_gradient_orientation", "bin36_differ") / max(1, tot("f1_gradient_orientation", "values")),
        "f1_bin8_differ_fraction": tot("f1_gradient_orientation", "bin8_differ") / max(1, tot("f1_gradient_orientation", "values")),
        "f3_oriented_points_baseline": tot("f3_oriented_points", "oriented_baseline"),
        "f3_oriented_points_changed": tot("f3_oriented_points", "only_in_baseline") + tot("f3_oriented_points", "only_in_fma"),
        "f3_changed_fraction": (tot("f3_oriented_points", "only_in_baseline") + tot("f3_oriented_points", "only_in_fma")) / max(1, tot("f3_oriented_points", "oriented_baseline")),
        "f4_entries_differ_fraction": tot("f4_descriptors", "entries_differ") / max(1, tot("f4_descriptors", "entries")),
        "f4_max_abs_diff": max(r["f4_descriptors"]["max_abs_diff"] for r in reps),
        "f4_descriptors_with_a_difference_fraction": tot("f4_descriptors", "descriptors_with_a_difference") / max(1, tot("f4_descriptors", "descriptors_compared")),
    }
    out = {
        "what": "rows (f)1 / (f)3 / (f)4 of SURVEY section 8 under the oracle's two f32 variants: baseline (every multiply and add rounded: "
                "OpenCV's SSE2 code, the GPU kernels) vs fused multiply-add in fastAtan32f's polynomial and the separable f32 filter's row / "
                "column passes (OpenCV's AVX2 + FMA3 dispatch).  Integer rows H1-H8, D1-D7 and (f)2 do not touch f32 multiply-adds.",
        "script": "tools/fma_risk_report.py" + (" --quick" if args.quick else ""),
        "seconds": round(time.time() - t0, 1),
        "summary": summary,
        "images": reps,
    }
    text = json.dumps(out, indent=1)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
