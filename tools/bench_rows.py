"""Per-image latency of the SURVEY section 8f rows (host-buffer C ABI, PCIe copies included):
pyramid build, initialKeypointDetection (+FeaturePointLocalization), filterKeypoints, SIFT.

    python tools/bench_rows.py [--rows 1080 --cols 1920 --reps 20] > profiles/rNN_rows.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from visualslam_amd import capi, synth  # noqa: E402


def timed(fn, reps):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = fn()
    return (time.perf_counter() - t0) / reps * 1e3, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--octaves", type=int, default=4)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    ctx = capi.Context(0)
    out = {"rows": a.rows, "cols": a.cols, "octaves": a.octaves, "reps": a.reps,
           "note": "host-buffer API: every call uploads its inputs and downloads its outputs (PCIe included)", "frames": {}}
    for kind in ("checker", "noise"):
        img = synth.frame_np(a.rows, a.cols, kind=kind)
        ms_build, p = timed(lambda: ctx.pyramid(img, a.octaves, 1.6), max(a.reps // 4, 2))
        rec = {"pyramid_build_ms": ms_build, "octaves": []}
        for o in range(a.octaves):
            ms_kp, (kp, n) = timed(lambda: p.keypoints(o, 3, cap=1 << 18), a.reps)
            ms_cand, (_, cand, nc) = timed(lambda: p.extrema(o, 3, 8, cap=1 << 18), a.reps)
            ms_dense, (_, _, nd) = timed(lambda: p.extrema_dense(o, 8, cap=1 << 18), max(a.reps // 4, 2))  # extension: dense 3x3x3 test
            ms_f, (fk, nf) = timed(lambda: p.filter_keypoints(o, kp, cap=1 << 18), a.reps)
            ms_s, (desc, ok) = timed(lambda: p.sift_descriptors(o, fk), a.reps) if nf else (0.0, (None, []))
            rec["octaves"].append({"octave": o, "candidates_ge8": int(nc), "keypoints": int(n), "oriented": int(nf),
                                   "descriptors_defined": int(sum(ok)), "extrema_ms": ms_cand, "keypoints_ms": ms_kp,
                                   "filter_keypoints_ms": ms_f, "sift_ms": ms_s,
                                   "dense_3x3x3_ge8": int(nd), "dense_3x3x3_ms": ms_dense})
        rec["total_ms"] = ms_build + sum(r["keypoints_ms"] + r["filter_keypoints_ms"] + r["sift_ms"] for r in rec["octaves"])
        out["frames"][kind] = rec
        p.close()
    ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
