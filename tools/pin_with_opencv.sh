#!/usr/bin/env bash
# Pin the CPU oracle to a REAL OpenCV in one command.  OpenCV is absent from the build image and from
# the GPU box, so this has never run there; it is for the first machine that has one (C++ dev files and/or python cv2).
#
#   tools/pin_with_opencv.sh            # build the C++ harness, run the cross-check, report
#   tools/pin_with_opencv.sh --help     # this text
#   tools/pin_with_opencv.sh --regen    # ... and, when every check passes, regenerate tests/golden/*.npz with the
#                                       #     OpenCV-made planes and an `opencv_version` stamp
#
# What it does: (1) builds tools/opencv_pin/pin_harness (the three call sites cv2 cannot reach from Python) when
# pkg-config or OPENCV_CXXFLAGS / OPENCV_LIBS find an OpenCV; (2) runs tests/test_opencv_crosscheck.py: every OpenCV call
# of the hot path, in the reference's order, against the oracle function that restates it; (3) with --regen runs
# tests/golden/make_golden.py --opencv.  A failing check names the recalled rule (the double-dagger
# items) that is wrong for THIS OpenCV build; each rule lives in one function of oracle/vslam_oracle.c and one of
# visualslam_amd/csrc (DESIGN "Oracle").
set -euo pipefail
cd "$(dirname "$0")/.."
regen=0
if [ "${1:-}" = "--help" ] || [ "${1:-}" = "-h" ]; then
    sed -n '2,15p' "$0" | sed 's/^# \{0,1\}//'
    exit 0
fi
[ "${1:-}" = "--regen" ] && regen=1

cxxflags="${OPENCV_CXXFLAGS:-}"
libs="${OPENCV_LIBS:-}"
if [ -z "$cxxflags$libs" ] && command -v pkg-config >/dev/null 2>&1; then
    for pc in opencv4 opencv; do
        if pkg-config --exists "$pc"; then
            cxxflags="$(pkg-config --cflags "$pc")"
            libs="$(pkg-config --libs "$pc")"
            break
        fi
    done
fi
if [ -n "$cxxflags$libs" ]; then
    echo "building tools/opencv_pin/pin_harness"
    g++ -O2 -std=c++17 $cxxflags tools/opencv_pin/pin_harness.cpp -o tools/opencv_pin/pin_harness $libs
    echo "OpenCV (C++): $(tools/opencv_pin/pin_harness version)"
else
    echo "no C++ OpenCV found (pkg-config opencv4 / OPENCV_CXXFLAGS + OPENCV_LIBS): the harness-backed checks will be skipped" >&2
fi
if ! python -c 'import cv2; print("OpenCV (python):", cv2.__version__)'; then
    echo "python has no cv2: nothing to pin against" >&2
    exit 3
fi
# the f32 checks try BOTH arithmetic variants of the oracle (every op rounded / fused multiply-add: oracle.fma_variant) and
# print which one this OpenCV build computes as an OPENCV_F32_VARIANT line (-s lets it through)
python -m pytest tests/test_opencv_crosscheck.py -q -rs -s
if [ "$regen" = 1 ]; then
    python tests/golden/make_golden.py --opencv
    python -m pytest tests/test_golden_cpu.py -q
fi
echo "oracle pinned against this OpenCV: record the version lines and the OPENCV_F32_VARIANT line above in DESIGN.md (Oracle) and drop 'parity unpinned'"
echo "(OPENCV_F32_VARIANT ... 1: this build fuses its f32 multiply-adds - rows (f)1/(f)3/(f)4 then hold within BASELINE.md section 5's tolerance: profiles/r06_fma_risk.json)"
