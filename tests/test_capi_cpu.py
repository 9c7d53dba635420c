"""C-ABI checks that need no GPU: the library loads, exports every symbol include/vslam.h
declares, its host-side parameter helpers agree with the oracle, and compute entry points
fail loudly (no CPU fallback) when there is no HIP device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from visualslam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    capi.build()
    return capi.lib()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "vslam.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vslam_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vslam.h but not exported"
    assert set(names) == set(capi.SIGNATURES), "capi.SIGNATURES out of sync with include/vslam.h"
    assert lib.vslam_version() == 200
    assert b"k_harris_strip" in lib.vslam_kernel_names()


def test_shipped_library_reads_five_environment_variables_only(lib):
    # the A/B and fault-reproducer switches of rounds 3-5 are compiled out of the shipped library
    # (-DVSLAM_DIAGNOSTICS builds lib/libvslam_diag.so with them); INTEGRATION.md section 5 lists both sets
    def env_names(path):
        data = open(path, "rb").read()
        return sorted(set(m.decode() for m in re.findall(rb"VSLAM_[A-Z0-9_]+", data)))

    assert env_names(capi.LIB_PATH) == ["VSLAM_F32_FUSED", "VSLAM_JOIN_WATCH", "VSLAM_MX", "VSLAM_SIDE_PRIORITY", "VSLAM_STREAM_TUNER"]
    diag = env_names(capi.DIAG_LIB_PATH)
    assert "VSLAM_CAPTURE_NESTED_FORKS" in diag and "VSLAM_TILE_SHAPE" in diag and "VSLAM_AUX_STREAMS" in diag and len(diag) >= 12
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in diag:
        assert n in doc, n + " is read by the diagnostics build but not listed in INTEGRATION.md"
    # both builds export the same C ABI
    import ctypes

    d = ctypes.CDLL(capi.DIAG_LIB_PATH) if not os.environ.get("VSLAM_LIBRARY") else lib
    for n in declared_symbols():
        assert hasattr(d, n), n


def test_struct_sizes_match_reference_types():
    assert capi.POINT_DTYPE.itemsize == 24 == oracle.POINT_DTYPE.itemsize  # SLAM::point, six ints
    assert capi.KP_DTYPE.itemsize == 12


def test_host_helpers_match_oracle(lib):
    for o in range(5):
        for l in range(6):
            s = capi.sigma_at(1.6, o, l)
            assert s == oracle.sigma_at(1.6, o, l)
            n = capi.gauss_ksize_u8(s)
            assert n == oracle.gauss_ksize_u8(s)
            assert (capi.gauss_taps_q8(n, s) == oracle.gauss_taps_q8(n, s)).all()
    for n in (1, 3, 5, 7, 9):
        assert (capi.gauss_taps_q8(n, 0.0) == oracle.gauss_taps_q8(n, 0.0)).all()
    with pytest.raises(capi.VslamError):
        capi.gauss_taps_q8(4, 1.0)
    for r, c in [(1080, 1920), (620, 877), (150, 217), (3, 5), (1, 1)]:
        assert capi.half_size(r, c) == oracle.half_size(r, c)
        assert capi.extrema_lattice(r, c, 3) == oracle.extrema_lattice(r, c, 3)
        assert capi.extrema_lattice(r, c, 5) == oracle.extrema_lattice(r, c, 5)
    for r, c in [(256, 256), (384, 512), (600, 868), (1240, 1754), (1080, 1920)]:
        assert capi.auto_num_octaves(r, c) == oracle.auto_num_octaves(r, c)


def test_batch_layout_1080p(lib):
    p = capi.default_params(1080, 1920)
    assert (p.n_octaves, p.sigma0, p.extrema_window, p.min_contrast) == (4, 1.6, 3, 8)
    assert abs(p.harris_k - 0.04) < 1e-9
    L = capi.batch_layout(p)
    assert [(L.rows[o], L.cols[o]) for o in range(4)] == [(2160, 3840), (1080, 1920), (540, 960), (270, 480)]
    # SURVEY section 8d algorithmic bytes
    assert L.algorithmic_bytes_harris == 12_441_600
    assert L.algorithmic_bytes_dog == 123_249_600
    assert L.pyramid_frame_bytes >= 11 * 11_016_000 and L.pyramid_frame_bytes % 256 == 0
    assert sum(3 * L.lat_rows[o] * L.lat_cols[o] for o in range(4)) == 3_672_000
    assert L.octave_offset[1] == 11 * 2160 * 3840
    assert [L.pitch[o] for o in range(4)] == [3840, 1920, 960, 480]  # multiples of 16: pitch == cols
    odd = capi.batch_layout(capi.default_params(1240, 1754))  # the reference's chessboard.png
    assert [(odd.cols[o], odd.pitch[o]) for o in range(4)] == [(3508, 3520), (1754, 1760), (877, 880), (438, 448)]
    assert odd.octave_offset[1] == 11 * 2480 * 3520 and odd.octave_offset[1] % 16 == 0
    bad = capi.default_params(0, 10)
    with pytest.raises(capi.VslamError):
        capi.batch_layout(bad)


def test_no_gpu_means_loud_failure(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.VslamError) as e:
        capi.Context(0)
    assert e.value.status == -2  # VSLAM_ERR_HIP: there is no CPU fallback


def test_null_arguments_are_rejected_without_touching_the_gpu(lib):
    assert lib.vslam_ctx_create(0, None, None) == -1
    assert lib.vslam_ctx_destroy(None) == -1
    assert lib.vslam_gauss_taps_q8(3, 0.0, None) == -1
    assert lib.vslam_pyramid_destroy(None) == -1
    assert lib.vslam_batch_layout_query(None, None) == -1
    assert lib.vslam_last_error(None) == b"null context"


def test_default_list_capacities_follow_the_frame_area(lib):
    # the fixed 2^18 default overflowed at 3840x2160
    caps = lambda r, c: (lambda p: (p.harris_cap, p.dog_cap, p.oriented_cap))(capi.default_params(r, c))
    assert caps(1080, 1920) == (262144, 262144, 65536)
    assert caps(2160, 3840) == (1040384, 1040384, 262144)
    assert caps(96, 160) == (65536, 65536, 16384)  # floors: an all-noise small frame still fits
    assert capi.default_params(1080, 1920).extrema_dense == 0


def test_batch_out_required_sizes(lib):
    p = capi.default_params(1080, 1920)
    L = capi.batch_layout(p)
    n = 256
    z = capi.batch_out_required(p, n)
    N = 1080 * 1920
    assert z.struct_size == C.sizeof(capi.BatchOut)
    assert (z.response_bytes, z.nms_mask_bytes, z.nms2_bytes) == (4 * n * N, n * N, 4 * n * N)
    assert (z.harris_kps_bytes, z.harris_counts_bytes) == (12 * n * p.harris_cap, 4 * n)
    assert (z.pyramid_bytes, z.extrema_bits_bytes) == (n * L.pyramid_frame_bytes, 8 * n * L.bits_frame_words)
    assert (z.dog_points_bytes, z.dog_counts_bytes) == (24 * n * p.dog_cap, 4 * n)
    assert (z.oriented_points_bytes, z.oriented_counts_bytes, z.oriented_survivors_bytes) == (24 * n * p.oriented_cap, 4 * n, 4 * n)
    assert (z.descriptors_bytes, z.descriptor_defined_bytes) == (512 * n * p.oriented_cap, n * p.oriented_cap)
    with pytest.raises(capi.VslamError):
        capi.batch_out_required(p, 0)
    # the header's struct: one size_t, then (pointer, size_t) per output, in the header's order
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "vslam.h")).read(), flags=re.S)
    body = text[text.index("size_t struct_size;"): text.index("} vslam_batch_out;")]
    fields = re.findall(r"\b(\w+);", body)
    want = ["struct_size"] + [f for n_ in capi.BATCH_OUT_FIELDS for f in (n_, n_ + "_bytes")]
    assert fields == want


def test_points16_expand_is_the_inverse_of_the_device_packing(lib):
    # vslam_points16_expand (host, no GPU): the 16-byte record {row, col, value, level | octave << 8 | padding << 16} back to
    # SLAM::point - every field combination the detector writes (padding 0 / 1, octave < VSLAM_MAX_OCTAVES, level < 6), extreme
    # coordinates and values, and the record layout itself (tests/test_gpu_batch.py::test_pack_points16 checks the device side)
    import ctypes as C

    rng = np.random.default_rng(3)
    n = 4096
    pts = np.zeros(n, capi.POINT_DTYPE)
    pts["row"] = rng.integers(-(1 << 31), (1 << 31) - 1, n)
    pts["col"] = rng.integers(-(1 << 31), (1 << 31) - 1, n)
    pts["value"] = rng.integers(-(1 << 31), (1 << 31) - 1, n)
    pts["padding"] = rng.integers(0, 2, n)
    pts["octave"] = rng.integers(0, 10, n)
    pts["level"] = rng.integers(0, 6, n)
    packed = np.zeros((n, 4), np.uint32)
    packed[:, 0] = pts["row"].view(np.uint32)
    packed[:, 1] = pts["col"].view(np.uint32)
    packed[:, 2] = pts["value"].view(np.uint32)
    packed[:, 3] = pts["level"].astype(np.uint32) | (pts["octave"].astype(np.uint32) << 8) | (pts["padding"].astype(np.uint32) << 16)
    assert capi.points16_expand(packed).tobytes() == pts.tobytes()
    assert capi.points16_expand(np.zeros((0, 4), np.uint32)).shape == (0,)
    lib.vslam_points16_expand(None, 5, None)  # null arguments: nothing happens


def test_dense_mode_layout(lib):
    # extension: the dense 3x3x3 scan's bitmask has one site per pixel of levels 1..3
    p = capi.default_params(1080, 1920, extrema_dense=1)
    L = capi.batch_layout(p)
    assert [(L.lat_rows[o], L.lat_cols[o], L.lat_words[o]) for o in range(4)] == [(2160, 3840, 60), (1080, 1920, 30), (540, 960, 15), (270, 480, 8)]
    assert L.bits_frame_words == 3 * (2160 * 60 + 1080 * 30 + 540 * 15 + 270 * 8)
    for bad in (dict(localize=1), dict(extrema_window=5), dict(localize=1, orient=1)):
        with pytest.raises(capi.VslamError):
            capi.batch_layout(capi.default_params(64, 64, extrema_dense=1, **bad))


def test_header_is_plain_c_and_ctypes_mirrors_its_struct_layouts(lib, tmp_path):
    # The boundary is a C ABI: include/vslam.h must compile as C99 (no C++ leaking in), a C program must link against the
    # library and get the same answers from its host-side helpers as the Python binding, and every struct the binding
    # mirrors with ctypes must have the C compiler's size and field offsets (a mismatch would corrupt arguments silently).
    import shutil
    import subprocess

    if not shutil.which("gcc"):
        pytest.skip("no gcc")
    structs = {"vslam_params": capi.Params, "vslam_batch_layout": capi.BatchLayout, "vslam_batch_out": capi.BatchOut,
               "vslam_host_lists": capi.HostLists, "vslam_pyramid_info": capi.PyramidInfo}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vslam.h"', "int main(void) {"]
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, *_ in ct._fields_:
            lines.append(f'  printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('  printf("\\n");')
    lines += ["  vslam_params p; vslam_batch_layout L; vslam_batch_out need;",
              "  vslam_params_default(&p, 1080, 1920);",
              "  if (vslam_batch_layout_query(&p, &L) != VSLAM_OK || vslam_batch_out_required(&p, 256, &need) != VSLAM_OK) return 2;",
              '  printf("helpers %d %d %u %u %zu %zu %d\\n", vslam_version(), vslam_gauss_ksize_u8(1.6), p.harris_cap, p.dog_cap, (size_t)L.pyramid_frame_bytes, need.pyramid_bytes, (int)sizeof(vslam_point));',
              "  return 0;", "}"]
    src = tmp_path / "abi.c"
    src.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "abi"
    libdir = os.path.join(ROOT, "visualslam_amd", "lib")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", libdir, "-lvslam", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rows = {l.split()[0]: l.split()[1:] for l in out.stdout.strip().splitlines()}
    for cname, ct in structs.items():
        want = [C.sizeof(ct)] + [getattr(ct, f[0]).offset for f in ct._fields_]
        assert [int(v) for v in rows[cname]] == want, (cname, rows[cname], want)
    p = capi.default_params(1080, 1920)
    Lp, need = capi.batch_layout(p), capi.batch_out_required(p, 256)
    assert [int(v) for v in rows["helpers"]] == [200, oracle.gauss_ksize_u8(1.6), p.harris_cap, p.dog_cap, Lp.pyramid_frame_bytes, need.pyramid_bytes, 24]
