"""ctypes binding of the C ABI in include/vslam.h (visualslam_amd/lib/libvslam.so).

This is plumbing only: argument marshalling for numpy (host entry points) and for torch
CUDA tensors (device entry points).  All compute happens in the hand-written HIP kernels
behind the ABI; there is no fallback -- if the library is missing or HIP cannot run, the
calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSLAM_LIBRARY: development switch for A/B timing of two builds of the same library on one box
# (tools/ab.sh); everything else uses the in-tree build.
LIB_PATH = os.environ.get("VSLAM_LIBRARY") or os.path.join(_HERE, "lib", "libvslam.so")
# the diagnostics build (-DVSLAM_DIAGNOSTICS: the A/B environment switches); a test / tool that needs one starts its child
# process with VSLAM_LIBRARY=DIAG_LIB_PATH
DIAG_LIB_PATH = os.path.join(_HERE, "lib", "libvslam_diag.so")
CSRC = os.path.join(_HERE, "csrc")

MAX_OCTAVES = 16
NUM_LEVELS = 6
NUM_DOGS = 5

POINT_DTYPE = np.dtype([(n, "<i4") for n in ("row", "col", "value", "padding", "octave", "level")], align=True)
KP_DTYPE = np.dtype([("row", "<i4"), ("col", "<i4"), ("response", "<f4")], align=True)


class VslamError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str = ""):
        self.status = status
        super().__init__(f"{what}: status {status} ({_status_string(status)}){': ' + detail if detail else ''}")


class Params(C.Structure):
    _fields_ = [
        ("rows", C.c_int), ("cols", C.c_int), ("n_octaves", C.c_int), ("sigma0", C.c_double),
        ("harris_k", C.c_float), ("do_harris", C.c_int), ("extrema_window", C.c_int),
        ("min_contrast", C.c_int), ("localize", C.c_int), ("orient", C.c_int), ("harris_cap", C.c_uint32), ("dog_cap", C.c_uint32),
        ("oriented_cap", C.c_uint32), ("extrema_dense", C.c_int),
    ]


class BatchLayout(C.Structure):
    _fields_ = [
        ("n_octaves", C.c_int),
        ("rows", C.c_int * MAX_OCTAVES), ("cols", C.c_int * MAX_OCTAVES),
        ("lat_rows", C.c_int * MAX_OCTAVES), ("lat_cols", C.c_int * MAX_OCTAVES), ("lat_words", C.c_int * MAX_OCTAVES),
        ("pitch", C.c_int * MAX_OCTAVES),
        ("octave_offset", C.c_size_t * MAX_OCTAVES), ("pyramid_frame_bytes", C.c_size_t),
        ("bits_offset", C.c_size_t * MAX_OCTAVES), ("bits_frame_words", C.c_size_t),
        ("algorithmic_bytes_harris", C.c_size_t), ("algorithmic_bytes_dog", C.c_size_t),
    ]


BATCH_OUT_FIELDS = ("response", "nms_mask", "nms2", "harris_kps", "harris_counts", "pyramid", "extrema_bits", "dog_points", "dog_counts",
                    "oriented_points", "oriented_counts", "oriented_survivors", "descriptors", "descriptor_defined")


class BatchOut(C.Structure):
    """vslam_batch_out: struct_size, then (pointer, bytes behind it) per output."""
    _fields_ = [("struct_size", C.c_size_t)] + [f for n in BATCH_OUT_FIELDS for f in ((n, C.c_void_p), (n + "_bytes", C.c_size_t))]


class HostLists(C.Structure):
    """vslam_host_lists: the packed host-side lists of vslam_detect_batch_host."""
    _fields_ = [("struct_size", C.c_size_t),
                ("harris", C.c_void_p), ("harris_bytes", C.c_size_t), ("harris_offsets", C.c_void_p), ("harris_counts", C.c_void_p),
                ("dog", C.c_void_p), ("dog_bytes", C.c_size_t), ("dog_offsets", C.c_void_p), ("dog_counts", C.c_void_p)]


class PyramidInfo(C.Structure):
    _fields_ = [
        ("n_octaves", C.c_int), ("n_levels", C.c_int), ("n_dogs", C.c_int), ("sigma0", C.c_double),
        ("rows", C.c_int * MAX_OCTAVES), ("cols", C.c_int * MAX_OCTAVES),
        ("sigma", (C.c_double * NUM_LEVELS) * MAX_OCTAVES), ("ksize", (C.c_int * NUM_LEVELS) * MAX_OCTAVES),
    ]


# name -> (restype, argtypes); mirrors include/vslam.h one to one
_P, _I, _Z, _D, _F = C.c_void_p, C.c_int, C.c_size_t, C.c_double, C.c_float
SIGNATURES = {
    "vslam_version": (_I, []),
    "vslam_status_string": (C.c_char_p, [_I]),
    "vslam_ctx_create": (_I, [_I, _P, C.POINTER(_P)]),
    "vslam_ctx_destroy": (_I, [_P]),
    "vslam_ctx_sync": (_I, [_P]),
    "vslam_last_error": (C.c_char_p, [_P]),
    "vslam_gauss_ksize_u8": (_I, [_D]),
    "vslam_gauss_taps_q8": (_I, [_I, _D, _P]),
    "vslam_sigma_at": (_D, [_D, _I, _I]),
    "vslam_auto_num_octaves": (_I, [_I, _I]),
    "vslam_half_size": (None, [_I, _I, C.POINTER(_I), C.POINTER(_I)]),
    "vslam_extrema_lattice": (None, [_I, _I, _I, C.POINTER(_I), C.POINTER(_I)]),
    "vslam_gaussian_blur_u8": (_I, [_P, _P, _I, _I, _Z, _I, _D, _P, _Z]),
    "vslam_sobel_k1_u8_f32": (_I, [_P, _P, _I, _I, _Z, _I, _I, _P, _Z]),
    "vslam_resize_linear2x_u8": (_I, [_P, _P, _I, _I, _Z, _P, _Z]),
    "vslam_resize_nearest_half_u8": (_I, [_P, _P, _I, _I, _Z, _P, _Z]),
    "vslam_convert_scale_abs_f32": (_I, [_P, _P, _I, _I, _Z, _P, _Z]),
    "vslam_harris_from_grad_f32": (_I, [_P, _P, _P, _I, _I, _Z, _F, _I, _P, _Z]),
    "vslam_harris_response_u8": (_I, [_P, _P, _I, _I, _Z, _F, _I, _P, _Z]),
    "vslam_nms_strict_u8": (_I, [_P, _P, _I, _I, _Z, _I, _P, _Z]),
    "vslam_nms_strict_f32": (_I, [_P, _P, _I, _I, _Z, _I, _P, _Z]),
    "vslam_nms2_f32": (_I, [_P, _P, _I, _I, _Z, _I, _P, _Z, C.POINTER(_F)]),
    "vslam_harris_keypoints_u8": (_I, [_P, _P, _I, _I, _Z, _F, _P, _Z, C.POINTER(_Z)]),
    "vslam_pyramid_build_u8": (_I, [_P, _P, _I, _I, _Z, _I, _D, C.POINTER(_P)]),
    "vslam_pyramid_destroy": (_I, [_P]),
    "vslam_pyramid_get_info": (_I, [_P, C.POINTER(PyramidInfo)]),
    "vslam_pyramid_get_base": (_I, [_P, _I, _P, _Z]),
    "vslam_pyramid_get_gauss": (_I, [_P, _I, _I, _P, _Z]),
    "vslam_pyramid_get_dog": (_I, [_P, _I, _I, _P, _Z]),
    "vslam_pyramid_get_gradients": (_I, [_P, _I, _I, _P, _P, _P, _P, _Z]),
    "vslam_dog_extrema": (_I, [_P, _P, _I, _I, _I, _P, _P, _Z, C.POINTER(_Z)]),
    "vslam_dog_extrema_dense": (_I, [_P, _P, _I, _I, _P, _P, _Z, C.POINTER(_Z)]),
    "vslam_dog_keypoints": (_I, [_P, _P, _I, _I, _P, _Z, C.POINTER(_Z)]),
    "vslam_localize_points": (_I, [_P, _P, _Z, _P, _P]),
    "vslam_filter_keypoints": (_I, [_P, _P, _I, _P, _Z, _P, _Z, C.POINTER(_Z)]),
    "vslam_cos_sin_deg": (None, [_F, C.POINTER(_F), C.POINTER(_F)]),
    "vslam_rotated_window_points": (_I, [_I, _I, _I, _F, _P]),
    "vslam_sift_descriptors": (_I, [_P, _P, _I, _P, _Z, _P, _P]),
    "vslam_descriptor_file_write": (_I, [C.c_char_p, _P, _Z]),
    "vslam_edge_response_windows": (_I, [_P, _P, _P, _I, _Z, _P]),
    "vslam_structure_matrix_windows": (_I, [_P, _P, _P, _I, _Z, _P]),
    "vslam_params_default": (None, [C.POINTER(Params), _I, _I]),
    "vslam_batch_layout_query": (_I, [C.POINTER(Params), C.POINTER(BatchLayout)]),
    "vslam_batch_out_required": (_I, [C.POINTER(Params), _I, C.POINTER(BatchOut)]),
    "vslam_detect_batch_dev": (_I, [_P, C.POINTER(Params), _P, _Z, _I, C.POINTER(BatchOut)]),
    "vslam_ctx_follow": (_I, [_P, _P]),
    "vslam_ctx_set_matrix_path": (_I, [_P, _I]),
    "vslam_ctx_get_matrix_path": (_I, [_P]),
    "vslam_ctx_side_stream_report": (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "vslam_ctx_tune_side_streams": (_I, [_P, _I]),
    "vslam_ctx_join_watch_report": (_I, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "vslam_ctx_set_side_stream_priority": (_I, [_P, _I]),
    "vslam_ctx_set_f32_fused": (_I, [_P, _I]),
    "vslam_ctx_get_f32_fused": (_I, [_P]),
    "vslam_ctx_set_join_watch": (_I, [_P, _I]),
    "vslam_ctx_pin_side_streams": (_I, [_P, _I]),
    "vslam_detect_batch_host": (_I, [_P, C.POINTER(Params), _P, _Z, _I, C.POINTER(HostLists)]),
    "vslam_pack_lists_dev": (_I, [_P, _P, _Z, C.c_uint32, _P, _I, _P, _Z, _P]),
    "vslam_pack_points16_dev": (_I, [_P, _P, C.c_uint32, _P, _I, _P, _Z, _P]),
    "vslam_points16_expand": (None, [_P, _Z, _P]),
    "vslam_count_totals_dev": (_I, [_P, _P, _P, _I, _P]),
    "vslam_kernel_timing_enable": (_I, [_P, C.c_char_p]),
    "vslam_kernel_timing_read": (_I, [_P, C.POINTER(_I), C.POINTER(_D)]),
    "vslam_kernel_names": (C.c_char_p, []),
}


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU): lib/libvslam.so, and beside it
    lib/libvslam_diag.so (DIAG_LIB_PATH: the same sources with -DVSLAM_DIAGNOSTICS, for the tests / tools that need an A/B
    switch - never loaded by default)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "vslam.h")]
    newest = max(os.path.getmtime(s) for s in srcs)
    stale = force or any(not os.path.exists(q) or os.path.getmtime(q) < newest for q in (LIB_PATH, DIAG_LIB_PATH))
    if stale and not os.environ.get("VSLAM_LIBRARY"):
        r = subprocess.run(["make", "-j4", "-C", CSRC] + (["-B"] if force else []), capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building libvslam.so failed:\n" + r.stdout + r.stderr)
    return LIB_PATH


_lib = None


def lib():
    """Load libvslam.so (raises if it has not been built: there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing -- run __graft_entry__.build() (the product has no CPU fallback)")
        # One HIP runtime per process: torch bundles its own libamdhip64.so.7 / libhsa-runtime64
        # (same sonames as /opt/rocm).  If libvslam.so pulled in /opt/rocm's copy first, torch
        # would later mix it with its bundled HSA runtime and HIP fails to initialise.  So in a
        # Python process that has torch, let torch load its runtime first; libvslam.so then binds
        # to that same copy.  (C++ callers without torch simply use /opt/rocm's runtime.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _status_string(s: int) -> str:
    try:
        return lib().vslam_status_string(s).decode()
    except Exception:
        return "?"


def gauss_ksize_u8(sigma: float) -> int:
    return lib().vslam_gauss_ksize_u8(float(sigma))


def gauss_taps_q8(n: int, sigma: float) -> np.ndarray:
    t = np.zeros(max(n, 1), np.uint16)
    rc = lib().vslam_gauss_taps_q8(n, float(sigma), t.ctypes.data)
    if rc:
        raise VslamError(rc, "vslam_gauss_taps_q8")
    return t


def sigma_at(sigma0: float, octave: int, level: int) -> float:
    return lib().vslam_sigma_at(float(sigma0), octave, level)


def auto_num_octaves(rows: int, cols: int) -> int:
    return lib().vslam_auto_num_octaves(rows, cols)


def half_size(rows: int, cols: int):
    r, c = C.c_int(), C.c_int()
    lib().vslam_half_size(rows, cols, C.byref(r), C.byref(c))
    return r.value, c.value


def extrema_lattice(rows: int, cols: int, window: int = 3):
    r, c = C.c_int(), C.c_int()
    lib().vslam_extrema_lattice(rows, cols, window, C.byref(r), C.byref(c))
    return r.value, c.value


def cos_sin_deg(theta_deg: float):
    c, s = C.c_float(), C.c_float()
    lib().vslam_cos_sin_deg(float(theta_deg), C.byref(c), C.byref(s))
    return np.float32(c.value), np.float32(s.value)


def rotated_window_points(cx: int, cy: int, window: int, theta_deg: float) -> np.ndarray:
    """Rotation::getRotatedWindowPoints: int32 [(window+1)^2, 2] = (x, y), rows outer."""
    xy = np.zeros(((max(window, 0) + 1) ** 2, 2), np.int32)
    rc = lib().vslam_rotated_window_points(int(cx), int(cy), int(window), float(theta_deg), xy.ctypes.data)
    if rc:
        raise VslamError(rc, "vslam_rotated_window_points")
    return xy


def descriptor_file_write(path: str, desc) -> None:
    """featureDescriptors.dat (Diff_of_Gauss.cpp:837-863): int32 {n, 128, 24} + n x 128 float32."""
    d = np.ascontiguousarray(desc, dtype=np.float32).reshape(-1, 128)
    rc = lib().vslam_descriptor_file_write(os.fsencode(path), d.ctypes.data, d.shape[0])
    if rc:
        raise VslamError(rc, "vslam_descriptor_file_write", path)


def points16_expand(packed) -> np.ndarray:
    """vslam_points16_expand (host): [m, 4] int32 / uint32 packed records -> [m] POINT_DTYPE records."""
    a = np.ascontiguousarray(packed).view(np.uint32).reshape(-1, 4)
    out = np.zeros(len(a), POINT_DTYPE)
    lib().vslam_points16_expand(a.ctypes.data, len(a), out.ctypes.data)
    return out


def default_params(rows: int, cols: int, **kw) -> Params:
    p = Params()
    lib().vslam_params_default(C.byref(p), rows, cols)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def batch_layout(p: Params) -> BatchLayout:
    L = BatchLayout()
    rc = lib().vslam_batch_layout_query(C.byref(p), C.byref(L))
    if rc:
        raise VslamError(rc, "vslam_batch_layout_query")
    return L


def batch_out_required(p: Params, n_frames: int) -> BatchOut:
    """Bytes every output buffer needs for n_frames frames (the x_bytes fields; pointers stay NULL)."""
    z = BatchOut()
    rc = lib().vslam_batch_out_required(C.byref(p), int(n_frames), C.byref(z))
    if rc:
        raise VslamError(rc, "vslam_batch_out_required")
    return z


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.ndim != 2:
        raise ValueError("expected a 2-D uint8 image")
    return a


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2:
        raise ValueError("expected a 2-D float32 image")
    return a


STREAM_LEGACY = 1  # VSLAM_STREAM_LEGACY == hipStreamLegacy: the device's NULL stream


class Context:
    """One vslam_ctx: one GPU, one HIP stream.

    stream: None -> the context owns a private non-blocking stream (results are complete after
    ``sync()``; nothing orders it against torch's streams).  An integer is a hipStream_t handle, e.g.
    ``torch.cuda.current_stream().cuda_stream``; torch's default stream has handle 0, which is passed
    on as the legacy NULL stream (VSLAM_STREAM_LEGACY) -- NOT as "no stream" -- so that tensors made
    and read on that stream are ordered with the kernels by the stream itself."""

    def __init__(self, device: int = 0, stream: int | None = None):
        h = C.c_void_p()
        if stream is None:
            sp = None
        else:
            sp = C.c_void_p(int(stream) or STREAM_LEGACY)
        rc = lib().vslam_ctx_create(device, sp, C.byref(h))
        if rc:
            raise VslamError(rc, "vslam_ctx_create", "no usable HIP device" if rc == -2 else "")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            lib().vslam_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc, what):
        if rc:
            raise VslamError(rc, what, lib().vslam_last_error(self._h).decode())

    def sync(self):
        self._chk(lib().vslam_ctx_sync(self._h), "vslam_ctx_sync")

    # ---- host-buffer primitives
    def gaussian_blur_u8(self, img, ksize: int, sigma: float):
        img = _u8(img)
        out = np.empty_like(img)
        self._chk(lib().vslam_gaussian_blur_u8(self._h, img.ctypes.data, *img.shape, img.strides[0], ksize, float(sigma), out.ctypes.data, out.strides[0]), "vslam_gaussian_blur_u8")
        return out

    def sobel_k1(self, img, dx: int, dy: int):
        img = _u8(img)
        out = np.empty(img.shape, np.float32)
        self._chk(lib().vslam_sobel_k1_u8_f32(self._h, img.ctypes.data, *img.shape, img.strides[0], dx, dy, out.ctypes.data, out.strides[0]), "vslam_sobel_k1_u8_f32")
        return out

    def resize_linear2x(self, img):
        img = _u8(img)
        out = np.empty((2 * img.shape[0], 2 * img.shape[1]), np.uint8)
        self._chk(lib().vslam_resize_linear2x_u8(self._h, img.ctypes.data, *img.shape, img.strides[0], out.ctypes.data, out.strides[0]), "vslam_resize_linear2x_u8")
        return out

    def resize_nearest_half(self, img):
        img = _u8(img)
        out = np.empty(half_size(*img.shape), np.uint8)
        self._chk(lib().vslam_resize_nearest_half_u8(self._h, img.ctypes.data, *img.shape, img.strides[0], out.ctypes.data, max(out.strides[0], 1)), "vslam_resize_nearest_half_u8")
        return out

    def convert_scale_abs(self, x):
        x = _f32(x)
        out = np.empty(x.shape, np.uint8)
        self._chk(lib().vslam_convert_scale_abs_f32(self._h, x.ctypes.data, *x.shape, x.strides[0], out.ctypes.data, out.strides[0]), "vslam_convert_scale_abs_f32")
        return out

    # ---- Harris
    def harris_from_grad(self, ix, iy, k: float = 0.04, window: int = 3):
        ix, iy = _f32(ix), _f32(iy)
        if ix.shape != iy.shape:
            raise ValueError("Ix and Iy differ in shape")
        out = np.empty(ix.shape, np.float32)
        self._chk(lib().vslam_harris_from_grad_f32(self._h, ix.ctypes.data, iy.ctypes.data, *ix.shape, ix.strides[0], k, window, out.ctypes.data, out.strides[0]), "vslam_harris_from_grad_f32")
        return out

    def harris_response(self, img, k: float = 0.04, window: int = 3):
        img = _u8(img)
        out = np.empty(img.shape, np.float32)
        self._chk(lib().vslam_harris_response_u8(self._h, img.ctypes.data, *img.shape, img.strides[0], k, window, out.ctypes.data, out.strides[0]), "vslam_harris_response_u8")
        return out

    def nms_strict(self, x, window: int = 3):
        x = np.asarray(x)
        out = np.empty(x.shape, np.uint8)
        if x.dtype == np.uint8:
            x = _u8(x)
            rc = lib().vslam_nms_strict_u8(self._h, x.ctypes.data, *x.shape, x.strides[0], window, out.ctypes.data, out.strides[0])
        else:
            x = _f32(x)
            rc = lib().vslam_nms_strict_f32(self._h, x.ctypes.data, *x.shape, x.strides[0], window, out.ctypes.data, out.strides[0])
        self._chk(rc, "vslam_nms_strict")
        return out

    def nms2(self, resp, window: int = 5):
        resp = _f32(resp)
        out = np.empty(resp.shape, np.float32)
        tm = C.c_float()
        self._chk(lib().vslam_nms2_f32(self._h, resp.ctypes.data, *resp.shape, resp.strides[0], window, out.ctypes.data, out.strides[0], C.byref(tm)), "vslam_nms2_f32")
        return out, tm.value

    def harris_keypoints(self, img, k: float = 0.04, cap: int = 1 << 20):
        img = _u8(img)
        out = np.zeros(cap, KP_DTYPE)
        n = C.c_size_t()
        self._chk(lib().vslam_harris_keypoints_u8(self._h, img.ctypes.data, *img.shape, img.strides[0], k, out.ctypes.data, cap, C.byref(n)), "vslam_harris_keypoints_u8")
        return out[: min(n.value, cap)], n.value

    # ---- DoG
    def localize_points(self, diffs):
        """FeaturePointLocalization for n candidates: diffs int32 [n, 4] = (d_x, d_y, d_scale, value)
        -> (keep bool [n], value int32 [n])."""
        d = np.ascontiguousarray(diffs, dtype=np.int32).reshape(-1, 4)
        n = d.shape[0]
        keep = np.zeros(n, np.int32)
        val = np.zeros(n, np.int32)
        self._chk(lib().vslam_localize_points(self._h, d.ctypes.data, n, keep.ctypes.data, val.ctypes.data), "vslam_localize_points")
        return keep.astype(bool), val

    def edge_response_windows(self, gx_windows, gy_windows):
        """computeEdgeResponse for n gathered windows: f32 [n, elems] each -> f32 [n]."""
        gx = np.ascontiguousarray(gx_windows, dtype=np.float32)
        gy = np.ascontiguousarray(gy_windows, dtype=np.float32)
        assert gx.shape == gy.shape and gx.ndim == 2
        out = np.zeros(gx.shape[0], np.float32)
        self._chk(lib().vslam_edge_response_windows(self._h, gx.ctypes.data, gy.ctypes.data, gx.shape[1], gx.shape[0], out.ctypes.data), "vslam_edge_response_windows")
        return out

    def structure_matrix_windows(self, gx_windows, gy_windows):
        """StructureMatrix for n gathered windows: f32 [n, elems] each -> f32 [n, 3] = (Ix2, IxIy, Iy2)."""
        gx = np.ascontiguousarray(gx_windows, dtype=np.float32)
        gy = np.ascontiguousarray(gy_windows, dtype=np.float32)
        assert gx.shape == gy.shape and gx.ndim == 2
        out = np.zeros((gx.shape[0], 3), np.float32)
        self._chk(lib().vslam_structure_matrix_windows(self._h, gx.ctypes.data, gy.ctypes.data, gx.shape[1], gx.shape[0], out.ctypes.data), "vslam_structure_matrix_windows")
        return out

    def pyramid(self, img, n_octaves: int = 4, sigma0: float = 1.6):
        return Pyramid(self, img, n_octaves, sigma0)

    # ---- device-resident batch (torch CUDA tensors)
    def detect_batch(self, params: Params, frames, **outs):
        """frames: uint8 CUDA tensor [n, rows, cols]; outs: CUDA tensors by BatchOut field name."""
        n, bo, fstride = self._batch_args(params, frames, outs)
        self._chk(lib().vslam_detect_batch_dev(self._h, C.byref(params), frames.data_ptr(), fstride, n, C.byref(bo)), "vslam_detect_batch_dev")

    def _batch_args(self, params: Params, frames, outs):
        import torch

        n = frames.shape[0]
        L = batch_layout(params)  # raises on bad parameters before anything is checked against them
        N = params.rows * params.cols

        def need(name, t, dtype):
            # sizes are checked by the library itself (vslam_batch_out carries them); placement, dtype and
            # contiguity are properties of the tensor that only this wrapper can see
            if not (t.is_cuda and t.device.index == self.device):
                raise ValueError(f"detect_batch: {name} is not on cuda:{self.device}")
            if t.dtype not in dtype:
                raise ValueError(f"detect_batch: {name} has dtype {t.dtype}, expected one of {dtype}")
            if not t.is_contiguous():
                raise ValueError(f"detect_batch: {name} is not contiguous")

        if frames.dim() != 3 or tuple(frames.shape[1:]) != (params.rows, params.cols) or frames.stride(2) != 1 or frames.stride(1) != params.cols:
            raise ValueError("detect_batch: frames must be [n, rows, cols] uint8 with dense rows")
        if not (frames.is_cuda and frames.device.index == self.device and frames.dtype == torch.uint8):
            raise ValueError(f"detect_batch: frames must be uint8 on cuda:{self.device}")
        if n > 1 and frames.stride(0) < N:
            raise ValueError("detect_batch: frames overlap")
        i32, u8, f32, i64 = (torch.int32,), (torch.uint8,), (torch.float32,), (torch.int64, torch.uint64)
        spec = {
            "response": f32, "nms_mask": u8, "nms2": f32, "harris_kps": i32 + f32, "harris_counts": i32,
            "pyramid": u8, "extrema_bits": i64, "dog_points": i32, "dog_counts": i32,
            "oriented_points": i32, "oriented_counts": i32, "oriented_survivors": i32,
            "descriptors": f32, "descriptor_defined": u8,
        }
        bo = BatchOut()
        bo.struct_size = C.sizeof(BatchOut)
        for k, t in outs.items():
            if t is not None:
                if k not in spec:
                    raise ValueError(f"detect_batch: unknown output {k!r}")
                need(k, t, spec[k])
                setattr(bo, k, t.data_ptr())
                setattr(bo, k + "_bytes", t.numel() * t.element_size())
        return n, bo, (frames.stride(0) if n > 1 else N)  # a size-1 dimension may carry any stride

    def side_stream_report(self):
        """vslam_ctx_side_stream_report: (index of the side-stream pair the batched path runs on: 0 = the first created,
        state of the comparison: 0 not started, 1 measuring, 2 decided)."""
        a, b = C.c_int(0), C.c_int(0)
        self._chk(lib().vslam_ctx_side_stream_report(self._h, C.byref(a), C.byref(b)), "vslam_ctx_side_stream_report")
        return a.value, b.value

    def set_side_stream_priority(self, low: bool):
        """vslam_ctx_set_side_stream_priority: low = True puts the two side streams at the device's lowest priority (yielding);
        the default is the context stream's priority.  Before the first batch call only."""
        self._chk(lib().vslam_ctx_set_side_stream_priority(self._h, int(bool(low))), "vslam_ctx_set_side_stream_priority")

    def join_watch_report(self):
        """vslam_ctx_join_watch_report: (level: 0 low-priority side streams / 1 flat priority / 2 no side streams, done,
        last measured fraction of a call the context's stream spent waiting for the side streams; -1: none yet)."""
        a, b, f = C.c_int(0), C.c_int(0), C.c_float(-1.0)
        self._chk(lib().vslam_ctx_join_watch_report(self._h, C.byref(a), C.byref(b), C.byref(f)), "vslam_ctx_join_watch_report")
        return a.value, bool(b.value), float(f.value)

    def set_matrix_path(self, on: bool):
        """vslam_ctx_set_matrix_path: OPT-IN matrix-core (MFMA) form of the LDS-tiled octave kernels; off by default."""
        self._chk(lib().vslam_ctx_set_matrix_path(self._h, 1 if on else 0), "vslam_ctx_set_matrix_path")

    def matrix_path(self) -> bool:
        return bool(lib().vslam_ctx_get_matrix_path(self._h))

    def set_f32_fused(self, on: bool):
        """vslam_ctx_set_f32_fused: the f32 stages with fused multiply-adds (an OpenCV that dispatches AVX2 + FMA3) instead of
        every product and sum rounded (its SSE2 baseline, the default); oracle.fma_variant is the checker's side of it."""
        self._chk(lib().vslam_ctx_set_f32_fused(self._h, 1 if on else 0), "vslam_ctx_set_f32_fused")

    def get_f32_fused(self) -> bool:
        return bool(lib().vslam_ctx_get_f32_fused(self._h))

    def set_join_watch(self, on: bool):
        """vslam_ctx_set_join_watch: False = this context takes no more join-lag measurements (stays at its level)."""
        self._chk(lib().vslam_ctx_set_join_watch(self._h, 1 if on else 0), "vslam_ctx_set_join_watch")

    def pin_side_streams(self, level: int):
        """vslam_ctx_pin_side_streams: 0 yielding / 1 the context stream's priority / 2 no side streams; before the first batch call."""
        self._chk(lib().vslam_ctx_pin_side_streams(self._h, int(level)), "vslam_ctx_pin_side_streams")

    def tune_side_streams(self, on: bool = True):
        """vslam_ctx_tune_side_streams: opt in to (or out of) the library's comparison of side-stream pairs; off by default."""
        self._chk(lib().vslam_ctx_tune_side_streams(self._h, 1 if on else 0), "vslam_ctx_tune_side_streams")

    def follow(self, leader: "Context"):
        """vslam_ctx_follow: this context's next work starts once `leader`'s latest batch is past its octave-0 kernels."""
        self._chk(lib().vslam_ctx_follow(self._h, leader._h), "vslam_ctx_follow")

    def detect_batch_host(self, params: Params, frames, harris_budget: int | None = None, dog_budget: int | None = None):
        """vslam_detect_batch_host: numpy uint8 frames [n, rows, cols] in, the packed lists out - no torch involved.
        Returns {"harris": (records, offsets, counts), "dog": (records, offsets, counts)}; a budget of 0 skips the list,
        None = room for every frame's full capacity."""
        fr = np.ascontiguousarray(frames, dtype=np.uint8)
        if fr.ndim != 3 or fr.shape[1:] != (params.rows, params.cols):
            raise ValueError("detect_batch_host: frames must be [n, rows, cols] uint8")
        n = fr.shape[0]
        hl = HostLists()
        hl.struct_size = C.sizeof(HostLists)
        res = {}
        keep = []
        for name, dtype, cap, budget in (("harris", KP_DTYPE, params.harris_cap, harris_budget), ("dog", POINT_DTYPE, params.dog_cap, dog_budget)):
            if budget == 0 or (name == "dog" and params.n_octaves == 0) or (name == "harris" and not params.do_harris):
                continue
            rec = np.zeros(n * cap if budget is None else budget, dtype)
            off = np.zeros(n + 1, np.uint64)
            cnt = np.zeros(n, np.uint32)
            setattr(hl, name, rec.ctypes.data)
            setattr(hl, name + "_bytes", rec.nbytes)
            setattr(hl, name + "_offsets", off.ctypes.data)
            setattr(hl, name + "_counts", cnt.ctypes.data)
            keep.append((name, rec, off, cnt))
        self._chk(lib().vslam_detect_batch_host(self._h, C.byref(params), fr.ctypes.data, params.rows * params.cols, n, C.byref(hl)), "vslam_detect_batch_host")
        for name, rec, off, cnt in keep:
            res[name] = (rec[: min(int(off[n]), len(rec))], off, cnt)
        return res

    def pack_lists(self, lists, counts, packed, offsets):
        """vslam_pack_lists_dev: lists [n, cap, k] (int32 / float32 records), counts [n] int32 -> packed (flat,
        any capacity), offsets [n + 1] int64.  CUDA tensors; asynchronous on the context stream."""
        n, cap = lists.shape[0], lists.shape[1]
        rb = lists[0, 0].numel() * lists.element_size()
        if not (lists.is_contiguous() and packed.is_contiguous() and counts.numel() >= n and offsets.numel() >= n + 1 and offsets.element_size() == 8):
            raise ValueError("pack_lists: bad tensors")
        self._chk(lib().vslam_pack_lists_dev(self._h, lists.data_ptr(), rb, cap, counts.data_ptr(), n, packed.data_ptr(),
                                             packed.numel() * packed.element_size(), offsets.data_ptr()), "vslam_pack_lists_dev")

    def pack_points16(self, lists, counts, packed, offsets):
        """vslam_pack_points16_dev: SLAM::point lists [n, cap, 6] int32, counts [n] int32 -> packed 16-byte records (any flat
        tensor; its byte size is the capacity), offsets [n + 1] int64.  CUDA tensors; asynchronous on the context stream."""
        n, cap = lists.shape[0], lists.shape[1]
        if not (lists.is_contiguous() and lists.shape[2] == 6 and lists.element_size() == 4 and packed.is_contiguous() and counts.numel() >= n and
                offsets.numel() >= n + 1 and offsets.element_size() == 8):
            raise ValueError("pack_points16: bad tensors")
        self._chk(lib().vslam_pack_points16_dev(self._h, lists.data_ptr(), cap, counts.data_ptr(), n, packed.data_ptr(),
                                                packed.numel() * packed.element_size(), offsets.data_ptr()), "vslam_pack_points16_dev")

    def count_totals(self, harris_counts, dog_counts, totals):
        """vslam_count_totals_dev: int32 [n] count tensors (either may be None) -> totals int64 [2], on the context stream."""
        n = (harris_counts if harris_counts is not None else dog_counts).numel()
        self._chk(lib().vslam_count_totals_dev(self._h, harris_counts.data_ptr() if harris_counts is not None else None,
                                               dog_counts.data_ptr() if dog_counts is not None else None, n, totals.data_ptr()), "vslam_count_totals_dev")

    def kernel_timing_enable(self, name: str | None):
        self._chk(lib().vslam_kernel_timing_enable(self._h, name.encode() if name else None), "vslam_kernel_timing_enable")

    def kernel_timing_read(self):
        n, ms = C.c_int(), C.c_double()
        self._chk(lib().vslam_kernel_timing_read(self._h, C.byref(n), C.byref(ms)), "vslam_kernel_timing_read")
        return n.value, ms.value


class Pyramid:
    """vslam_pyramid: the Gaussian / DoG stacks of one image, resident in HBM."""

    def __init__(self, ctx: Context, img, n_octaves: int = 4, sigma0: float = 1.6):
        img = _u8(img)
        self.ctx = ctx
        h = C.c_void_p()
        ctx._chk(lib().vslam_pyramid_build_u8(ctx._h, img.ctypes.data, *img.shape, img.strides[0], n_octaves, float(sigma0), C.byref(h)), "vslam_pyramid_build_u8")
        self._h = h
        info = PyramidInfo()
        lib().vslam_pyramid_get_info(h, C.byref(info))
        self.n_octaves = info.n_octaves
        self.sizes = [(info.rows[o], info.cols[o]) for o in range(info.n_octaves)]
        self.sigmas = [[info.sigma[o][l] for l in range(NUM_LEVELS)] for o in range(info.n_octaves)]
        self.ksizes = [[info.ksize[o][l] for l in range(NUM_LEVELS)] for o in range(info.n_octaves)]

    def _get(self, fn, what, o, *lvl):
        if not 0 <= o < self.n_octaves:
            out = np.empty((1, 1), np.uint8)
        else:
            out = np.empty(self.sizes[o], np.uint8)
        self.ctx._chk(fn(self._h, o, *lvl, out.ctypes.data, out.strides[0]), what)
        return out

    def base(self, o):
        return self._get(lib().vslam_pyramid_get_base, "vslam_pyramid_get_base", o)

    def gauss(self, o, l):
        return self._get(lib().vslam_pyramid_get_gauss, "vslam_pyramid_get_gauss", o, l)

    def dog(self, o, l):
        return self._get(lib().vslam_pyramid_get_dog, "vslam_pyramid_get_dog", o, l)

    def gradients(self, o, l):
        """(grad_x, grad_y, magnitude, orientation[deg]) of Gaussian level (o, l), computed on demand."""
        shape = self.sizes[o] if 0 <= o < self.n_octaves else (1, 1)
        outs = [np.empty(shape, np.float32) for _ in range(4)]
        self.ctx._chk(lib().vslam_pyramid_get_gradients(self._h, o, l, *[a.ctypes.data for a in outs], outs[0].strides[0]), "vslam_pyramid_get_gradients")
        return tuple(outs)

    def extrema(self, octave: int, window: int = 3, min_contrast: int = 8, cap: int = 1 << 22):
        """(mask[3, lat_rows, lat_cols] u8 unpacked from the bitmask, points, total count)."""
        lr, lc = extrema_lattice(*self.sizes[octave], window) if 0 <= octave < self.n_octaves else (0, 0)
        wpr = (lc + 63) // 64
        bits = np.zeros((3, lr, max(wpr, 1)), np.uint64)
        pts = np.zeros(cap, POINT_DTYPE)
        n = C.c_size_t()
        self.ctx._chk(lib().vslam_dog_extrema(self.ctx._h, self._h, octave, window, min_contrast, bits.ctypes.data, pts.ctypes.data, cap, C.byref(n)), "vslam_dog_extrema")
        mask = np.unpackbits(bits.view(np.uint8), axis=-1, bitorder="little")[..., :lc] if lc else np.zeros((3, lr, 0), np.uint8)
        return mask, pts[: min(n.value, cap)], n.value

    def extrema_dense(self, octave: int, min_contrast: int = 8, cap: int = 1 << 22):
        """Extension: dense 3x3x3 test on every pixel; (mask[3, rows, cols] u8, points, total count)."""
        r, c = self.sizes[octave] if 0 <= octave < self.n_octaves else (0, 0)
        wpr = (c + 63) // 64
        bits = np.zeros((3, r, max(wpr, 1)), np.uint64)
        pts = np.zeros(cap, POINT_DTYPE)
        n = C.c_size_t()
        self.ctx._chk(lib().vslam_dog_extrema_dense(self.ctx._h, self._h, octave, min_contrast, bits.ctypes.data, pts.ctypes.data, cap, C.byref(n)), "vslam_dog_extrema_dense")
        mask = np.unpackbits(bits.view(np.uint8), axis=-1, bitorder="little")[..., :c] if c else np.zeros((3, r, 0), np.uint8)
        return mask, pts[: min(n.value, cap)], n.value

    def keypoints(self, octave: int, window: int = 3, cap: int = 1 << 22):
        """initialKeypointDetection incl. FeaturePointLocalization: (points, total count)."""
        pts = np.zeros(cap, POINT_DTYPE)
        n = C.c_size_t()
        self.ctx._chk(lib().vslam_dog_keypoints(self.ctx._h, self._h, octave, window, pts.ctypes.data, cap, C.byref(n)), "vslam_dog_keypoints")
        return pts[: min(n.value, cap)], n.value

    def filter_keypoints(self, octave: int, kps, cap: int = 1 << 22):
        """filterKeypoints for one octave: (oriented points, total count)."""
        kps = np.ascontiguousarray(kps, dtype=POINT_DTYPE)
        out = np.zeros(cap, POINT_DTYPE)
        n = C.c_size_t()
        self.ctx._chk(lib().vslam_filter_keypoints(self.ctx._h, self._h, octave, kps.ctypes.data, len(kps), out.ctypes.data, cap, C.byref(n)), "vslam_filter_keypoints")
        return out[: min(n.value, cap)], n.value

    def sift_descriptors(self, octave: int, oriented):
        """SIFT() for one octave's oriented keypoints: (desc f32 [n, 128], defined bool [n])."""
        kps = np.ascontiguousarray(oriented, dtype=POINT_DTYPE)
        desc = np.zeros((len(kps), 128), np.float32)
        ok = np.zeros(max(len(kps), 1), np.uint8)
        self.ctx._chk(lib().vslam_sift_descriptors(self.ctx._h, self._h, octave, kps.ctypes.data, len(kps), desc.ctypes.data, ok.ctypes.data), "vslam_sift_descriptors")
        return desc, ok[: len(kps)].astype(bool)

    def close(self):
        if getattr(self, "_h", None):
            lib().vslam_pyramid_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
