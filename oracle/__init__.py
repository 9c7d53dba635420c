"""ctypes front end of the CPU oracle (oracle/vslam_oracle.c).

TEST INFRASTRUCTURE ONLY -- parity unpinned (see vslam_oracle.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product package ``visualslam_amd`` never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvslam_oracle.so")

NUM_LEVELS = 6
NUM_DOGS = 5
MAX_OCTAVES = 16


def build(force: bool = False) -> str:
    """Compile the C restatement with gcc (Makefile next to this file)."""
    src = os.path.join(_HERE, "vslam_oracle.c")
    hdr = os.path.join(_HERE, "vslam_oracle.h")
    stale = (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B", "libvslam_oracle.so"], check=True, capture_output=True)
    return _LIB_PATH


class _Point(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("row", "col", "value", "padding", "octave", "level")]


class _Kp(C.Structure):
    _fields_ = [("row", C.c_int32), ("col", C.c_int32), ("response", C.c_float)]


class _Pyr(C.Structure):
    _fields_ = [
        ("n_octaves", C.c_int),
        ("sigma0", C.c_double),
        ("rows", C.c_int * MAX_OCTAVES),
        ("cols", C.c_int * MAX_OCTAVES),
        ("sigma", (C.c_double * NUM_LEVELS) * MAX_OCTAVES),
        ("ksize", (C.c_int * NUM_LEVELS) * MAX_OCTAVES),
        ("base", C.POINTER(C.c_uint8) * MAX_OCTAVES),
        ("gauss", (C.POINTER(C.c_uint8) * NUM_LEVELS) * MAX_OCTAVES),
        ("dog", (C.POINTER(C.c_uint8) * NUM_DOGS) * MAX_OCTAVES),
    ]


POINT_DTYPE = np.dtype(
    [(n, "<i4") for n in ("row", "col", "value", "padding", "octave", "level")], align=True
)
KP_DTYPE = np.dtype([("row", "<i4"), ("col", "<i4"), ("response", "<f4")], align=True)

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.vo_reflect101.restype = C.c_int
        L.vo_gauss_ksize_u8.argtypes = [C.c_double]
        L.vo_gauss_taps_q8.argtypes = [C.c_int, C.c_double, C.c_void_p]
        L.vo_gaussian_blur_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_double, C.c_void_p, C.c_size_t]
        L.vo_sobel_k1_u8_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.vo_resize_linear2x_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
        L.vo_half_size.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vo_half_size.restype = None
        L.vo_resize_nearest_half_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
        L.vo_convert_scale_abs_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
        L.vo_harris_from_grad_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_float, C.c_int, C.c_void_p, C.c_size_t]
        L.vo_harris_response_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_float, C.c_int, C.c_void_p, C.c_size_t]
        L.vo_nms_strict_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        L.vo_nms_strict_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        L.vo_nms2_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_float)]
        L.vo_harris_keypoints.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t]
        L.vo_harris_keypoints.restype = C.c_size_t
        L.vo_auto_num_octaves.argtypes = [C.c_int, C.c_int]
        L.vo_sigma.argtypes = [C.c_double, C.c_int, C.c_int]
        L.vo_sigma.restype = C.c_double
        L.vo_pyramid_build_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_double]
        L.vo_pyramid_build_u8.restype = C.POINTER(_Pyr)
        L.vo_pyramid_free.argtypes = [C.POINTER(_Pyr)]
        L.vo_pyramid_free.restype = None
        L.vo_fast_atan2_deg.argtypes = [C.c_float, C.c_float]
        L.vo_fast_atan2_deg.restype = C.c_float
        L.vo_level_gradients.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.vo_extrema_lattice.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vo_extrema_lattice.restype = None
        L.vo_dog_extrema.argtypes = [C.POINTER(_Pyr), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.vo_dog_extrema.restype = C.c_size_t
        L.vo_dog_extrema_dense.argtypes = [C.POINTER(_Pyr), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.vo_dog_extrema_dense.restype = C.c_size_t
        L.vo_feature_point_localization.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.vo_dog_keypoints.argtypes = [C.POINTER(_Pyr), C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.vo_dog_keypoints.restype = C.c_size_t
        L.vo_gauss_ksize_f32.argtypes = [C.c_double]
        L.vo_gauss_kernel_f32.argtypes = [C.c_int, C.c_double, C.c_void_p]
        L.vo_compute_edge_response.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int]
        L.vo_compute_edge_response.restype = C.c_float
        L.vo_filter_keypoints.argtypes = [C.POINTER(_Pyr), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.vo_filter_keypoints.restype = C.c_size_t
        L.vo_cos_sin_deg.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.vo_cos_sin_deg.restype = None
        L.vo_rotated_window_points.argtypes = [C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.vo_sift_descriptors.argtypes = [C.POINTER(_Pyr), C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.vo_sift_descriptors.restype = C.c_size_t
        L.vo_baseline_frames.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_ulonglong)]
        L.vo_set_fma_variant.argtypes = [C.c_int]
        L.vo_set_fma_variant.restype = None
        L.vo_get_fma_variant.restype = C.c_int
        _lib = L
    return _lib


def baseline_frames(frames, n_octaves: int = 4, threads: int = 1) -> int:
    """The whole hot path (Harris + NMS + keypoints + pyramid + extrema) on frames [n, rows, cols]
    with `threads` OpenMP threads over frames; returns the keypoint total.  bench.py times this."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    assert frames.ndim == 3
    kp = C.c_ulonglong(0)
    _chk(lib().vo_baseline_frames(frames.ctypes.data, frames.shape[0], frames.shape[1], frames.shape[2], n_octaves, threads, C.byref(kp)), "baseline_frames")
    return kp.value


def cos_sin_deg(theta_deg):
    c, s = C.c_float(), C.c_float()
    lib().vo_cos_sin_deg(float(theta_deg), C.byref(c), C.byref(s))
    return np.float32(c.value), np.float32(s.value)


def rotated_window_points(cx, cy, window, theta_deg):
    """Rotation::getRotatedWindowPoints: int32 [(window+1)^2, 2] = (x, y), rows outer."""
    xy = np.zeros(((window + 1) ** 2, 2), np.int32)
    _chk(lib().vo_rotated_window_points(int(cx), int(cy), int(window), float(theta_deg), xy.ctypes.data), "rotated_window_points")
    return xy


def gauss_ksize_f32(sigma):
    return lib().vo_gauss_ksize_f32(float(sigma))


def gauss_kernel_f32(n, sigma):
    k = np.zeros(n, np.float32)
    if lib().vo_gauss_kernel_f32(n, float(sigma), k.ctypes.data) != 0:
        raise ValueError("bad kernel size")
    return k


def blur_f32_roi(parent, x0, y0, w, h, sigma):
    """cv::GaussianBlur on the ROI Rect(x0, y0, w, h) of a CV_32F parent (the filter reads the parent around the window)."""
    parent = _f32(parent)
    out = np.zeros((h, w), np.float32)
    L = lib()
    L.vo_blur_f32_roi.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p]
    if L.vo_blur_f32_roi(parent.ctypes.data, parent.shape[0], parent.shape[1], int(x0), int(y0), int(w), int(h), float(sigma), out.ctypes.data) != 0:
        raise ValueError("blur_f32_roi: bad arguments")
    return out


def compute_edge_response(gx, gy, row, col, padding=1):
    gx, gy = _f32(gx), _f32(gy)
    return float(lib().vo_compute_edge_response(gx.ctypes.data, gy.ctypes.data, gx.shape[0], gx.shape[1], gx.shape[1], row, col, padding))


def feature_point_localization(d_x, d_y, d_scale, value):
    """(kept, new_value) of FeaturePointLocalization for one candidate."""
    nv = C.c_int(0)
    k = lib().vo_feature_point_localization(int(d_x), int(d_y), int(d_scale), int(value), C.byref(nv))
    return bool(k), nv.value


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2
    return a


def _chk(rc, what):
    if rc != 0:
        raise ValueError(f"oracle {what} rejected its arguments (rc={rc})")


def reflect101(p: int, n: int) -> int:
    return lib().vo_reflect101(int(p), int(n))


def gauss_ksize_u8(sigma: float) -> int:
    return lib().vo_gauss_ksize_u8(float(sigma))


def gauss_taps_q8(n: int, sigma: float) -> np.ndarray:
    t = np.zeros(n, np.uint16)
    _chk(lib().vo_gauss_taps_q8(n, float(sigma), t.ctypes.data), "gauss_taps_q8")
    return t


def gaussian_blur_u8(img, ksize: int, sigma: float) -> np.ndarray:
    img = _u8(img)
    out = np.empty_like(img)
    _chk(lib().vo_gaussian_blur_u8(img.ctypes.data, *img.shape, img.strides[0], ksize, float(sigma), out.ctypes.data, out.strides[0]), "gaussian_blur_u8")
    return out


def sobel_k1(img, dx: int, dy: int) -> np.ndarray:
    img = _u8(img)
    out = np.empty(img.shape, np.float32)
    _chk(lib().vo_sobel_k1_u8_f32(img.ctypes.data, *img.shape, img.strides[0], dx, dy, out.ctypes.data, out.strides[0]), "sobel_k1")
    return out


def resize_linear2x(img) -> np.ndarray:
    img = _u8(img)
    out = np.empty((img.shape[0] * 2, img.shape[1] * 2), np.uint8)
    _chk(lib().vo_resize_linear2x_u8(img.ctypes.data, *img.shape, img.strides[0], out.ctypes.data, out.strides[0]), "resize_linear2x")
    return out


def half_size(rows: int, cols: int):
    r, c = C.c_int(), C.c_int()
    lib().vo_half_size(rows, cols, C.byref(r), C.byref(c))
    return r.value, c.value


def resize_nearest_half(img) -> np.ndarray:
    img = _u8(img)
    out = np.empty(half_size(*img.shape), np.uint8)
    _chk(lib().vo_resize_nearest_half_u8(img.ctypes.data, *img.shape, img.strides[0], out.ctypes.data, out.strides[0]), "resize_nearest_half")
    return out


def convert_scale_abs(x) -> np.ndarray:
    x = _f32(x)
    out = np.empty(x.shape, np.uint8)
    _chk(lib().vo_convert_scale_abs_f32(x.ctypes.data, *x.shape, x.strides[0], out.ctypes.data, out.strides[0]), "convert_scale_abs")
    return out


def harris_from_grad(ix, iy, k: float = 0.04, window: int = 3) -> np.ndarray:
    ix, iy = _f32(ix), _f32(iy)
    assert ix.shape == iy.shape
    out = np.empty(ix.shape, np.float32)
    _chk(lib().vo_harris_from_grad_f32(ix.ctypes.data, iy.ctypes.data, *ix.shape, ix.strides[0], k, window, out.ctypes.data, out.strides[0]), "harris_from_grad")
    return out


def harris_response(img, k: float = 0.04, window: int = 3) -> np.ndarray:
    img = _u8(img)
    out = np.empty(img.shape, np.float32)
    _chk(lib().vo_harris_response_u8(img.ctypes.data, *img.shape, img.strides[0], k, window, out.ctypes.data, out.strides[0]), "harris_response")
    return out


def nms_strict(x, window: int = 3) -> np.ndarray:
    x = np.asarray(x)
    out = np.empty(x.shape, np.uint8)
    if x.dtype == np.uint8:
        x = _u8(x)
        rc = lib().vo_nms_strict_u8(x.ctypes.data, *x.shape, x.strides[0], window, out.ctypes.data, out.strides[0])
    else:
        x = _f32(x)
        rc = lib().vo_nms_strict_f32(x.ctypes.data, *x.shape, x.strides[0], window, out.ctypes.data, out.strides[0])
    _chk(rc, "nms_strict")
    return out


def nms2(resp, window: int = 5):
    resp = _f32(resp)
    out = np.empty(resp.shape, np.float32)
    tm = C.c_float()
    _chk(lib().vo_nms2_f32(resp.ctypes.data, *resp.shape, resp.strides[0], window, out.ctypes.data, out.strides[0], C.byref(tm)), "nms2")
    return out, tm.value


def harris_keypoints(nms2_map) -> np.ndarray:
    m = _f32(nms2_map)
    n = lib().vo_harris_keypoints(m.ctypes.data, *m.shape, m.strides[0], None, 0)
    out = np.zeros(n, KP_DTYPE)
    if n:
        lib().vo_harris_keypoints(m.ctypes.data, *m.shape, m.strides[0], out.ctypes.data, n)
    return out


class fma_variant:
    """`with oracle.fma_variant(True):` runs the f32 stages (fastAtan2's polynomial, the separable f32 filter's two
    passes) with fused multiply-adds, as OpenCV's AVX2 + FMA3 dispatch would; the default - and what the GPU kernels
    compute - rounds every multiply and add (vslam_oracle.c: g_fma_variant).  `on`: True / False, or a mask (ATAN = 1,
    FILTER = 2: the two OpenCV modules dispatch independently).  Restores the previous setting."""

    ATAN, FILTER, BOTH = 1, 2, 3  # the mask's bits: fastAtan32f's polynomial; the f32 filter's row / column passes

    def __init__(self, on):
        self.mask = 3 if on is True else int(on) & 3

    def __enter__(self):
        self.prev = lib().vo_get_fma_variant()
        lib().vo_set_fma_variant(self.mask)
        return self

    def __exit__(self, *exc):
        lib().vo_set_fma_variant(self.prev)
        return False


def fast_atan2_deg(y: float, x: float) -> float:
    return lib().vo_fast_atan2_deg(float(y), float(x))


def level_gradients(g):
    """(gx, gy, mag, orient) f32 images of one Gaussian level (processGradients)."""
    g = _u8(g)
    outs = [np.empty(g.shape, np.float32) for _ in range(4)]
    _chk(lib().vo_level_gradients(g.ctypes.data, *g.shape, g.strides[0], *[o.ctypes.data for o in outs], outs[0].strides[0]), "level_gradients")
    return tuple(outs)


def auto_num_octaves(rows: int, cols: int) -> int:
    return lib().vo_auto_num_octaves(rows, cols)


def sigma_at(sigma0: float, octave: int, level: int) -> float:
    return lib().vo_sigma(float(sigma0), octave, level)


def extrema_lattice(rows: int, cols: int, window: int = 3):
    r, c = C.c_int(), C.c_int()
    lib().vo_extrema_lattice(rows, cols, window, C.byref(r), C.byref(c))
    return r.value, c.value


class Pyramid:
    """Owns a vo_pyramid; exposes numpy copies of its images."""

    def __init__(self, img, n_octaves: int = 4, sigma0: float = 1.6):
        img = _u8(img)
        self._p = lib().vo_pyramid_build_u8(img.ctypes.data, *img.shape, img.strides[0], n_octaves, float(sigma0))
        if not self._p:
            raise ValueError("oracle pyramid_build rejected its arguments")
        s = self._p.contents
        self.n_octaves = s.n_octaves
        self.sizes = [(s.rows[o], s.cols[o]) for o in range(self.n_octaves)]
        self.sigmas = [[s.sigma[o][l] for l in range(NUM_LEVELS)] for o in range(self.n_octaves)]
        self.ksizes = [[s.ksize[o][l] for l in range(NUM_LEVELS)] for o in range(self.n_octaves)]

    def _img(self, ptr, o):
        r, c = self.sizes[o]
        return np.ctypeslib.as_array(ptr, shape=(r, c)).copy()

    def base(self, o):
        return self._img(self._p.contents.base[o], o)

    def gauss(self, o, l):
        return self._img(self._p.contents.gauss[o][l], o)

    def dog(self, o, l):
        return self._img(self._p.contents.dog[o][l], o)

    def extrema(self, octave: int, window: int = 3, min_contrast: int = 8):
        """(mask[3, lat_rows, lat_cols] u8, points[POINT_DTYPE]) for one octave."""
        lr, lc = extrema_lattice(*self.sizes[octave], window)
        mask = np.zeros((3, lr, lc), np.uint8)
        n = lib().vo_dog_extrema(self._p, octave, window, min_contrast, mask.ctypes.data, None, 0)
        pts = np.zeros(n, POINT_DTYPE)
        if n:
            lib().vo_dog_extrema(self._p, octave, window, min_contrast, None, pts.ctypes.data, n)
        return mask, pts

    def extrema_dense(self, octave: int, min_contrast: int = 8):
        """Extension: dense 3x3x3 test; (mask[3, rows, cols] u8, points[POINT_DTYPE])."""
        r, c = self.sizes[octave]
        mask = np.zeros((3, r, c), np.uint8)
        n = lib().vo_dog_extrema_dense(self._p, octave, min_contrast, mask.ctypes.data, None, 0)
        pts = np.zeros(n, POINT_DTYPE)
        if n:
            lib().vo_dog_extrema_dense(self._p, octave, min_contrast, None, pts.ctypes.data, n)
        return mask, pts

    def keypoints(self, octave, window=3):
        """initialKeypointDetection incl. FeaturePointLocalization: points[POINT_DTYPE]."""
        n = lib().vo_dog_keypoints(self._p, octave, window, None, 0)
        pts = np.zeros(n, POINT_DTYPE)
        if n:
            lib().vo_dog_keypoints(self._p, octave, window, pts.ctypes.data, n)
        return pts

    def filter_keypoints(self, octave, kps):
        """filterKeypoints for one octave: oriented keypoints[POINT_DTYPE] (value = angle in degrees)."""
        kps = np.ascontiguousarray(kps, dtype=POINT_DTYPE)
        n = lib().vo_filter_keypoints(self._p, octave, kps.ctypes.data, len(kps), None, 0)
        if n == C.c_size_t(-1).value:
            raise ValueError("filterKeypoints: keypoint outside the pyramid data")
        out = np.zeros(n, POINT_DTYPE)
        if n:
            lib().vo_filter_keypoints(self._p, octave, kps.ctypes.data, len(kps), out.ctypes.data, n)
        return out

    def sift_descriptors(self, octave, oriented):
        """SIFT() for one octave's oriented keypoints: (desc f32 [n, 128], defined bool [n])."""
        kps = np.ascontiguousarray(oriented, dtype=POINT_DTYPE)
        desc = np.zeros((len(kps), 128), np.float32)
        ok = np.zeros(len(kps), np.uint8)
        r = lib().vo_sift_descriptors(self._p, octave, kps.ctypes.data, len(kps), desc.ctypes.data, ok.ctypes.data)
        if r == C.c_size_t(-1).value:
            raise ValueError("SIFT: keypoint outside the pyramid data")
        return desc, ok.astype(bool)

    def close(self):
        if self._p:
            lib().vo_pyramid_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
