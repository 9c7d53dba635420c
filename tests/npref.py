"""Independent numpy/scipy restatements used to cross-check the C oracle.

These are written from SURVEY.md Appendix A in *integer* form (exact sums), not from
oracle/vslam_oracle.c, so that an indexing or rounding slip in either shows up as a
disagreement.  They are slow and only used on small inputs.
"""
import numpy as np
from scipy import ndimage


def blur_q8(img: np.ndarray, taps: np.ndarray) -> np.ndarray:
    """Exact 2-D integer correlation with 8.8 taps, reflect-101, (acc+32768)>>16."""
    t = taps.astype(np.int64)
    a = img.astype(np.int64)
    h = ndimage.correlate1d(a, t, axis=1, mode="mirror")
    v = ndimage.correlate1d(h, t, axis=0, mode="mirror")
    return ((v + 32768) >> 16).astype(np.uint8)


def harris_int(img: np.ndarray, k=np.float32(0.04)) -> np.ndarray:
    """Harris response for window 3 with all sums in exact integers (SURVEY section 7 recipe)."""
    a = img.astype(np.int64)
    p = np.pad(a, 1, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    s = (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:]
         + 2 * p[1:-1, :-2] + 4 * p[1:-1, 1:-1] + 2 * p[1:-1, 2:]
         + p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:])
    b = (s + 8) >> 4
    bp = np.pad(b, 1, mode="reflect")
    ix = bp[1:-1, 2:] - bp[1:-1, :-2]
    iy = bp[2:, 1:-1] - bp[:-2, 1:-1]

    def box(x):
        q = np.pad(x, 1, mode="edge")  # BORDER_REPLICATE
        h, w = x.shape
        return sum(q[i:i + h, j:j + w] for i in range(3) for j in range(3))

    sxx, syy, sxy = box(ix * ix), box(iy * iy), box(ix * iy)
    det = (sxx * syy - sxy * sxy).astype(np.float32)  # int64 -> f32, round to nearest even
    tr = (sxx + syy).astype(np.float32)
    r = det - np.float32(k) * (tr * tr)
    return np.where(r > 0, r, np.float32(0)).astype(np.float32)


def resize2x(img: np.ndarray) -> np.ndarray:
    """INTER_LINEAR x2 with the 11-bit fixed-point steps of SURVEY A4."""
    h, w = img.shape
    a = img.astype(np.int64)

    def coeff(n_dst, n_src):
        d = np.arange(n_dst)
        f = (d + 0.5) * 0.5 - 0.5
        s = np.floor(f).astype(np.int64)
        fr = f - s
        return s, fr

    sx, fx = coeff(2 * w, w)
    lo = sx < 0
    fx = np.where(lo, 0.0, fx)
    sx = np.where(lo, 0, sx)
    hi = sx >= w - 1
    fx = np.where(hi, 0.0, fx)
    sx = np.where(hi, w - 1, sx)
    a0 = np.rint((1 - fx) * 2048).astype(np.int64)
    a1 = np.rint(fx * 2048).astype(np.int64)
    hrow = a[:, sx] * a0 + a[:, np.minimum(sx + 1, w - 1)] * a1
    sy, fy = coeff(2 * h, h)
    b0 = np.rint((1 - fy) * 2048).astype(np.int64)[:, None]
    b1 = np.rint(fy * 2048).astype(np.int64)[:, None]
    r0 = np.clip(sy, 0, h - 1)
    r1 = np.clip(sy + 1, 0, h - 1)
    v = (((b0 * (hrow[r0] >> 4)) >> 16) + ((b1 * (hrow[r1] >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def extrema_mask(dogs, window=3):
    """initialKeypointDetection candidate mask for levels 1..3, vectorised."""
    pad = (window - 1) // 2
    rows, cols = dogs[0].shape
    P = [np.pad(d.astype(np.int64), pad, mode="edge") for d in dogs]
    ii = np.arange(pad, rows, window)
    jj = np.arange(pad, cols, window)
    out = []
    for level in (1, 2, 3):
        vals = []
        for l in (level - 1, level, level + 1):
            for du in range(-pad, pad):
                for dv in range(-pad, pad):
                    vals.append(P[l][np.ix_(ii + du, jj + dv)])
        vals = np.stack(vals)
        this = P[level][np.ix_(ii, jj)]
        out.append(((this == vals.min(0)) | (this == vals.max(0))).astype(np.uint8))
    return np.stack(out), ii, jj


def localize(d_x, d_y, d_s, value):
    """FeaturePointLocalization (Diff_of_Gauss.cpp:223-251) written with numpy matrices instead
    of the oracle's unrolled scalars: (keep, new_value).  Same rounding points (SURVEY B-6)."""
    f32, f64 = np.float32, np.float64
    a = np.array([d_x, d_y, d_s], f32) / f32(255.0)
    B = np.outer(a, a).astype(f32)  # f32 * f32 rounded once
    Bd = B.astype(f64)

    def cof(r0, c0, r1, c1, r2, c2, r3, c3):
        return Bd[r0, c0] * Bd[r1, c1] - Bd[r2, c2] * Bd[r3, c3]

    det = Bd[0, 0] * cof(1, 1, 2, 2, 1, 2, 2, 1) - Bd[0, 1] * cof(1, 0, 2, 2, 1, 2, 2, 0) + Bd[0, 2] * cof(1, 0, 2, 1, 1, 1, 2, 0)
    inv = np.zeros((3, 3), f32)
    if det != 0.0:
        with np.errstate(all="ignore"):
            d = f64(1.0) / det
            # adjugate: inv[i][j] = cofactor(j, i) / det, expanded exactly as cv::invert writes it
            idx = {
                (0, 0): (1, 1, 2, 2, 1, 2, 2, 1), (0, 1): (0, 2, 2, 1, 0, 1, 2, 2), (0, 2): (0, 1, 1, 2, 0, 2, 1, 1),
                (1, 0): (1, 2, 2, 0, 1, 0, 2, 2), (1, 1): (0, 0, 2, 2, 0, 2, 2, 0), (1, 2): (0, 2, 1, 0, 0, 0, 1, 2),
                (2, 0): (1, 0, 2, 1, 1, 1, 2, 0), (2, 1): (0, 1, 2, 0, 0, 0, 2, 1), (2, 2): (0, 0, 1, 1, 0, 1, 1, 0),
            }
            for (i, j), t in idx.items():
                inv[i, j] = f32(cof(*t) * d)
    with np.errstate(all="ignore"):
        ninv = -inv
        z = np.zeros(3, f32)
        for i in range(3):
            t = f32(ninv[i, 0] * a[0])
            t = f32(t + f32(ninv[i, 1] * a[1]))
            t = f32(t + f32(ninv[i, 2] * a[2]))
            z[i] = t
        s = f64(0.0)
        for k in range(3):
            s = s + f64(z[k]) * f64(a[k])
        zhat = f32(f32(value) / f32(255.0) + f32(s * 0.5))
        if not zhat > f32(0.03):
            return False, value
        x = f32(zhat * f32(255.0))
    if not (x > -2147483904.0 and x < 2147483648.0):
        return True, -(2 ** 31)
    return True, int(x)


def filter_keypoints(gauss_level, kps_level, sigma_blur, kernel_f32):
    """filterKeypoints (Diff_of_Gauss.cpp:301-372) for the keypoints of ONE Gaussian level, written
    with whole-array numpy operations: gradients -> padded magnitude/orientation -> per-keypoint
    window blur taken from np.pad(reflect) of the padded parent -> histogram -> peaks.
    kps_level: iterable of (row, col, padding); returns [(row, col, angle)]."""
    import oracle  # gradients only (tested on their own); the filtering below is independent

    f32 = np.float32
    gx, gy, mag, ori = oracle.level_gradients(gauss_level)
    pmag = np.pad(mag, 8, mode="edge")
    pori = np.pad(ori, 8, mode="edge")
    k = np.asarray(kernel_f32, f32)
    n = len(k)
    R = n // 2
    out = []
    for (y, x, p) in kps_level:
        ru = np.clip(np.arange(y - p, y + p), 0, gx.shape[0] - 1)  # out-of-image reads clamped, like the oracle
        cv = np.clip(np.arange(x - p, x + p), 0, gx.shape[1] - 1)
        win_x = gx[np.ix_(ru, cv)]
        win_y = gy[np.ix_(ru, cv)]
        ix2 = iy2 = ixy = f32(0)
        for a, b in zip(win_x.ravel(), win_y.ravel()):
            ix2 = f32(ix2 + f32(a * a))
            iy2 = f32(iy2 + f32(b * b))
            ixy = f32(ixy + f32(a * b))
        with np.errstate(all="ignore"):
            det = f32(np.float64(ix2) * np.float64(iy2) - np.float64(ixy) * np.float64(ixy))
            tr = f32(np.float64(ix2) + np.float64(iy2))
            resp = f32(f32(tr * tr) / det)
        if not resp < f32(12.1):
            continue
        # parent rows/cols the window's blur touches, reflect-101 at the parent's edges
        ext = pmag
        while ext.shape[0] < pmag.shape[0] + 2 * R or ext.shape[1] < pmag.shape[1] + 2 * R:
            pr = min(R - (ext.shape[0] - pmag.shape[0]) // 2, ext.shape[0] - 1)
            pc = min(R - (ext.shape[1] - pmag.shape[1]) // 2, ext.shape[1] - 1)
            ext = np.pad(ext, ((max(pr, 0),) * 2, (max(pc, 0),) * 2), mode="reflect")
        oy = (ext.shape[0] - pmag.shape[0]) // 2
        ox = (ext.shape[1] - pmag.shape[1]) // 2
        strip = ext[oy + y - R: oy + y + 16 + R, ox + x - R: ox + x + 16 + R]
        rowf = (k[0] * strip[:, 0:16]).astype(f32)
        for i in range(1, n):
            rowf = (rowf + (k[i] * strip[:, i:i + 16]).astype(f32)).astype(f32)
        colf = (k[R] * rowf[R:R + 16]).astype(f32)
        for i in range(1, R + 1):
            colf = (colf + (k[R + i] * (rowf[R + i:R + i + 16] + rowf[R - i:R - i + 16]).astype(f32)).astype(f32)).astype(f32)
        histo = np.zeros(36, f32)
        wo = pori[y:y + 16, x:x + 16]
        for i in range(16):
            for j in range(16):
                histo[int(f32(wo[i, j] * f32(f32(36) / f32(360))))] += colf[i, j]
        thr = f32(histo.max() * f32(0.8))
        out += [(y, x, 10 * b) for b in range(36) if histo[b] > thr]
    return out


def rotated_window_points(cx, cy, window, theta_deg):
    """Rotation::getRotatedWindowPoints (rotation.cpp:112-130) with numpy float32 arithmetic:
    int32 [(window+1)^2, 2] = (x, y), rows outer."""
    f32 = np.float32
    ang = f32(np.float64(f32(theta_deg)) * (np.float64(3.1415926535897932384626433832795) / np.float64(f32(180.0))))
    c, s = f32(np.cos(np.float64(ang))), f32(np.sin(np.float64(ang)))
    pad = window // 2
    jj, ii = np.meshgrid(np.arange(cx - pad, cx + pad + 1), np.arange(cy - pad, cy + pad + 1))
    rx, ry = (jj - cx).astype(f32), (ii - cy).astype(f32)
    xr = np.trunc((rx * c).astype(f32) - (ry * s).astype(f32)).astype(np.int32) + cx
    yr = np.trunc((rx * s).astype(f32) + (ry * c).astype(f32)).astype(np.int32) + cy
    return np.stack([xr.ravel(), yr.ravel()], axis=1).astype(np.int32)


def sift_descriptor(gauss_level, kp_row, kp_col, angle_deg, kernel_f32):
    """SIFT() (Diff_of_Gauss.cpp:561-693) for ONE oriented keypoint, whole-array numpy: returns the 128
    floats, or None when the rotated window leaves the padded level's buffer."""
    import oracle  # gradients only (tested on their own)

    f32 = np.float32
    _, _, mag, ori = oracle.level_gradients(gauss_level)
    pmag, pori = np.pad(mag, 20, mode="edge"), np.pad(ori, 20, mode="edge")
    pts = rotated_window_points(kp_col + 20, kp_row + 20, 16, angle_deg)
    q = (np.arange(16)[:, None] * 16 + np.arange(16)[None, :]).ravel()       # :545, stride 16 into the 17-wide list
    lin = pts[q, 0].astype(np.int64) * pmag.shape[1] + pts[q, 1]              # at<>(x, y): x is the row
    if lin.min() < 0 or lin.max() >= pmag.size:
        return None
    m = pmag.ravel()[lin].reshape(16, 16)
    o = pori.ravel()[lin].reshape(16, 16)
    k = np.asarray(kernel_f32, f32)
    n, R = len(k), len(k) // 2
    ext = m
    while ext.shape[0] < 16 + 2 * R:  # BORDER_REFLECT_101 of the 16 x 16 Mat itself, repeated
        p = min(R - (ext.shape[0] - 16) // 2, ext.shape[0] - 1)
        ext = np.pad(ext, p, mode="reflect")
    off = (ext.shape[0] - 16) // 2 - R
    ext = ext[off:off + 16 + 2 * R, off:off + 16 + 2 * R]
    rowf = (k[0] * ext[:, 0:16]).astype(f32)
    for i in range(1, n):
        rowf = (rowf + (k[i] * ext[:, i:i + 16]).astype(f32)).astype(f32)
    colf = (k[R] * rowf[R:R + 16]).astype(f32)
    for i in range(1, R + 1):
        colf = (colf + (k[R + i] * (rowf[R + i:R + i + 16] + rowf[R - i:R - i + 16]).astype(f32)).astype(f32)).astype(f32)
    idx = (o * f32(f32(8) / f32(360.0))).astype(f32).astype(np.int32)
    d = np.zeros(128, f32)
    for rb in range(4):
        for cb in range(4):
            h = np.zeros(8, f32)
            for i in range(4 * rb, 4 * rb + 4):
                for j in range(4 * cb, 4 * cb + 4):
                    h[idx[i, j]] += colf[i, j]
            d[(rb * 4 + cb) * 8:(rb * 4 + cb) * 8 + 8] = h
    with np.errstate(all="ignore"):
        mx = d[0]
        for v in d[1:]:
            if mx < v:
                mx = v
        d = (d / mx).astype(f32)
        d = np.where(f32(0.2) < d, f32(0.2), d).astype(f32)
        mx = d[0]
        for v in d[1:]:
            if mx < v:
                mx = v
        d = (d / mx).astype(f32)
    return d
