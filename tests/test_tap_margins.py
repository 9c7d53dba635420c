"""How much room the quantised Gaussian taps leave for a different exp().

The oracle (and the library) form the Gaussian kernels with libm's exp; OpenCV's bit-exact path (`getGaussianKernelBitExact`,
smooth.dispatch.cpp) uses its softfloat exp.  Both are accurate to an ulp or so, but they are not the same function, and
OpenCV is not available here to compare.  What CAN be checked is that it does not matter: an 8.8 tap of
`getGaussianKernelFixedPoint_ED` changes only if one of the error-diffusion sums `k[i] * 256 + err` crosses a rounding
boundary, and the f32 kernels of filterKeypoints / SIFT only if a double lands within the perturbation of the midpoint of
two floats.  This test restates the two roundings with their intermediate values and measures the distance to the nearest
boundary for every kernel the default pyramid uses (up to five octaves): the closest call is 1.4e-4 of a tap unit (octave 4,
level 3, 309 taps) and 4.8e-5 of a float spacing, against the 4e-12 tap units / 2e-9 spacings a one-ulp difference in exp
can move them - seven and four orders of magnitude of room.
"""
import math

import numpy as np

import oracle

SIGMA0 = 1.6


def _kernel_f64(n, sigma):
    # oracle/vslam_oracle.c gauss_kernel_f64 (sigma > 0 branch), same operation order
    n2 = (n - 1) // 2
    scale = -0.125 / (sigma * sigma)
    t = [math.exp(float(x * x) * scale) for x in range(1 - n, 0, 2)][:n2]
    s = 0.0
    for v in t:
        s += v
    s = s * 2.0 + 1.0
    mul = 1.0 / s
    k = [v * mul for v in t]
    return k + [1.0 * mul] + k[::-1]


def _q8_with_margin(n, sigma):
    k = _kernel_f64(n, sigma)
    err, taps, margin = 0.0, [], 1.0
    for i in range(n // 2):
        adj = k[i] * 256.0 + err
        v = round(adj)  # half to even, like lrint
        margin = min(margin, abs(abs(adj - math.floor(adj)) - 0.5))
        err = adj - v
        taps.append(v)
    centre = 256 - 2 * sum(taps)
    return np.array(taps + [centre] + taps[::-1], dtype=np.uint16), margin


def _pyramid_kernels(n_octaves=5):
    for o in range(n_octaves):
        for l in range(6):
            yield o, l, oracle.sigma_at(SIGMA0, o, l)


def test_restatement_equals_the_oracle():
    for o, l, s in _pyramid_kernels():
        n = oracle.gauss_ksize_u8(s)
        taps, _ = _q8_with_margin(n, s)
        assert (taps == oracle.gauss_taps_q8(n, s)).all(), (o, l)
        n32 = oracle.gauss_ksize_f32(1.5 * s)
        assert (np.array(_kernel_f64(n32, 1.5 * s), dtype=np.float64).astype(np.float32) == oracle.gauss_kernel_f32(n32, 1.5 * s)).all(), (o, l)


def test_q8_taps_do_not_depend_on_the_last_bits_of_exp():
    worst = 1.0
    for o, l, s in _pyramid_kernels():
        n = oracle.gauss_ksize_u8(s)
        taps, margin = _q8_with_margin(n, s)
        worst = min(worst, margin)
        # the diffusion sums carry at most 256 * (n / 2) * eps of accumulated relative error: eps = 1e-12 (thousands of ulps) still leaves room
        assert margin > 256.0 * (n / 2) * 1e-12, (o, l, n, margin)
        # and directly: every exp value pushed up / down by 64 ulps gives the same taps
        for sign in (-1.0, 1.0):
            n2 = (n - 1) // 2
            scale = -0.125 / (s * s)
            t = [math.exp(float(x * x) * scale) * (1.0 + sign * 64 * 2.0 ** -52) for x in range(1 - n, 0, 2)][:n2]
            tot = sum(t) * 2.0 + 1.0
            k = [v / tot for v in t]
            err, q = 0.0, []
            for i in range(n // 2):
                adj = k[i] * 256.0 + err
                v = round(adj)
                err = adj - v
                q.append(v)
            assert q == list(taps[: n // 2]), (o, l, sign)
    assert worst > 1e-5, worst


def test_f32_orientation_kernels_do_not_depend_on_the_last_bits_of_exp():
    worst = 1.0
    for o, l, s in _pyramid_kernels():
        sg = 1.5 * s
        n = oracle.gauss_ksize_f32(sg)
        k = np.array(_kernel_f64(n, sg), dtype=np.float64)
        f = k.astype(np.float32)
        # distance of the double from the nearer rounding midpoint, in units of the float spacing at that value
        up, dn = np.nextafter(f, np.float32(np.inf)).astype(np.float64), np.nextafter(f, np.float32(-np.inf)).astype(np.float64)
        f64 = f.astype(np.float64)
        d = np.minimum(np.abs((f64 + up) / 2 - k), np.abs((f64 + dn) / 2 - k)) / (up - f64)
        worst = min(worst, float(d.min()))
        # a relative perturbation of 1e-12 moves a value by 1e-12 / 2^-24 = 1.7e-5 float spacings
        assert d.min() > 1.7e-5 * 0.06, (o, l, float(d.min()))
    assert worst > 1e-5, worst
