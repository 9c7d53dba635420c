// FeaturePointLocalization (Diff_of_Gauss.cpp:223-251): the contrast
// test the reference applies to every scale-space candidate before keeping it.  Per candidate
// and tiny; what matters is that each operation is rounded exactly where the reference's
// OpenCV calls round (f32 / f64 mix spelled out below, -ffp-contract=off), because B = A*A^T
// is singular by construction and the outcome for three non-zero differences is rounding noise
// that has to come out the same on the GPU.
#pragma once
#include <hip/hip_runtime.h>

#include <stdint.h>

namespace vslam {

// (int)x as the reference's x86-64 build does it (cvttss2si: INT_MIN when out of range or NaN)
__device__ __forceinline__ int cvtt_f32_i32(float x) {
    if (!(x > -2147483904.0f && x < 2147483648.0f)) return INT32_MIN;
    return (int)x;
}

// The quadratic term 0.5 * A^T * z_hat of :241 for three NON-ZERO differences (with a zero
// difference det(A A^T) is exactly 0, cv::invert returns zeros and the term is +-0).  It depends
// only on the magnitudes: flipping the sign of one difference flips matching pairs of factors in
// every product below, and IEEE rounding is sign-symmetric.
__device__ __forceinline__ float localization_quadratic_term(int d_x, int d_y, int d_scale) {
    const float a0 = (float)d_x / 255.0f, a1 = (float)d_y / 255.0f, a2 = (float)d_scale / 255.0f;  // :233
    // :238 outer product, each exact product rounded once to f32; held as f64 for :239
    const double b00 = (double)(a0 * a0), b01 = (double)(a0 * a1), b02 = (double)(a0 * a2);
    const double b11 = (double)(a1 * a1), b12 = (double)(a1 * a2), b22 = (double)(a2 * a2);
    const double b10 = b01, b20 = b02, b21 = b12;
    // :239 cv::invert, 3x3 CV_32F closed form
    double d = b00 * (b11 * b22 - b12 * b21) - b01 * (b10 * b22 - b12 * b20) + b02 * (b10 * b21 - b11 * b20);
    float i00 = 0.f, i01 = 0.f, i02 = 0.f, i10 = 0.f, i11 = 0.f, i12 = 0.f, i20 = 0.f, i21 = 0.f, i22 = 0.f;
    if (d != 0.) {
        d = 1. / d;
        i00 = (float)((b11 * b22 - b12 * b21) * d);
        i01 = (float)((b02 * b21 - b01 * b22) * d);
        i02 = (float)((b01 * b12 - b02 * b11) * d);
        i10 = (float)((b12 * b20 - b10 * b22) * d);
        i11 = (float)((b00 * b22 - b02 * b20) * d);
        i12 = (float)((b02 * b10 - b00 * b12) * d);
        i20 = (float)((b10 * b21 - b11 * b20) * d);
        i21 = (float)((b01 * b20 - b00 * b21) * d);
        i22 = (float)((b00 * b11 - b01 * b10) * d);
    }
    // :239 negation, :240 z_hat = B_inverse * A in f32, left to right
    const float z0 = ((-i00) * a0 + (-i01) * a1) + (-i02) * a2;
    const float z1 = ((-i10) * a0 + (-i11) * a1) + (-i12) * a2;
    const float z2 = ((-i20) * a0 + (-i21) * a1) + (-i22) * a2;
    // :241 gemm(A_T, z_hat, alpha = 0.5): f64 accumulation from 0, scaled, narrowed
    double s = 0.;
    s += (double)z0 * (double)a0;
    s += (double)z1 * (double)a1;
    s += (double)z2 * (double)a2;
    return (float)(s * 0.5);
}

// keep (dog_zhat > 0.03f, :245) and the value stored at :246, from the candidate's value and the
// quadratic term (0 when a difference is 0: value/255 + (+-0) is value/255 either way).
__device__ __forceinline__ bool localization_finish(int value, float quad, int& new_value) {
    const float dog_zhat = (float)value / 255.0f + quad;
    new_value = cvtt_f32_i32(dog_zhat * 255.0f);
    return dog_zhat > 0.03f;
}

// Table of the quadratic term for |d| < LOC_LUT_N in all three differences (4096 floats), filled
// once per context by k_build_localization_lut with the function above - so a lookup IS that
// arithmetic.  On the synthetic 1080p frames every candidate with three non-zero differences has
// magnitudes below 16, and the ~70 f64 operations become one 4-byte load.
constexpr int LOC_LUT_N = 16;
__device__ __forceinline__ int loc_lut_index(int ax, int ay, int as) { return (ax * LOC_LUT_N + ay) * LOC_LUT_N + as; }

__global__ __launch_bounds__(256) void k_build_localization_lut(float* __restrict__ lut) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= LOC_LUT_N * LOC_LUT_N * LOC_LUT_N) return;
    const int as = i % LOC_LUT_N, ay = (i / LOC_LUT_N) % LOC_LUT_N, ax = i / (LOC_LUT_N * LOC_LUT_N);
    lut[i] = (ax && ay && as) ? localization_quadratic_term(ax, ay, as) : 0.0f;
}

// Returns keep and the value stored at :246.  lut may be null.
__device__ __forceinline__ bool feature_point_localization(int d_x, int d_y, int d_scale, int value, int& new_value,
                                                           const float* __restrict__ lut = nullptr) {
    float quad = 0.0f;
    if (d_x != 0 && d_y != 0 && d_scale != 0) {
        const int ax = abs(d_x), ay = abs(d_y), as = abs(d_scale);
        if (lut && (ax | ay | as) < LOC_LUT_N)
            quad = lut[loc_lut_index(ax, ay, as)];
        else
            quad = localization_quadratic_term(d_x, d_y, d_scale);
    }
    return localization_finish(value, quad, new_value);
}

// Per-point form behind vslam_localize_points: in = (d_x, d_y, d_scale, value), out = (keep, value').
__global__ __launch_bounds__(256) void k_localize_points(const int4* __restrict__ in, int n, int2* __restrict__ out,
                                                          const float* __restrict__ lut) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int4 q = in[i];
    int nv;
    const bool keep = feature_point_localization(q.x, q.y, q.z, q.w, nv, lut);
    out[i] = make_int2(keep ? 1 : 0, keep ? nv : q.w);
}

}  // namespace vslam
