// Generic (any size / any window / any kernel width) HIP kernels for gfx950.
// They back the host-buffer primitives of include/vslam.h for arbitrary parameters and
// are the fallback of the batched path for parameter sets the specialised kernels
// (kernels_harris_strip.hip.h, kernels_pyramid.hip.h) do not cover.  One thread per output
// element, coalesced row accesses, caches do the rest.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_localize.hip.h"

#include <cfloat>
#include <cstdint>

namespace vslam {

// cv::borderInterpolate(p, len, BORDER_REFLECT_101), repeated until inside.
__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Correctly rounded f32 square root (cv::magnitude on x86 is sqrtss / sqrtps) for the arguments this
// library has: x*x + y*y of integer Sobel differences, i.e. integers in [0, 2 * 255^2] - normal numbers
// or zero, so the denormal scaling of the general sequence is not needed.  (Round 3's form, kept as the reference of
// tools/sqrt_check.hip; the kernels use the 6-operation sqrt_rn_small_nr / sqrt_rn_small_pk below.)  v_sqrt_f32 is within one ulp;
// the two exact residuals (one FMA each) decide between the result and its neighbours.  9 instructions
// instead of the f64 square root the round-2 kernels used (tools/sqrt_check.hip compares the two over
// every possible argument).
__device__ __forceinline__ float sqrt_rn_small(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    s = rm <= 0.0f ? sm : s;  // s*(s - ulp) >= x: the result was one ulp high
    s = rp > 0.0f ? sp : s;   // s*(s + ulp) <  x: one ulp low
    return s;
}

// The same square root for two arguments at once in packed-f32 form (gfx90a+: v_pk_mul_f32 / v_pk_fma_f32 do two
// lanes' worth per issue): r = rsq(max(x, tiny)) (within one ulp), s = x * r, then ONE Newton step on the exact
// residual, s' = fma(fma(-s, s, x), r / 2, s).  6 issue slots per PAIR instead of 18.  That this is the correctly
// rounded root for every argument the library can form is checked exhaustively by tools/sqrt_check.hip (on the GPU:
// it depends on this chip's v_rsq_f32); x = 0 gives s = 0 * rsq(tiny) = 0 and a zero residual.
typedef float vslam_f2 __attribute__((ext_vector_type(2)));
typedef float vslam_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ vslam_f2 sqrt_rn_small_pk(vslam_f2 x) {
    vslam_f2 r;
    r.x = __builtin_amdgcn_rsqf(fmaxf(x.x, 1e-30f));
    r.y = __builtin_amdgcn_rsqf(fmaxf(x.y, 1e-30f));
    const vslam_f2 s = x * r, h = r * 0.5f;
    const vslam_f2 e = __builtin_elementwise_fma(-s, s, x);
    return __builtin_elementwise_fma(e, h, s);
}

// One argument, the same six operations (tools/sqrt_check.hip covers it as well).
__device__ __forceinline__ float sqrt_rn_small_nr(float x) {
    const float r = __builtin_amdgcn_rsqf(fmaxf(x, 1e-30f));
    const float s = x * r, h = r * 0.5f;
    const float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}

// cv::convertScaleAbs element as the reference's x86-64 OpenCV evaluates it: cvRound (cvtss2si /
// cvtps2dq, round half even) returns INT_MIN for NaN and for |x| >= 2^31, which
// saturate_cast<uchar> maps to 0; [255.5, 2^31) saturates to 255.
__device__ __forceinline__ int cvt_abs_u8(float x) {
    const float a = fabsf(x);
    if (!(a < 2147483648.0f)) return 0;
    return __float2int_rn(fminf(a, 255.0f));
}

// ---- GaussianBlur CV_8U, separable, exact integer -------------

// h(y,x) = sum_k taps[k] * src(y, reflect101(x - r + k)); <= 255*256, fits u16.
__global__ __launch_bounds__(256) void k_blur_h_generic(const uint8_t* __restrict__ src, size_t sstep,
                                                         size_t sframe, uint16_t* __restrict__ h, size_t hframe,
                                                         int rows, int cols, const uint16_t* __restrict__ taps,
                                                         int n) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cols) return;
    const uint8_t* s = src + blockIdx.z * sframe + (size_t)y * sstep;
    const int r = n >> 1;
    uint32_t acc = 0;
    if (x - r >= 0 && x + r < cols) {
        for (int k = 0; k < n; ++k) acc += (uint32_t)taps[k] * s[x - r + k];
    } else {
        for (int k = 0; k < n; ++k) acc += (uint32_t)taps[k] * s[reflect101(x - r + k, cols)];
    }
    h[blockIdx.z * hframe + (size_t)y * cols + x] = (uint16_t)acc;
}

// dst(y,x) = (sum_k taps[k] * h(reflect101(y - r + k), x) + 32768) >> 16.
__global__ __launch_bounds__(256) void k_blur_v_generic(const uint16_t* __restrict__ h, size_t hframe,
                                                         uint8_t* __restrict__ dst, size_t dstep, size_t dframe,
                                                         int rows, int cols, const uint16_t* __restrict__ taps,
                                                         int n) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= cols) return;
    const uint16_t* hp = h + blockIdx.z * hframe + x;
    const int r = n >> 1;
    uint32_t acc = 32768u;
    if (y - r >= 0 && y + r < rows) {
        for (int k = 0; k < n; ++k) acc += (uint32_t)taps[k] * hp[(size_t)(y - r + k) * cols];
    } else {
        for (int k = 0; k < n; ++k) acc += (uint32_t)taps[k] * hp[(size_t)reflect101(y - r + k, rows) * cols];
    }
    dst[blockIdx.z * dframe + (size_t)y * dstep + x] = (uint8_t)(acc >> 16);
}

// D_l = saturate_u8(G_{l+1} - G_l), l = 0..4 (GaussPyramid.cpp:191-200).  gauss: 6
// images P bytes apart, dog: 5 images P bytes apart; dense rows.
__global__ __launch_bounds__(256) void k_dog5(const uint8_t* __restrict__ gauss, uint8_t* __restrict__ dog,
                                               size_t P, size_t frame) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    const uint8_t* g = gauss + blockIdx.z * frame + i;
    uint8_t* d = dog + blockIdx.z * frame + i;
    int prev = g[0];
#pragma unroll
    for (int l = 0; l < 5; ++l) {
        const int cur = g[(size_t)(l + 1) * P];
        d[(size_t)l * P] = (uint8_t)(cur > prev ? cur - prev : 0);
        prev = cur;
    }
}

// ---- resize --------------------------------------------------------

// Coefficients of cv::resize INTER_LINEAR x2 along one axis: source index and the
// 11-bit weight of the SECOND sample (0, 512 or 1536); the first weight is 2048 - w1.
__device__ __forceinline__ void lin2x_coeff_x(int d, int n_src, int& s, int& w1) {
    // f = (d + 0.5)/2 - 0.5 ; even d: floor = d/2 - 1, frac .75 ; odd d: floor = d/2, frac .25
    s = (d >> 1) - 1 + (d & 1);
    w1 = (d & 1) ? 512 : 1536;
    if (s < 0) s = 0, w1 = 0;
    if (s >= n_src - 1) s = n_src - 1, w1 = 0;
}

__global__ __launch_bounds__(256) void k_resize_linear2x(const uint8_t* __restrict__ src, size_t sstep,
                                                          size_t sframe, uint8_t* __restrict__ dst, size_t dstep,
                                                          size_t dframe, int rows, int cols) {
    const int dx = blockIdx.x * 256 + threadIdx.x;
    const int dy = blockIdx.y;
    if (dx >= 2 * cols) return;
    int sx, a1;
    lin2x_coeff_x(dx, cols, sx, a1);
    const int a0 = 2048 - a1;
    const int sx1 = sx + 1 < cols ? sx + 1 : sx;
    // vertical: no coefficient clamping, rows are clipped instead (resizeGeneric_Invoker)
    const int sy = (dy >> 1) - 1 + (dy & 1);
    const int b1 = (dy & 1) ? 512 : 1536, b0 = 2048 - b1;
    const uint8_t* s0 = src + blockIdx.z * sframe + (size_t)clampi(sy, 0, rows - 1) * sstep;
    const uint8_t* s1 = src + blockIdx.z * sframe + (size_t)clampi(sy + 1, 0, rows - 1) * sstep;
    const int h0 = s0[sx] * a0 + s0[sx1] * a1;
    const int h1 = s1[sx] * a0 + s1[sx1] * a1;
    const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    dst[blockIdx.z * dframe + (size_t)dy * dstep + dx] = (uint8_t)clampi(v, 0, 255);
}

__global__ __launch_bounds__(256) void k_resize_nearest_half(const uint8_t* __restrict__ src, size_t sstep,
                                                              size_t sframe, uint8_t* __restrict__ dst,
                                                              size_t dstep, size_t dframe, int rows, int cols,
                                                              int drows, int dcols) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (x >= dcols) return;
    const int sy = min(2 * y, rows - 1), sx = min(2 * x, cols - 1);
    dst[blockIdx.z * dframe + (size_t)y * dstep + x] = src[blockIdx.z * sframe + (size_t)sy * sstep + sx];
}

// ---- Sobel ksize=1, convertScaleAbs -----------------------------------------------------

__global__ __launch_bounds__(256) void k_sobel_k1(const uint8_t* __restrict__ src, size_t sstep,
                                                   float* __restrict__ dst, size_t dstep_elems, int rows,
                                                   int cols, int dx) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= cols) return;
    int a, b;
    if (dx) {
        a = src[(size_t)r * sstep + reflect101(c + 1, cols)];
        b = src[(size_t)r * sstep + reflect101(c - 1, cols)];
    } else {
        a = src[(size_t)reflect101(r + 1, rows) * sstep + c];
        b = src[(size_t)reflect101(r - 1, rows) * sstep + c];
    }
    dst[(size_t)r * dstep_elems + c] = (float)(a - b);
}

__global__ __launch_bounds__(256) void k_convert_scale_abs(const float* __restrict__ src, size_t sstep_elems,
                                                            uint8_t* __restrict__ dst, size_t dstep, int rows,
                                                            int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= cols) return;
    dst[(size_t)r * dstep + c] = (uint8_t)cvt_abs_u8(src[(size_t)r * sstep_elems + c]);
}

// ---- HarrisCorner(Mat& Ix, Mat& Iy), literal op order (Harris_corners.cpp:10-68) ---------
// This TU is compiled with -ffp-contract=off, so every f32/f64 operation below is
// individually rounded exactly like the reference's SSE2 code.
__global__ __launch_bounds__(256) void k_harris_from_grad(const float* __restrict__ ix,
                                                           const float* __restrict__ iy, size_t step_elems,
                                                           int rows, int cols, float k, int pad,
                                                           float* __restrict__ resp, size_t rstep_elems) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= cols) return;
    float Ix2 = 0.0f, Iy2 = 0.0f, IxIy = 0.0f;
    for (int u = r - pad; u <= r + pad; ++u) {
        const size_t ro = (size_t)clampi(u, 0, rows - 1) * step_elems;
        for (int v = c - pad; v <= c + pad; ++v) {
            const int vv = clampi(v, 0, cols - 1);
            const float a = ix[ro + vv], b = iy[ro + vv];
            Ix2 += a * a;
            Iy2 += b * b;
            IxIy += a * b;
        }
    }
    const float det = (float)((double)Ix2 * (double)Iy2 - (double)IxIy * (double)IxIy);
    const float tr = (float)((double)Ix2 + (double)Iy2);
    const float trtr = tr * tr;
    const float ktr = k * trtr;
    const float response = det - ktr;
    resp[(size_t)r * rstep_elems + c] = response > 0 ? response : 0.0f;
}

// ---- NonMaximumSuppression / NMS2, any window --------------------------------------------

template <typename T>
__global__ __launch_bounds__(256) void k_nms_strict_generic(const T* __restrict__ src, size_t sstep_elems,
                                                             int rows, int cols, int p,
                                                             uint8_t* __restrict__ mask, size_t mstep) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= cols) return;
    // dilate's constant border never wins: minimum of the type
    float m = sizeof(T) == 1 ? 0.0f : -FLT_MAX;
    for (int u = max(r - p, 0); u <= min(r + p, rows - 1); ++u)
        for (int v = max(c - p, 0); v <= min(c + p, cols - 1); ++v) {
            if (u == r && v == c) continue;
            const float t = (float)src[(size_t)u * sstep_elems + v];
            if (t > m) m = t;
        }
    mask[(size_t)r * mstep + c] = (float)src[(size_t)r * sstep_elems + c] > m ? 255 : 0;
}

// NMS2: half-open window [i-p, i+p) x [j-p, j+p), '>=' (Harris_corners.cpp:94-119).
// true_max_bits accumulates the :107-109 running maximum as the bit pattern of a
// non-negative float (monotone under unsigned compare).
__global__ __launch_bounds__(256) void k_nms2_generic(const float* __restrict__ resp, size_t sstep_elems,
                                                       int rows, int cols, int p, float* __restrict__ out,
                                                       size_t ostep_elems, unsigned int* true_max_bits) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    const int i = blockIdx.y;
    float wmax = 0.0f;
    const bool inside = j < cols && i >= p && i < rows - p && j >= p && j < cols - p;
    if (inside) {
        for (int u = i - p; u < i + p; ++u)
            for (int v = j - p; v < j + p; ++v) {
                const float t = resp[(size_t)u * sstep_elems + v];
                if (t > wmax) wmax = t;
            }
    }
    if (j < cols) {
        float o = 0.0f;
        if (inside && resp[(size_t)i * sstep_elems + j] >= wmax) o = wmax;
        out[(size_t)i * ostep_elems + j] = o;
    }
    if (true_max_bits) {
        float m = wmax;
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(true_max_bits, __float_as_uint(m));
    }
}

// ---- scale-space extrema (initialKeypointDetection, Diff_of_Gauss.cpp:254-297) ------------

// Pitched plane -> dense rows (pyramid getters of widths that are not a multiple of 16).
__global__ __launch_bounds__(256) void k_pack_rows(const uint8_t* __restrict__ src, int spitch, uint8_t* __restrict__ dst,
                                                    int rows, int cols) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c < cols) dst[(size_t)r * cols + c] = src[(size_t)r * spitch + c];
}

struct ExtGeom {
    int n_oct, window, pad, min_contrast;
    int localize;  // list = FeaturePointLocalization survivors (min_contrast unused)
    int rows[VSLAM_MAX_OCTAVES], cols[VSLAM_MAX_OCTAVES], pitch[VSLAM_MAX_OCTAVES];  // pitch: bytes per plane row
    int lat_rows[VSLAM_MAX_OCTAVES], lat_cols[VSLAM_MAX_OCTAVES], wpr[VSLAM_MAX_OCTAVES];
    unsigned long long oct_off[VSLAM_MAX_OCTAVES];   // byte offset of the octave in a pyramid frame block
    unsigned long long bits_off[VSLAM_MAX_OCTAVES];  // word offset of the octave in a bits frame block
    const float* loc_lut;  // table of the localization's quadratic term for small differences, may be null
};

// The three finite differences of FeaturePointLocalization (Diff_of_Gauss.cpp:226-228) at padded
// (i, j) of DoG `level`; padOctave's replicate border = clamped addressing.
__device__ __forceinline__ void dog_differences(const uint8_t* __restrict__ dog, size_t P, int rows, int cols, int pitch, int pad,
                                                int level, int i, int j, int& d_x, int& d_y, int& d_s) {
    const int r = clampi(i - pad, 0, rows - 1), c = clampi(j - pad, 0, cols - 1);
    const uint8_t* D = dog + (size_t)level * P;
    d_x = (int)D[(size_t)r * pitch + clampi(j - 1 - pad, 0, cols - 1)] - (int)D[(size_t)r * pitch + clampi(j + 1 - pad, 0, cols - 1)];
    d_y = (int)D[(size_t)clampi(i - 1 - pad, 0, rows - 1) * pitch + c] - (int)D[(size_t)clampi(i + 1 - pad, 0, rows - 1) * pitch + c];
    d_s = (int)D[(size_t)r * pitch + c - P] - (int)D[(size_t)r * pitch + c + P];
}

// One thread per lattice site; a wave's 64 candidate flags leave as one ballot word, which
// IS the bitmask layout of include/vslam.h.  blockIdx.z = frame*3 + (level-1).
__global__ __launch_bounds__(256) void k_extrema(const uint8_t* __restrict__ pyr, size_t pframe, ExtGeom g,
                                                  int o, unsigned long long* __restrict__ bits,
                                                  unsigned long long* __restrict__ lflags, size_t bframe) {
    const int lj = blockIdx.x * 256 + threadIdx.x;
    const int li = blockIdx.y;
    const int f = blockIdx.z / 3, level = blockIdx.z % 3 + 1;
    const int rows = g.rows[o], cols = g.cols[o], pitch = g.pitch[o], pad = g.pad;
    const size_t P = (size_t)rows * pitch;
    const uint8_t* dog = pyr + f * pframe + g.oct_off[o] + (size_t)VSLAM_NUM_LEVELS * P;
    bool cand = false, listed = false;
    if (lj < g.lat_cols[o]) {
        const int i = pad + li * g.window, j = pad + lj * g.window;  // padded coordinates
        int mn = 256, mx = -1;
        for (int u = i - pad; u < i + pad; ++u) {
            const size_t ro = (size_t)clampi(u - pad, 0, rows - 1) * pitch;
            for (int v = j - pad; v < j + pad; ++v) {
                const size_t idx = ro + clampi(v - pad, 0, cols - 1);
#pragma unroll
                for (int l = -1; l <= 1; ++l) {
                    const int t = dog[(size_t)(level + l) * P + idx];
                    mn = min(mn, t);
                    mx = max(mx, t);
                }
            }
        }
        const int self = dog[(size_t)level * P + (size_t)(i - pad) * pitch + (j - pad)];
        cand = self == mn || self == mx;
        if (!g.localize) {
            listed = cand && self >= g.min_contrast;
        } else if (cand) {
            int d_x, d_y, d_s, nv;
            dog_differences(dog, P, rows, cols, pitch, pad, level, i, j, d_x, d_y, d_s);
            listed = feature_point_localization(d_x, d_y, d_s, self, nv, g.loc_lut);
        }
    }
    const unsigned long long wc = __ballot(cand), wl = __ballot(listed);
    if ((threadIdx.x & 63) == 0 && (lj >> 6) < g.wpr[o]) {
        const size_t w = f * bframe + g.bits_off[o] + ((size_t)(level - 1) * g.lat_rows[o] + li) * g.wpr[o] + (lj >> 6);
        if (bits) bits[w] = wc;
        lflags[w] = wl;
    }
}

}  // namespace vslam
