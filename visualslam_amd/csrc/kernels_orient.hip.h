// filterKeypoints (Diff_of_Gauss.cpp:301-372): edge rejection by
// computeEdgeResponse (:79-109), then the 36-bin orientation histogram (:112-133) of the
// keypoint's 16x16 window, magnitudes weighted by GaussianBlur(sigma = 1.5 * sigma(o, l)).
//
// The reference blurs a 16x16 ROI of the 8-padded magnitude image once per keypoint; the ROI is
// not isolated, so the filter reads the parent around the window and reflect-101 applies at the
// PARENT's edges.  One workgroup per keypoint does exactly that: the row pass over the
// (16 + 2R) x 16 strip the column pass needs goes to LDS, the symmetric column pass, the
// histogram and the peak test follow in the workgroup.  Every f32 operation is rounded where
// OpenCV's scalar filters round (row: s = k0*S0; s += ki*Si left to right; column: s = kc*S0;
// s += k(c+i)*(S(+i) + S(-i)); histogram: row-major += per bin), so the result is the oracle's
// bit for bit.  Peaks leave as a 36-bit mask per keypoint; kernels_compact.hip.h turns the masks
// into the ordered SLAM::point list.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_generic.hip.h"

namespace vslam {

constexpr int OR_WIN = 16;   // windowSize, Diff_of_Gauss.cpp:304
constexpr int OR_PAD = 8;    // padding = windowSize / 2, :305
constexpr int OR_BINS = 36;  // :352

struct OrientLevels {  // per Gaussian level of the octave (null / 0 when no keypoint uses it)
    const uint8_t* gauss[VSLAM_NUM_LEVELS];
    const float* mag[VSLAM_NUM_LEVELS];
    const float* orient[VSLAM_NUM_LEVELS];
    const float* kern[VSLAM_NUM_LEVELS];
    int kn[VSLAM_NUM_LEVELS];
};

// computeEdgeResponse on the Sobel (ksize 1, reflect-101) gradients of an 8-bit Gaussian level,
// formed on the fly (they are integers, exact in f32).  Window [y-p, y+p) x [x-p, x+p) of the
// unpadded image at the keypoint's padded coordinates; out-of-image reads (only possible for
// windowSize > 3, where the reference itself reads out of bounds) are clamped like the oracle.
__device__ __forceinline__ float edge_response_u8(const uint8_t* __restrict__ G, int gpitch, int rows, int cols, int y, int x, int pad) {
    float Ix2 = 0.f, Iy2 = 0.f, IxIy = 0.f;
    for (int u = y - pad; u < y + pad; ++u) {
        const int r = clampi(u, 0, rows - 1);
        for (int v = x - pad; v < x + pad; ++v) {
            const int c = clampi(v, 0, cols - 1);
            const float gx = (float)((int)G[(size_t)r * gpitch + reflect101(c + 1, cols)] - (int)G[(size_t)r * gpitch + reflect101(c - 1, cols)]);
            const float gy = (float)((int)G[(size_t)reflect101(r + 1, rows) * gpitch + c] - (int)G[(size_t)reflect101(r - 1, rows) * gpitch + c]);
            Ix2 += gx * gx;
            Iy2 += gy * gy;
            IxIy += gx * gy;
        }
    }
    const float det = (float)((double)Ix2 * (double)Iy2 - (double)IxIy * (double)IxIy);  // cv::determinant, :105
    const float tr = (float)(0.0 + (double)Ix2 + (double)Iy2);                            // cv::trace, :106
    return (tr * tr) / det;                                                               // :107
}

// Lane B of the wave receives the 64-bit mask of the lanes whose bin is B, for B = 0 .. OR_BINS - 1: one compare into VCC
// and two v_writelane with an immediate lane select per bin (this compiler has no builtin for v_writelane; the wait
// states between the compare and the reads of VCC are written out because the hazard recogniser does not look inside
// inline asm).
template <int B>
__device__ __forceinline__ void bin_masks_to_lanes(int mybin, unsigned int& lo, unsigned int& hi) {
    if constexpr (B < OR_BINS) {
        asm volatile("v_cmp_eq_u32_e32 vcc, %2, %3\n\ts_nop 3\n\tv_writelane_b32 %0, vcc_lo, %2\n\tv_writelane_b32 %1, vcc_hi, %2"
                     : "+v"(lo), "+v"(hi)
                     : "n"(B), "v"(mybin)
                     : "vcc");
        bin_masks_to_lanes<B + 1>(mybin, lo, hi);
    }
}

// grid = (keypoints), 256 threads, dynamic LDS = orient_lds_bytes(max R).  FMA: the filter's multiply-adds fused (mad_f32).
template <bool FMA>
__global__ __launch_bounds__(256) void k_orient_keypoints(const vslam_point* __restrict__ kps, int n, OrientLevels lv, int gpitch, int rows, int cols,
                                                           unsigned long long* __restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) float orient_smem[];
    __shared__ float mw[OR_WIN * OR_WIN];
    __shared__ unsigned long long binmask[OR_BINS][4];  // per bin, per wave: the wave's pixels that fall into it
    __shared__ int keep_s;
    const int q = blockIdx.x;
    const vslam_point kp = kps[q];
    const int x = kp.col, y = kp.row, level = kp.level;
    if (threadIdx.x == 0) {
        const float r = 10.0f, threshold = ((r + 1.0f) * (r + 1.0f)) / r;  // :331-332
        const float response = edge_response_u8(lv.gauss[level], gpitch, rows, cols, y, x, kp.padding);
        keep_s = response < threshold;  // :335
    }
    __syncthreads();
    if (!keep_s) {
        if (threadIdx.x == 0) masks[q] = 0ull;
        return;
    }
    const int kn = lv.kn[level], R = kn >> 1;
    const float* __restrict__ k = lv.kern[level];
    const float* __restrict__ M = lv.mag[level];
    const int prows = rows + 2 * OR_PAD, pcols = cols + 2 * OR_PAD;
    const int span = OR_WIN + 2 * R;
    float* rb = orient_smem;                                  // [span][16] row-filtered strip
    int* cx = reinterpret_cast<int*>(orient_smem + span * OR_WIN);  // [span] source column of padded column x - R + i
    float* kl = orient_smem + span * OR_WIN + span;                 // [span >= kn] the taps: from LDS, not from memory, inside the filter loops
    for (int i = threadIdx.x; i < span; i += 256) {
        cx[i] = clampi(reflect101(x + i - R, pcols) - OR_PAD, 0, cols - 1);
        if (i < kn) kl[i] = k[i];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < span * OR_WIN; it += 256) {
        const int rr = it >> 4, c = it & 15;
        const int sy = clampi(reflect101(y + rr - R, prows) - OR_PAD, 0, rows - 1);  // padOctave = replicate
        const float* __restrict__ S = M + (size_t)sy * cols;
        float s0 = kl[0] * S[cx[c]];
#pragma unroll 4
        for (int i = 1; i < kn; ++i) s0 = mad_f32<FMA>(kl[i], S[cx[c + i]], s0);
        rb[it] = s0;
    }
    __syncthreads();
    int bin_of;
    {
        const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
        float s0 = kl[R] * rb[(i + R) * OR_WIN + j];
#pragma unroll 4
        for (int t = 1; t <= R; ++t) s0 = mad_f32<FMA>(kl[R + t], rb[(i + R + t) * OR_WIN + j] + rb[(i + R - t) * OR_WIN + j], s0);
        mw[threadIdx.x] = s0;
        const float reductionCoeff = (float)OR_BINS / 360.0f;  // :114
        const float o = lv.orient[level][(size_t)clampi(y + i - OR_PAD, 0, rows - 1) * cols + clampi(x + j - OR_PAD, 0, cols - 1)];
        const int index = (int)(o * reductionCoeff);  // :126; fastAtan2 on integer gradients stays below 359.8
        bin_of = min(max(index, 0), OR_BINS - 1);
    }
    // histogram (:112-133): a bin's magnitudes in pixel order.  Every wave ballots its 64 pixels bin by bin, the bin's
    // lane then adds only its own pixels (ascending bit = ascending pixel index) - as in k_orient_survivors
    {
        unsigned int mlo = 0, mhi = 0;
        bin_masks_to_lanes<0>(bin_of, mlo, mhi);
        if ((threadIdx.x & 63) < OR_BINS) binmask[threadIdx.x & 63][threadIdx.x >> 6] = ((unsigned long long)mhi << 32) | mlo;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        float h = 0.0f;
        if (threadIdx.x < OR_BINS) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                unsigned long long m = binmask[threadIdx.x][w];
                while (m) {
                    h += mw[64 * w + __builtin_ctzll(m)];
                    m &= m - 1;
                }
            }
        }
        float mx = threadIdx.x < OR_BINS ? h : 0.0f;  // sums of non-negative weights: 0 is neutral
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        const float peakThreshold = mx * 0.8f;                          // :358
        const bool peak = threadIdx.x < OR_BINS && h > peakThreshold;  // :362
        const unsigned long long m = __ballot(peak);
        if (threadIdx.x == 0) masks[q] = m;
    }
}

static inline size_t orient_lds_bytes(int R) { return (size_t)(OR_WIN + 2 * R) * OR_WIN * 4 + (size_t)(OR_WIN + 2 * R) * 8; }  // strip + column map + taps

// StructureMatrix (Harris_corners.cpp:10-29) / computeEdgeResponse (Diff_of_Gauss.cpp:79-109) on
// caller-gathered windows (the per-point C++ entry points): gxw / gyw hold n windows of `elems`
// gradient values in the reference's loop order.  sums (may be null): n x (Ix2, IxIy, Iy2), the
// entries of M; response (may be null): tr^2 / det.
__global__ __launch_bounds__(256) void k_edge_response_windows(const float* __restrict__ gxw, const float* __restrict__ gyw, int elems,
                                                                int n, float* __restrict__ sums, float* __restrict__ response) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float Ix2 = 0.f, Iy2 = 0.f, IxIy = 0.f;
    for (int e = 0; e < elems; ++e) {
        const float gx = gxw[(size_t)i * elems + e], gy = gyw[(size_t)i * elems + e];
        Ix2 += gx * gx;
        Iy2 += gy * gy;
        IxIy += gx * gy;
    }
    if (sums) sums[3 * (size_t)i] = Ix2, sums[3 * (size_t)i + 1] = IxIy, sums[3 * (size_t)i + 2] = Iy2;
    if (response) {
        const float det = (float)((double)Ix2 * (double)Iy2 - (double)IxIy * (double)IxIy);
        const float tr = (float)(0.0 + (double)Ix2 + (double)Iy2);
        response[i] = (tr * tr) / det;
    }
}

}  // namespace vslam
