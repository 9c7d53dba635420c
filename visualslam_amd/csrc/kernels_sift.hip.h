// SIFT descriptor stage (Diff_of_Gauss.cpp:561-693 with rotateImageSection :528-559 and
// Rotation::getRotatedWindowPoints rotation.cpp:112-130).
//
// One 256-thread workgroup per oriented keypoint, thread (i, j) = ROI pixel:
//   1. its rotated sample point: entry 16 i + j of the 17 x 17 point list (:545 walks the 17-wide
//      list with stride 16), rotate_pt_CW in separately rounded f32 (rotation.cpp:22-23), the cosine
//      and sine of the keypoint's angle computed by the host (libm, like the reference);
//   2. the sample itself: Mat::at<>(point.x, point.y) on the 20-padded level images uses x as the ROW
//      (:549-554) and checks nothing, i.e. it reads linear element x * (cols + 40) + y of the padded
//      Mat.  All 256 indices inside the buffer -> defined; otherwise the reference reads foreign
//      memory and the keypoint is reported undefined (zero descriptor, defined[k] = 0).  Magnitude
//      and orientation of the sampled pixel are formed on the fly from the 8-bit Gaussian level
//      (processGradients, GaussPyramid.cpp:65-104: Sobel ksize 1, correctly rounded sqrt, fastAtan2),
//      the 20-pixel padding is replicate (clamp);
//   3. GaussianBlur of the 16 x 16 magnitude ROI (an isolated Mat: reflect-101 inside the 16 x 16,
//      repeated for the 25 ... 309-tap kernels), row pass then symmetric column pass in OpenCV's
//      accumulation order;
//   4. sixteen 4 x 4 sub-regions x 8 bins: thread (region, bin) walks its region's 16 pixels in
//      row-major order (the reference's += order);
//   5. the two max-normalisations with the 0.2 clip (:659-675), IEEE f32 division.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_aux.hip.h"
#include "kernels_generic.hip.h"

namespace vslam {

constexpr int SIFT_WIN = 16;       // windowSize, Diff_of_Gauss.cpp:565
constexpr int SIFT_PAD = 20;       // maxPadding, :571
constexpr int SIFT_DESC = 128;     // 16 sub-regions x 8 bins

struct SiftLevels {  // per Gaussian level of the octave (null / 0 when no keypoint uses it)
    const uint8_t* gauss[VSLAM_NUM_LEVELS];
    const float* kern[VSLAM_NUM_LEVELS];
    int kn[VSLAM_NUM_LEVELS];
};

// One keypoint by one 256-thread workgroup (all threads call this with the same arguments).
// G = the keypoint's 8-bit Gaussian level, k / kn = the f32 taps of sigma = 1.5 * sigma(octave, level),
// a = (cos, sin) of its angle.  Writes the 128 floats and the defined flag (0: the rotated window
// leaves the padded level, descriptor zeroed).  No barrier at the end: wave 0 may still be reading sh.d while the others start the
// next keypoint, whose first write to it is five barriers away.
constexpr int SIFT_KMAX = 640;  // blur taps staged in LDS (309 for octave 3 of the reference pyramid); longer kernels read them from memory
constexpr int SIFT_EXT_R = 47;  // kernels up to 95 taps (octaves 0 and 1 of the reference pyramid: 25 ... 77) filter explicitly extended rows
constexpr int SIFT_EXT_SPAN = SIFT_WIN + 2 * SIFT_EXT_R;
struct SiftShared {
    union {  // a keypoint takes one of the two filter forms
        struct {
            float ext[SIFT_WIN * SIFT_EXT_SPAN];   // rows of magROI extended by the repeated reflect-101, [row][R + 16 + R]
            float extc[SIFT_EXT_SPAN * SIFT_WIN];  // the row-filtered ROI extended likewise along the rows, [R + 16 + R][col]
        };
        struct {
            float kt[SIFT_KMAX];
            float rowf[SIFT_WIN * SIFT_WIN];  // row-filtered ROI
        };
    };
    float mag[SIFT_WIN * SIFT_WIN];   // magROI
    float mw[SIFT_WIN * SIFT_WIN];    // magWeighted
    float d[SIFT_DESC];
    uint8_t bin[SIFT_WIN * SIFT_WIN];
    int badw[4];  // per wave: one of its 64 sample points lies outside the padded Mat
};

template <bool FMA>
__device__ __forceinline__ void sift_one_keypoint(SiftShared& sh, const vslam_point kp, const float2 a, const uint8_t* __restrict__ G,
                                                  int gpitch, int rows, int cols, const float* __restrict__ k, int kn,
                                                  float* __restrict__ desc, uint8_t* __restrict__ defined) {
    const int t = threadIdx.x;
    const int i = t >> 4, j = t & 15;
    const int pr = rows + 2 * SIFT_PAD, pc = cols + 2 * SIFT_PAD;
    float o = 0.0f, mval = 0.0f;
    {
        const int cx = kp.col + SIFT_PAD, cy = kp.row + SIFT_PAD;  // :591
        const int e = t;                                            // rotatedPoints[i * imgROI.rows + j], :545
        const int iy = e / (SIFT_WIN + 1), jx = e - iy * (SIFT_WIN + 1);
        const int rx = jx - SIFT_WIN / 2, ry = iy - SIFT_WIN / 2;  // pt - center
        const float xr = (float)rx * a.x - (float)ry * a.y;        // rotation.cpp:22 (no FMA: -ffp-contract=off)
        const float yr = (float)rx * a.y + (float)ry * a.x;        // :23
        const int px = (int)xr + cx, py = (int)yr + cy;            // truncation; |xr|, |yr| < 16
        // at<>(x, y): x is the row (:549-554), i.e. linear element lin = px * pc + py of the padded Mat; defined iff
        // 0 <= lin < pr * pc.  px, py > 0 here (centre >= 20, |offset| < 16), so lin >= 0 and floor(lin / pc) = px + py / pc:
        // py < pc unless the image is taller than wide - a subtraction loop of rows / cols trips instead of the 64-bit
        // division this was (a tenth of the kernel's instructions)
        int prow = px, pcol = py;
        while (pcol >= pc) pcol -= pc, ++prow;
        const bool outside = prow >= pr;
        // one flag word per wave instead of an atomic into a word that had to be cleared behind a barrier of its own
        const bool wave_bad = __builtin_amdgcn_ballot_w64(outside) != 0ull;
        if ((t & 63) == 0) sh.badw[t >> 6] = wave_bad;
        if (!outside) {
            const int r = clampi(prow - SIFT_PAD, 0, rows - 1), c = clampi(pcol - SIFT_PAD, 0, cols - 1);  // padOctave(20): replicate
            const float gx = (float)((int)G[(size_t)r * gpitch + reflect101(c + 1, cols)] - (int)G[(size_t)r * gpitch + reflect101(c - 1, cols)]);
            const float gy = (float)((int)G[(size_t)reflect101(r + 1, rows) * gpitch + c] - (int)G[(size_t)reflect101(r - 1, rows) * gpitch + c]);
            const float xx = gx * gx, yy = gy * gy;
            mval = sqrt_rn_small_nr(xx + yy);            // cv::magnitude, correctly rounded (kernels_generic.hip.h)
            sh.mag[t] = mval;
            o = fast_atan2_deg(gy, gx);                  // cv::phase(..., true)
            if ((kn >> 1) <= SIFT_EXT_R) {
                // Kernels up to 95 taps filter explicitly extended rows (below).  Pixel (i, j)'s magnitude stands at every extended
                // position of row i that reflects onto column j: R + j + 30 z and R - j + 30 z (round 3 walked the extended rows
                // element by element: a division and a modulo each).  Written here, by the pixel's own thread, the extension
                // needs no barrier of its own.
                constexpr int PER = 2 * (SIFT_WIN - 1);
                const int R = kn >> 1, span = SIFT_WIN + 2 * R;
                float* E = sh.ext + i * span;
                int p = R + j;  // < 47 + 16
                while (p >= PER) p -= PER;
                for (; p < span; p += PER) E[p] = mval;
                if (j != 0 && j != SIFT_WIN - 1) {
                    int p0 = R - j;
                    while (p0 < 0) p0 += PER;
                    while (p0 >= PER) p0 -= PER;
                    for (; p0 < span; p0 += PER) E[p0] = mval;
                }
            }
        }
    }
    __syncthreads();
    if (sh.badw[0] | sh.badw[1] | sh.badw[2] | sh.badw[3]) {  // block-uniform; the barrier below keeps a fast wave's next keypoint off these words
        if (t < SIFT_DESC) desc[t] = 0.0f;
        if (t == 0 && defined) *defined = 0;
        __syncthreads();
        return;
    }
    {
        const float reductionCoeff = (float)8 / 360.0f;  // :114 with size = 8 (:631)
        const int index = (int)(o * reductionCoeff);     // :126
        sh.bin[t] = (uint8_t)min(max(index, 0), 7);
    }
    const int R = kn >> 1;
    if (R <= SIFT_EXT_R) {
        // The ROI is 16 wide and the kernels 25 ... 77 taps: the reflect-101 extension of a row is the triangle wave of
        // period 30.  Written out once per row (and once per column of the row-filtered ROI), the two filter loops are a
        // tap, a value at a fixed stride and the multiply-add - no index arithmetic per tap (it was two thirds of the
        // loops' instructions).  Same products, same order.
        constexpr int PER = 2 * (SIFT_WIN - 1);
        const int span = SIFT_WIN + 2 * R;
        // (the extended rows were written with the samples, in front of the barrier that also covers the flags)
        // The taps come through SCALAR loads (the same address in every lane: s_load, the tap a scalar operand of the
        // multiply): with both operands read from the LDS the two filter loops were bound by LDS issue (two reads per
        // product for four waves), not by arithmetic.
        typedef const float __attribute__((address_space(4)))* ktaps_p;
        const ktaps_p ks = (ktaps_p)k;
        {   // row filter of ROI row i: s = k[0]*S[0]; s += k[m]*S[m]
            const float* S = sh.ext + i * span + j;
            float s0 = ks[0] * S[0];
#pragma unroll 4
            for (int m = 1; m < kn; ++m) s0 = mad_f32<FMA>(ks[m], S[m], s0);
            // its value stands at every extended row position that reflects onto row i: R + i + 30 z and R - i + 30 z
            for (int p = (R + i) % PER; p < span; p += PER) sh.extc[p * SIFT_WIN + j] = s0;
            if (i != 0 && i != SIFT_WIN - 1) {
                int p0 = (R - i) % PER;
                p0 = p0 < 0 ? p0 + PER : p0;
                for (int p = p0; p < span; p += PER) sh.extc[p * SIFT_WIN + j] = s0;
            }
        }
        __syncthreads();
        {   // symmetric column filter: s = k[R]*S(0); s += k[R+m]*(S(+m) + S(-m))
            const float* C = sh.extc + (R + i) * SIFT_WIN + j;
            float s0 = ks[R] * C[0];
#pragma unroll 4
            for (int m = 1; m <= R; ++m) s0 = mad_f32<FMA>(ks[R + m], C[m * SIFT_WIN] + C[-m * SIFT_WIN], s0);
            sh.mw[t] = s0;
        }
    } else if (kn <= SIFT_KMAX) {
        // Taps from LDS (read from memory inside the loops, every multiply waited for a load round trip), and the
        // repeated reflect-101 of the 16-wide ROI as the triangle wave of period 30 it is, advanced by one per tap
        // instead of a reflection loop per tap: position p -> q = p mod 30, index = q < 16 ? q : 30 - q.
        constexpr int PER = 2 * (SIFT_WIN - 1);
        for (int m = t; m < kn; m += 256) sh.kt[m] = k[m];
        __syncthreads();
        {   // row filter of ROI row i: s = k[0]*S[0]; s += k[m]*S[m], S = the row extended by reflect-101
            const float* S = sh.mag + i * SIFT_WIN;
            int q = (j - R) % PER;
            q = q < 0 ? q + PER : q;
            float s0 = sh.kt[0] * S[q < SIFT_WIN ? q : PER - q];
#pragma unroll 4
            for (int m = 1; m < kn; ++m) {
                q = q + 1 == PER ? 0 : q + 1;
                s0 = mad_f32<FMA>(sh.kt[m], S[q < SIFT_WIN ? q : PER - q], s0);
            }
            sh.rowf[t] = s0;
        }
        __syncthreads();
        {   // symmetric column filter: s = k[R]*S(0); s += k[R+m]*(S(+m) + S(-m)), rows reflected likewise
            float s0 = sh.kt[R] * sh.rowf[i * SIFT_WIN + j];
            int qp = i, qm = i;  // i is inside the ROI: 0 <= i < 16 < PER
#pragma unroll 4
            for (int m = 1; m <= R; ++m) {
                qp = qp + 1 == PER ? 0 : qp + 1;
                qm = qm == 0 ? PER - 1 : qm - 1;
                s0 = mad_f32<FMA>(sh.kt[R + m], sh.rowf[(qp < SIFT_WIN ? qp : PER - qp) * SIFT_WIN + j] + sh.rowf[(qm < SIFT_WIN ? qm : PER - qm) * SIFT_WIN + j], s0);
            }
            sh.mw[t] = s0;
        }
    } else {
        {
            const float* S = sh.mag + i * SIFT_WIN;
            float s0 = k[0] * S[reflect101(j - R, SIFT_WIN)];
            for (int m = 1; m < kn; ++m) s0 = mad_f32<FMA>(k[m], S[reflect101(j + m - R, SIFT_WIN)], s0);
            sh.rowf[t] = s0;
        }
        __syncthreads();
        {
            float s0 = k[R] * sh.rowf[i * SIFT_WIN + j];
            for (int m = 1; m <= R; ++m)
                s0 = mad_f32<FMA>(k[R + m], sh.rowf[reflect101(i + m, SIFT_WIN) * SIFT_WIN + j] + sh.rowf[reflect101(i - m, SIFT_WIN) * SIFT_WIN + j], s0);
            sh.mw[t] = s0;
        }
    }
    __syncthreads();
    if (t < SIFT_DESC) {  // thread = (sub-region, bin); regions row-major, columns advance first (:637-652)
        const int region = t >> 3, b = t & 7;
        const int r0 = (region >> 2) * 4, c0 = (region & 3) * 4;
        float h = 0.0f;
        for (int u = r0; u < r0 + 4; ++u)
            for (int v = c0; v < c0 + 4; ++v)
                if (sh.bin[u * SIFT_WIN + v] == b) h += sh.mw[u * SIFT_WIN + v];
        sh.d[t] = h;
    }
    __syncthreads();
    // *max_element (operator< scan), c / max, min(c, 0.2f) as std::min, again c / max (:659-675).  The scan keeps a NaN
    // first element and skips every later NaN, i.e. its result is d[0] if that is NaN and otherwise the IEEE maxNum of all
    // elements (v_max_f32 ignores a NaN operand; the values are sums of non-negative products, so no -0 can decide a tie).
    // Round 2 let each of the 128 threads scan all 128 values, twice; round 3 reduced them in the two waves that held
    // them, meeting through two LDS words and two barriers per maximum.  Now ONE wave holds two values per lane and both
    // maxima are lane exchanges: no barrier behind the histogram at all - the other three waves are already at the next
    // keypoint's samples (its first write to anything read here, sh.d, is five barriers away).
    if (t < 64) {
        const float d0 = sh.d[t], d1 = sh.d[t + 64];
        auto max_element128 = [&](float x0, float x1) -> float {
            const float first = __shfl(x0, 0);
            float m = fmaxf(x0 != x0 ? -INFINITY : x0, x1 != x1 ? -INFINITY : x1);  // NaNs do not take part; d[0]'s NaN is handled below
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            return first != first ? first : m;
        };
        const float mx1 = max_element128(d0, d1);
        float v0 = d0 / mx1, v1 = d1 / mx1;
        v0 = 0.2f < v0 ? 0.2f : v0;
        v1 = 0.2f < v1 ? 0.2f : v1;
        const float mx2 = max_element128(v0, v1);
        desc[t] = v0 / mx2;
        desc[t + 64] = v1 / mx2;
        if (t == 0 && defined) *defined = 1;
    }
}

// Per-image entry point: grid = (keypoints), block = 256.  cs[q] = (cos, sin) of keypoint q's angle.
template <bool FMA>  // the f32 filter's multiply-adds fused (mad_f32, kernels_aux.hip.h)
__global__ __launch_bounds__(256) void k_sift_descriptors(const vslam_point* __restrict__ kps, const float2* __restrict__ cs, int n, SiftLevels lv,
                                                           int gpitch, int rows, int cols, float* __restrict__ desc,
                                                           uint8_t* __restrict__ defined) {
    __shared__ SiftShared sh;
    const int q = blockIdx.x;
    const vslam_point kp = kps[q];
    const int level = __builtin_amdgcn_readfirstlane(kp.level);
    sift_one_keypoint<FMA>(sh, kp, cs[q], lv.gauss[level], gpitch, rows, cols, lv.kern[level], lv.kn[level],
                      desc + (size_t)q * SIFT_DESC, defined ? defined + q : nullptr);
}

// Batched, device-resident form (params.describe): the oriented points of every frame of a batch
// (filterKeypoints output of kernels_orient_batch.hip.h: angle = bin * 10), grid = (G, frames),
// workgroups stride over the frame's list.  cs36[b] = (cos, sin) of b * 10 degrees from the host's
// libm (the angles the pipeline produces); the blur taps are those of the orientation stage
// (same sigma = 1.5 * sigma(octave, level), Diff_of_Gauss.cpp:346 and :616).
struct SiftBatchGeom {
    int rows[VSLAM_MAX_OCTAVES], cols[VSLAM_MAX_OCTAVES], pitch[VSLAM_MAX_OCTAVES];
    unsigned long long oct_off[VSLAM_MAX_OCTAVES];
    const float* kern[VSLAM_MAX_OCTAVES][VSLAM_NUM_LEVELS];
    int kn[VSLAM_MAX_OCTAVES][VSLAM_NUM_LEVELS];
    float2 cs36[36];
};

// Eight workgroups per CU (the wave limit): 16.8 KB of LDS each - the two filter forms share their arrays - and at most 64
// vector registers (57 used, nothing spilled); with 20.3 KB and 73 registers it was seven, and 8.4 ms per dense step against 7.1.
template <bool FMA>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_sift_descriptors_batch(const vslam_point* __restrict__ oriented, const unsigned int* __restrict__ counts,
                                                                 unsigned int cap, const uint8_t* __restrict__ pyr, size_t pframe,
                                                                 SiftBatchGeom g, float* __restrict__ desc, uint8_t* __restrict__ defined) {
    __shared__ SiftShared sh;
    const int f = blockIdx.y;
    const unsigned int n = min(counts[f], cap);
    vslam_point kp_next{};  // the next record is fetched while the current descriptor is computed
    if (blockIdx.x < n) kp_next = oriented[(size_t)f * cap + blockIdx.x];
    for (unsigned int q = blockIdx.x; q < n; q += gridDim.x) {
        const vslam_point kp = kp_next;
        if (q + gridDim.x < n) kp_next = oriented[(size_t)f * cap + q + gridDim.x];
        const int o = __builtin_amdgcn_readfirstlane(kp.octave), level = __builtin_amdgcn_readfirstlane(kp.level);  // one record per workgroup: scalar
        const uint8_t* G = pyr + f * pframe + g.oct_off[o] + (size_t)level * g.rows[o] * g.pitch[o];
        const unsigned int b = (unsigned int)__builtin_amdgcn_readfirstlane(kp.value) / 10u;
        sift_one_keypoint<FMA>(sh, kp, g.cs36[b < 36u ? b : 0u], G, g.pitch[o], g.rows[o], g.cols[o], g.kern[o][level], g.kn[o][level],
                          desc + ((size_t)f * cap + q) * SIFT_DESC, defined ? defined + (size_t)f * cap + q : nullptr);
    }
}

}  // namespace vslam
