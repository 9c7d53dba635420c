// filterKeypoints (Diff_of_Gauss.cpp:301-372) inside the batched, device-resident path
// (params.orient): every frame's keypoint list (vslam_detect_batch_dev with localize = 1) goes
// through the edge test and the orientation histogram without leaving HBM.
//
//   k_edge_flags          one thread per list record: computeEdgeResponse (:79-109) on Sobel
//                         differences formed from the u8 Gaussian level, tr^2/det < 12.1 (:335);
//                         ballot words of the survivors -> ordered survivor index list
//   k_orient_survivors    one workgroup per survivor (grid-stride over the list): unlike the
//                         per-image kernel (kernels_orient.hip.h) there are no dense magnitude /
//                         orientation images - materialising them for a batch costs more than the
//                         whole detection - so the (16+2R)^2 magnitude region the window's blur
//                         reaches is formed in LDS from the Gaussian level (same f32 arithmetic:
//                         exact integer gradients, correctly rounded sqrt), then row pass, symmetric
//                         column pass, histogram and peak mask exactly as in the per-image kernel.
//                         Regions that do not fit the LDS (the 300-tap kernels of octave 3, where
//                         survivors are rare) take the magnitudes tap by tap instead.
// The masks are compacted into SLAM::point{row, col, angle, 0, octave, level} records in list
// order, which is the reference's order (octave, keypoint, bin).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_aux.hip.h"
#include "kernels_orient.hip.h"

namespace vslam {

struct OrientBatchGeom {
    int n_oct;
    int rows[VSLAM_MAX_OCTAVES], cols[VSLAM_MAX_OCTAVES], pitch[VSLAM_MAX_OCTAVES];
    unsigned long long oct_off[VSLAM_MAX_OCTAVES];  // byte offset of the octave in a pyramid frame block
    const float* kern[VSLAM_MAX_OCTAVES][VSLAM_NUM_LEVELS];  // f32 Gaussian taps of sigma = 1.5 * sigma(o, l)
    int kn[VSLAM_MAX_OCTAVES][VSLAM_NUM_LEVELS];
};

// grid = (ceil(cap/256), frames): flags[f][i/64] bit i%64 = record i of frame f passes the edge test.
// Records [begins[f], ends[f]) of frame f (begins == nullptr: from 0).  The list is in octave order and
// complete octave by octave, so the records of octave 0 (four fifths of a 1080p frame's list) are
// tested as soon as they exist, beside the coarse octaves' kernels, and the rest when the list is
// complete: the first launch writes every word (zeros past its range), the second leaves the words
// below its range alone and ORs into the one the two ranges share.
__global__ __launch_bounds__(256) void k_edge_flags(const vslam_point* __restrict__ pts, const unsigned int* __restrict__ begins,
                                                     const unsigned int* __restrict__ ends, unsigned int cap,
                                                     const uint8_t* __restrict__ pyr, size_t pframe, OrientBatchGeom g,
                                                     unsigned long long* __restrict__ flags, size_t fwords) {
    const int f = blockIdx.y;
    const unsigned int i = blockIdx.x * 256 + threadIdx.x;
    const unsigned int begin = begins ? min(begins[f], cap) : 0u, end = min(ends[f], cap);
    bool keep = false;
    if (i >= begin && i < end) {
        const vslam_point kp = pts[(size_t)f * cap + i];
        const int o = kp.octave;
        const uint8_t* G = pyr + f * pframe + g.oct_off[o] + (size_t)kp.level * g.rows[o] * g.pitch[o];
        const float r = 10.0f, threshold = ((r + 1.0f) * (r + 1.0f)) / r;  // :331-332
        keep = edge_response_u8(G, g.pitch[o], g.rows[o], g.cols[o], kp.row, kp.col, kp.padding) < threshold;  // :335
    }
    const unsigned long long w = __ballot(keep);
    const unsigned int word = i >> 6;
    if ((threadIdx.x & 63) == 0 && word < fwords) {
        unsigned long long* dst = flags + (size_t)f * fwords + word;
        if ((word + 1) * 64u <= begin) {
            // below this launch's range: the earlier launch's word
        } else if (word * 64u < begin) {
            if (w) atomicOr(dst, w);  // the word the two ranges share
        } else {
            *dst = w;
        }
    }
}

// cv::magnitude / cv::phase of the level's Sobel gradients at one pixel (processGradients,
// GaussPyramid.cpp:65-104), formed from the u8 Gaussian level like k_level_gradients does.
__device__ __forceinline__ void gradient_at(const uint8_t* __restrict__ G, int gpitch, int rows, int cols, int r, int c, float& x, float& y) {
    x = (float)((int)G[(size_t)r * gpitch + reflect101(c + 1, cols)] - (int)G[(size_t)r * gpitch + reflect101(c - 1, cols)]);
    y = (float)((int)G[(size_t)reflect101(r + 1, rows) * gpitch + c] - (int)G[(size_t)reflect101(r - 1, rows) * gpitch + c]);
}
__device__ __forceinline__ float magnitude_at(const uint8_t* __restrict__ G, int gpitch, int rows, int cols, int r, int c) {
    float x, y;
    gradient_at(G, gpitch, rows, cols, r, c, x, y);
    const float xx = x * x, yy = y * y;
    return sqrt_rn_small_nr(xx + yy);  // correctly rounded f32 square root (6 operations; tools/sqrt_check.hip)
}

// The survivor list of a frame is in list order, i.e. octave by octave and, inside an octave, level by level
// (initialKeypointDetection's loops, Diff_of_Gauss.cpp:264-267).  ranges[f][o] = index of the first survivor whose octave is
// >= o (o = 0 .. n_oct; ranges[f][n_oct] = the survivor count); ranges[f][OR_LEVEL_RANGES + 4 o + j] = the first one at or
// past (octave o, level j + 1), j = 0 .. 3 (j = 3: the next octave's first).  grid = (frames), 128 threads, one boundary
// each.  One launch of k_orient_survivors per octave - of k_orient_survivors_pk per octave and level - then takes exactly
// its own survivors with the LDS budget of that octave / level (round 2 ran two launches that each walked the whole list
// and skipped the other's octaves - two dependent loads per skipped survivor - and gave octave 0 the LDS footprint of
// octave 1: 3 workgroups per CU instead of 8).
constexpr int OR_LEVEL_RANGES = VSLAM_MAX_OCTAVES + 1;
constexpr int OR_RANGE_STRIDE = OR_LEVEL_RANGES + 4 * VSLAM_MAX_OCTAVES;
__global__ __launch_bounds__(128) void k_survivor_ranges(const vslam_point* __restrict__ pts, unsigned int cap, const unsigned int* __restrict__ surv,
                                                          const unsigned int* __restrict__ scounts, unsigned int scap, int n_oct,
                                                          unsigned int* __restrict__ ranges) {
    const int f = blockIdx.x, t = threadIdx.x;
    int ko, kl;  // the boundary's key
    if (t <= n_oct) {
        ko = t, kl = 0;
    } else if (t >= OR_LEVEL_RANGES && t - OR_LEVEL_RANGES < 4 * n_oct) {
        ko = (t - OR_LEVEL_RANGES) >> 2, kl = ((t - OR_LEVEL_RANGES) & 3) + 1;
    } else {
        return;
    }
    const unsigned int ns = min(scounts[f], scap);
    unsigned int lo = 0, hi = ns;  // first index with (octave, level) >= (ko, kl)
    while (lo < hi) {
        const unsigned int mid = (lo + hi) >> 1;
        const vslam_point q = pts[(size_t)f * cap + surv[(size_t)f * scap + mid]];
        if (q.octave > ko || (q.octave == ko && q.level >= kl))
            hi = mid;
        else
            lo = mid + 1;
    }
    ranges[(size_t)f * OR_RANGE_STRIDE + t] = lo;
}

// grid = (G, frames), 256 threads, dynamic LDS = lds_floats * 4 bytes (row / column maps + taps + strip + region + patch);
// the survivors of octave `oct` (ranges, k_survivor_ranges).  FMA: the filter's multiply-adds fused (mad_f32, kernels_aux.hip.h).
template <bool FMA>
__global__ __launch_bounds__(256) void k_orient_survivors(const vslam_point* __restrict__ pts, unsigned int cap,
                                                           const unsigned int* __restrict__ surv, const unsigned int* __restrict__ ranges,
                                                           unsigned int scap, const uint8_t* __restrict__ pyr, size_t pframe,
                                                           OrientBatchGeom g, int lds_floats, int oct,
                                                           unsigned long long* __restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) float orient_smem[];
    __shared__ float mw[OR_WIN * OR_WIN];
    __shared__ unsigned long long binmask[OR_BINS][4];  // per bin, per wave: which of the wave's 64 pixels fall into it
    const int f = blockIdx.y;
    const unsigned int k_begin = ranges[(size_t)f * OR_RANGE_STRIDE + oct], k_end = ranges[(size_t)f * OR_RANGE_STRIDE + oct + 1];
    // the next survivor's record is fetched while the current one is processed (two dependent loads: index, record)
    vslam_point kp_next{};
    if (k_begin + blockIdx.x < k_end) kp_next = pts[(size_t)f * cap + surv[(size_t)f * scap + k_begin + blockIdx.x]];
    for (unsigned int k = k_begin + blockIdx.x; k < k_end; k += gridDim.x) {
        const vslam_point kp = kp_next;
        if (k + gridDim.x < k_end) kp_next = pts[(size_t)f * cap + surv[(size_t)f * scap + k + gridDim.x]];
        const int o = oct, level = kp.level, x = kp.col, y = kp.row;
        const int rows = g.rows[o], cols = g.cols[o], gpitch = g.pitch[o];
        const uint8_t* __restrict__ G = pyr + f * pframe + g.oct_off[o] + (size_t)level * rows * gpitch;
        const int kn = g.kn[o][level], R = kn >> 1;
        const float* __restrict__ kt = g.kern[o][level];
        const int prows = rows + 2 * OR_PAD, pcols = cols + 2 * OR_PAD;
        const int span = OR_WIN + 2 * R;
        // LDS: [span] row map, [span] column map, [span >= kn] taps, [span][16] strip, then the magnitude region if it fits.
        // The taps go through LDS: read from global memory inside the filter loops (round 2) every multiply waited
        // for a load round trip (global_load + s_waitcnt vmcnt(0) per tap: the stage was latency-bound on them).
        int* ry = reinterpret_cast<int*>(orient_smem);
        int* cx = ry + span;
        float* kl = orient_smem + 2 * span;
        float* rb = orient_smem + 3 * span;
        float* M = rb + span * OR_WIN;
        const bool region = 3 * span + span * OR_WIN + span * span <= lds_floats;
        __syncthreads();  // the previous survivor's reads of the LDS are done
        for (int i = threadIdx.x; i < span; i += 256) {
            ry[i] = clampi(reflect101(y + i - R, prows) - OR_PAD, 0, rows - 1);  // parent reflect-101, then padOctave's replicate
            cx[i] = clampi(reflect101(x + i - R, pcols) - OR_PAD, 0, cols - 1);
            if (i < kn) kl[i] = kt[i];
        }
        __syncthreads();
        // Interior survivors (no reflection / replication anywhere in the region or its Sobel neighbours - all but a
        // border strip of the image): the (span + 2)^2 pixels the region's gradients read are staged once with
        // aligned dword loads and the magnitudes are formed from LDS bytes: 4 byte reads from LDS per magnitude
        // instead of 4 byte loads from global memory (the region phase was ~45 % of the stage).
        const int pdw = (span + 2 + 6) >> 2;  // dwords per patch row: span + 2 bytes at any alignment
        uint32_t* Pw = reinterpret_cast<uint32_t*>(M + span * span);
        const int py0 = y - R - OR_PAD - 1, px0 = x - R - OR_PAD - 1;  // image coordinates of the patch origin
        const bool patch = region && py0 >= 0 && px0 >= 0 && py0 + span + 2 <= rows && px0 + span + 2 <= cols &&
                           3 * span + span * OR_WIN + span * span + (span + 2) * pdw <= lds_floats;
        if (patch) {
            const int a0 = px0 & ~3, sh = px0 - a0;
            for (int it = threadIdx.x; it < (span + 2) * pdw; it += 256) {
                const int pr = it / pdw, k = it - pr * pdw;
                Pw[it] = *reinterpret_cast<const uint32_t*>(G + (size_t)(py0 + pr) * gpitch + a0 + 4 * k);  // planes are 16-byte aligned, pitch % 16 == 0
            }
            __syncthreads();
            const uint8_t* Pb = reinterpret_cast<const uint8_t*>(Pw) + sh;
            const int pb = 4 * pdw;
            // element it = rr * span + cc; it += 256 moves (rr, cc) by (256 / span, 256 % span) with at most one wrap:
            // one division per survivor instead of one per element
            const int dq = 256 / span, dr = 256 - dq * span;
            int rr = (int)threadIdx.x / span, cc = (int)threadIdx.x - rr * span;
            int off = (rr + 1) * pb + (cc + 1);  // the region pixel inside the patch
            for (int it = threadIdx.x; it < span * span; it += 256) {
                const uint8_t* c0 = Pb + off;
                const float gx = (float)((int)c0[1] - (int)c0[-1]), gy = (float)((int)c0[pb] - (int)c0[-pb]);
                const float xx = gx * gx, yy = gy * gy;
                M[it] = sqrt_rn_small_nr(xx + yy);
                cc += dr;
                const bool wrap = cc >= span;
                cc -= wrap ? span : 0;
                off += dq * pb + dr + (wrap ? pb - span : 0);
            }
            __syncthreads();
        } else if (region) {
            for (int it = threadIdx.x; it < span * span; it += 256) {
                const int rr = it / span, cc = it - rr * span;
                M[it] = magnitude_at(G, gpitch, rows, cols, ry[rr], cx[cc]);
            }
            __syncthreads();
        }
        if (region) {
            for (int it = threadIdx.x; it < span * OR_WIN; it += 256) {
                const int rr = it >> 4, c = it & 15;
                const float* __restrict__ S = M + rr * span + c;
                float s0 = kl[0] * S[0];
#pragma unroll 4
                for (int i = 1; i < kn; ++i) s0 = mad_f32<FMA>(kl[i], S[i], s0);
                rb[it] = s0;
            }
        } else {
            for (int it = threadIdx.x; it < span * OR_WIN; it += 256) {
                const int rr = it >> 4, c = it & 15;
                float s0 = kl[0] * magnitude_at(G, gpitch, rows, cols, ry[rr], cx[c]);
                for (int i = 1; i < kn; ++i) s0 = mad_f32<FMA>(kl[i], magnitude_at(G, gpitch, rows, cols, ry[rr], cx[c + i]), s0);
                rb[it] = s0;
            }
        }
        __syncthreads();
        int bin_of;
        {
            const int i = threadIdx.x >> 4, j = threadIdx.x & 15;
            float s0 = kl[R] * rb[(i + R) * OR_WIN + j];
#pragma unroll 4
            for (int t = 1; t <= R; ++t) s0 = mad_f32<FMA>(kl[R + t], rb[(i + R + t) * OR_WIN + j] + rb[(i + R - t) * OR_WIN + j], s0);
            mw[threadIdx.x] = s0;
            float gx, gy;
            if (patch) {  // interior survivor: the window's pixels and their Sobel neighbours are in the staged patch
                const uint8_t* c0 = reinterpret_cast<const uint8_t*>(Pw) + (px0 & 3) + (i + R + 1) * (4 * pdw) + (j + R + 1);
                gx = (float)((int)c0[1] - (int)c0[-1]), gy = (float)((int)c0[4 * pdw] - (int)c0[-4 * pdw]);
            } else {
                gradient_at(G, gpitch, rows, cols, clampi(y + i - OR_PAD, 0, rows - 1), clampi(x + j - OR_PAD, 0, cols - 1), gx, gy);
            }
            const float reductionCoeff = (float)OR_BINS / 360.0f;            // :114
            const int index = (int)(fast_atan2_deg(gy, gx) * reductionCoeff);  // :126
            bin_of = min(max(index, 0), OR_BINS - 1);
        }
        // The histogram (:112-133) adds a bin's magnitudes in pixel order; a lane per bin walking all 256 pixels
        // (round 2) kept one wave busy for 256 dependent iterations while three waited.  Each wave now ballots
        // its 64 pixels bin by bin (ascending pixel index inside a mask = the reference's order), and the bin's
        // lane adds only its own pixels: a handful instead of 256.
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
            // the 36 sums sit in lanes 0..35 of this wave: maximum and peak test without another pass through LDS
            float mx = threadIdx.x < OR_BINS ? h : 0.0f;  // sums of non-negative weights: 0 is neutral, and bin 0 is among them
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            const float peakThreshold = mx * 0.8f;                          // :358
            const bool peak = threadIdx.x < OR_BINS && h > peakThreshold;  // :362
            const unsigned long long m = __ballot(peak);
            if (threadIdx.x == 0) masks[(size_t)f * scap + k] = m;
        }
    }
}

}  // namespace vslam
