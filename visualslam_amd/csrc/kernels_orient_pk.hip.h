// filterKeypoints' orientation histogram (Diff_of_Gauss.cpp:338-366, orientationHistogram :112-133) for the survivors of
// the FINE octaves of a batch, where four fifths of a frame's keypoints are: k_orient_survivors_pk.
//
// Same arithmetic, bit for bit, as k_orient_survivors (kernels_orient_batch.hip.h) - exact integer gradients, correctly
// rounded square root, the row filter's products and sums rounded one by one left to right, the symmetric column
// filter, a bin's magnitudes added in pixel order - but organised for the issue slots, which is what that kernel was
// bound by (3.9 k vector instructions per survivor at 82 % of the issue rate):
//   * two waves per survivor instead of four, four barriers instead of seven, the next survivor's patch staged by wave 1
//     while wave 0 adds up the previous survivor's histogram;
//   * magnitudes: one item = 2 rows x 4 columns from eight LDS dwords; every byte is converted once per item (20
//     conversions for 8 values instead of 32), differences / squares / the Newton step of the square root in packed
//     f32 (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two values per instruction - NOT twice the rate on this chip, 4.3
//     cycles against 2.3 for the one-value forms, but half the instructions, LDS words and loop trips), the pair (row 2p,
//     row 2p+1) of a column is what one packed register holds and what one 8-byte LDS word of the magnitude region holds;
//   * row pass: one item = the same row pair x 4 output columns, a sliding window of 8-byte words, v_pk_mul_f32 by a
//     tap held in a SCALAR register (op_sel splat: no LDS read, no vector register per tap) + v_pk_add_f32: one
//     instruction per product-and-sum of two rows, where the round-3 form took 2.5 and two LDS reads;
//   * column pass: two adjacent columns per thread, the same way;
//   * one launch per (octave, level): the span, the tap count and the LDS layout are the launch's constants and the
//     footprint is the level's own (11 / 14 / 18 KB for octave 0's three levels: room for 14 / 11 / 9 workgroups per CU; the register cap below
//     makes it 12 / 11 / 9).
// A zero tap adds +0 to a non-negative sum, which leaves every bit of it alone: the tap arrays are zero-padded (three in
// front for the sliding window, to a multiple of four behind), steps that would read past a row are skipped.
//
// Survivors whose region or its Sobel neighbours touch the image border (a strip of R + 9 pixels) take their magnitudes
// and window gradients pixel by pixel from the image with the reference's border rules, then the same passes.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_orient_batch.hip.h"

namespace vslam {

typedef const vslam_f4 __attribute__((address_space(4)))* orient_taps4_p;  // uniform address => s_load_dwordx4

// result = k.lo * w (SEL 0) or k.hi * w (SEL 1), both halves; k in a scalar register pair
template <int SEL>
__device__ __forceinline__ vslam_f2 pk_mul_splat(vslam_f2 k, vslam_f2 w) {
    vslam_f2 r;
    if constexpr (SEL == 0)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]" : "=v"(r) : "s"(k), "v"(w));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(r) : "s"(k), "v"(w));
    return r;
}
// tap E (0..7) of the eight scalar taps (a, b)
template <int E>
__device__ __forceinline__ vslam_f2 tap_mul(const vslam_f4& a, const vslam_f4& b, vslam_f2 w) {
    static_assert(E >= 0 && E < 8, "tap index");
    if constexpr (E < 2) return pk_mul_splat<E & 1>(vslam_f2{a.x, a.y}, w);
    else if constexpr (E < 4) return pk_mul_splat<E & 1>(vslam_f2{a.z, a.w}, w);
    else if constexpr (E < 6) return pk_mul_splat<E & 1>(vslam_f2{b.x, b.y}, w);
    else return pk_mul_splat<E & 1>(vslam_f2{b.z, b.w}, w);
}
// acc + tap E * w (FMA = false: v_pk_mul_f32 + v_pk_add_f32, each rounded) or fma(tap E, w, acc) (FMA = true: one v_pk_fma_f32)
template <int SEL>
__device__ __forceinline__ vslam_f2 pk_fma_splat(vslam_f2 k, vslam_f2 w, vslam_f2 acc) {
    vslam_f2 r;
    if constexpr (SEL == 0)
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "s"(k), "v"(w), "v"(acc));
    else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "s"(k), "v"(w), "v"(acc));
    return r;
}
template <int E, bool FMA>
__device__ __forceinline__ vslam_f2 tap_mac(const vslam_f4& a, const vslam_f4& b, vslam_f2 w, vslam_f2 acc) {
    if constexpr (!FMA) return acc + tap_mul<E>(a, b, w);
    else if constexpr (E < 2) return pk_fma_splat<E & 1>(vslam_f2{a.x, a.y}, w, acc);
    else if constexpr (E < 4) return pk_fma_splat<E & 1>(vslam_f2{a.z, a.w}, w, acc);
    else if constexpr (E < 6) return pk_fma_splat<E & 1>(vslam_f2{b.x, b.y}, w, acc);
    else return pk_fma_splat<E & 1>(vslam_f2{b.z, b.w}, w, acc);
}
// {x.hi - y.lo, y.hi - x.lo}: the vertical differences of a row pair from {row, row + 1} and {row - 1, row + 2}
__device__ __forceinline__ vslam_f2 pk_cross_diff(vslam_f2 x, vslam_f2 y) {
    vslam_f2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(r) : "v"(x), "v"(y));
    return r;
}
__device__ __forceinline__ float ubyte_f32(uint32_t w, int b) { return (float)((w >> (8 * b)) & 0xffu); }

// Lane B of the wave gets the 64-bit mask of the lanes whose bin is B (bins 0..63) from the six ballots of the bin
// index's bit planes: mask = AND_k (plane_k XOR (bit k of B ? 0 : ~0)).  12 + 22 instructions for the wave, where a
// compare and two v_writelane per bin (bin_masks_to_lanes, kernels_orient.hip.h) took 108.  nsel[k] = the lane's own
// (bit k of lane ? 0 : ~0), formed once per kernel.
__device__ __forceinline__ void bin_masks_from_planes(int mybin, const unsigned int (&nsel)[6], unsigned int& lo, unsigned int& hi) {
    lo = hi = ~0u;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const unsigned long long plane = __ballot((mybin >> k) & 1);
        lo &= (unsigned int)plane ^ nsel[k];
        hi &= (unsigned int)(plane >> 32) ^ nsel[k];
    }
}

// Device layout of the padded taps of one (octave, level) (written by get_orient_taps, vslam_hip.hip):
//   [0, n)            the taps themselves (k_orient_survivors, SIFT)
//   row  = A .. :     0 0 0 k[0] .. k[n-1] 0 0 ..    A = (n + 3) & ~3, length NR = ((n + 6) & ~3) + 8
//   col  = A + NR .. : k[R+1] .. k[2R] 0 0 ..         length ((R + 3) & ~3) + 4   (k[R], the centre tap, is row[3 + R])
__host__ __device__ inline int orient_taps_row_off(int n) { return (n + 3) & ~3; }
__host__ __device__ inline int orient_taps_col_off(int n) { return orient_taps_row_off(n) + ((n + 6) & ~3) + 8; }
__host__ __device__ inline int orient_taps_pk_floats(int n) { return orient_taps_col_off(n) + (((n >> 1) + 3) & ~3) + 4; }

constexpr int OR_PK_MAX_SPAN = 60;  // 16 four-column groups cover byte sh + span of a patch row (sh <= 3)
// LDS floats of a launch for one level's span (even, <= OR_PK_MAX_SPAN): magnitude region as row pairs (the window's
// weights take its first KB once the row pass has read it), patch / strip overlay, bin masks.  11 / 14 / 18 KB for the three
// levels of octave 0 = room for 14 / 11 / 9 workgroups per CU, where one launch per octave (the widest level's footprint) had 8.
__host__ __device__ inline int orient_pk_mp(int span) { return span + 7; }                     // 8-byte words per row pair: the row pass reads up to column span + 6; odd: bank spread
__host__ __device__ inline int orient_pk_pp(int span) { return (((span + 8) >> 2) + 2) | 1; }  // dwords per patch row, one pad dword either side
__host__ __device__ inline int orient_pk_m2_floats(int span) { return ((span >> 1) * orient_pk_mp(span) * 2 + 3) & ~3; }  // what follows stays 16-byte aligned
__host__ __device__ inline int orient_pk_lds_floats(int span) {
    const int m2 = orient_pk_m2_floats(span), patch = (span + 2) * orient_pk_pp(span), strip = span * OR_WIN;
    return m2 + (patch > strip ? patch : strip) + OR_BINS * 4 * 2;
}

// grid = (G, frames), 128 threads, dynamic LDS = orient_pk_lds_floats(span of the level) * 4 bytes; the survivors of
// (octave `oct`, level `level`) (k_survivor_ranges).  KN = the level's tap count as a compile-time constant (the three of
// the default pyramid's octave 0 are instantiated: loop bounds, guards and LDS offsets fold), 0 = read it from g.
// At least six waves per SIMD (<= 85 vector registers): the unrolled instantiations otherwise take 90 - 124 and the registers, not
// the LDS, would bound the workgroups per CU (28.7 -> 28.2 ms per dense step).
template <int KN, bool FMA>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_orient_survivors_pk(const vslam_point* __restrict__ pts, unsigned int cap,
                                                              const unsigned int* __restrict__ surv, const unsigned int* __restrict__ ranges,
                                                              unsigned int scap, const uint8_t* __restrict__ pyr, size_t pframe,
                                                              OrientBatchGeom g, int oct, int level, unsigned long long* __restrict__ masks) {
    extern __shared__ __attribute__((aligned(16))) float orient_smem[];
    const int o = oct;
    const int kn = KN ? KN : g.kn[o][level], R = kn >> 1, span = OR_WIN + 2 * R, hs = span >> 1, pdw = (span + 2 + 6) >> 2;
    const int MP = orient_pk_mp(span), PP = orient_pk_pp(span);
    const int GQ = ((span + 3) >> 2) + 1, gq_inv = 65536 / GQ + 1;
    vslam_f2* const M2 = reinterpret_cast<vslam_f2*>(orient_smem);                 // [span / 2][MP] {row 2p, row 2p + 1}
    float* const mw = orient_smem;                                                 // [16][16] blurred magnitudes of the window: written when the region has been read
    float* const ov = orient_smem + orient_pk_m2_floats(span);                     // patch, then the row-filtered strip
    uint32_t* const Pw = reinterpret_cast<uint32_t*>(ov);                          // [span + 2][PP]
    float* const rb = ov;                                                          // [span][16]
    const int ovf = max((span + 2) * PP, span * OR_WIN);
    unsigned long long* const binmask = reinterpret_cast<unsigned long long*>(ov + ovf);  // [36][4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int f = blockIdx.y;
    const unsigned int k_begin = ranges[(size_t)f * OR_RANGE_STRIDE + OR_LEVEL_RANGES + 4 * oct + level - 1],
                       k_end = ranges[(size_t)f * OR_RANGE_STRIDE + OR_LEVEL_RANGES + 4 * oct + level];
    const int rows = g.rows[o], cols = g.cols[o], gpitch = g.pitch[o];
    const int prows = rows + 2 * OR_PAD, pcols = cols + 2 * OR_PAD;
    const uint8_t* __restrict__ G = pyr + f * pframe + g.oct_off[o] + (size_t)level * rows * gpitch;
    const float* __restrict__ kt = g.kern[o][level];
    unsigned int nsel[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) nsel[b] = ((lane >> b) & 1) ? 0u : ~0u;
    // The patch of a survivor is fetched one survivor ahead, into registers of wave 1 (lane = patch row): its load round
    // trip - a good part of a survivor's whole time - runs beside the previous survivor's passes.  So the records are
    // fetched two ahead.
    struct Geo {
        int x, y, py0, px0, sh;
        bool interior;
    };
    auto geo_of = [&](const vslam_point& r) {
        Geo q;
        q.x = __builtin_amdgcn_readfirstlane(r.col), q.y = __builtin_amdgcn_readfirstlane(r.row);  // the record is the same in every lane
        q.py0 = q.y - R - OR_PAD - 1, q.px0 = q.x - R - OR_PAD - 1;  // image coordinates of the patch origin
        q.interior = q.py0 >= 0 && q.px0 >= 0 && q.py0 + span + 2 <= rows && q.px0 + span + 2 <= cols;
        q.sh = q.interior ? (q.px0 & 3) : 0;  // region column cc is column cc + sh + 1 of M2 and byte cc + sh + 1 of a patch row
        return q;
    };
    constexpr int PDW_MAX = (OR_PK_MAX_SPAN + 8) >> 2;
    uint32_t pre[PDW_MAX];
    auto fetch_patch = [&](const Geo& q) {  // wave 1, lane = patch row
        if (q.interior && lane < span + 2) {
            const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(G + (size_t)(q.py0 + lane) * gpitch + (q.px0 - q.sh));  // pitch % 16 == 0
#pragma unroll
            for (int d = 0; d < PDW_MAX; ++d)
                if (d < pdw) pre[d] = src[d];
        }
    };
    unsigned int k = k_begin + blockIdx.x;
    vslam_point rec0{}, rec1{};
    if (k < k_end) rec0 = pts[(size_t)f * cap + surv[(size_t)f * scap + k]];
    if (k + gridDim.x < k_end) rec1 = pts[(size_t)f * cap + surv[(size_t)f * scap + k + gridDim.x]];
    if (wave == 1 && k < k_end) fetch_patch(geo_of(rec0));
    bool has_prev = false;
    unsigned int k_prev = 0;
    for (;; k += gridDim.x) {
        const bool have = k < k_end;
        const vslam_point kp = rec0;
        rec0 = rec1;
        if (k + 2 * gridDim.x < k_end) rec1 = pts[(size_t)f * cap + surv[(size_t)f * scap + k + 2 * gridDim.x]];
        const Geo cur = geo_of(kp);
        const int x = cur.x, y = cur.y, sh = cur.sh;
        const bool interior = cur.interior;
        // ---- [A] wave 1: this survivor's patch from its registers to the LDS, the next one's loads; wave 0: the previous
        // survivor's histogram and peaks
        if (wave == 1) {
            if (have && interior && lane < span + 2) {
                uint32_t* dst = Pw + lane * PP + 1;
#pragma unroll
                for (int d = 0; d < PDW_MAX; ++d)
                    if (d < pdw) dst[d] = pre[d];
            }
            if (k + gridDim.x < k_end) fetch_patch(geo_of(rec0));
        } else if (has_prev) {
            float h = 0.0f;
            if (lane < OR_BINS) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    unsigned long long m = binmask[lane * 4 + w];
                    while (m) {
                        h += mw[64 * w + __builtin_ctzll(m)];
                        m &= m - 1;
                    }
                }
            }
            float mx = lane < OR_BINS ? h : 0.0f;  // sums of non-negative weights: 0 is neutral
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            const float peakThreshold = mx * 0.8f;                 // :358
            const bool peak = lane < OR_BINS && h > peakThreshold;  // :362
            const unsigned long long m = __ballot(peak);
            if (lane == 0) masks[(size_t)f * scap + k_prev] = m;
        }
        if (!have) break;
        __syncthreads();
        // ---- [B] magnitudes of the region, and the histogram bin of every window pixel
        if (interior) {
            // GQ four-column groups reach byte span + 3 of a patch row (sh <= 3); it = rp * GQ + gq, the division by the launch's
            // constant as a multiplication (it < 2^10, GQ <= 16: exact)
            for (int it = tid; it < hs * GQ; it += 128) {
                const int rp = (it * gq_inv) >> 16, gq = it - rp * GQ;
                if (4 * gq > sh + span) continue;  // columns past the region
                const uint32_t* P = Pw + (2 * rp) * PP + gq;  // dword gq - 1 of patch row 2 rp (one pad dword in front)
                const uint32_t A = P[1], Bl = P[PP], Bm = P[PP + 1], Br = P[PP + 2], Cl = P[2 * PP], Cm = P[2 * PP + 1], Cr = P[2 * PP + 2], D = P[3 * PP + 1];
                // FBC[e] = bytes 4 gq - 1 + e of rows 2 rp + 1 (region row 2 rp) and 2 rp + 2; FAD[t] = bytes 4 gq + t of the rows above / below
                vslam_f2 FBC[6], FAD[4];
                FBC[0] = vslam_f2{ubyte_f32(Bl, 3), ubyte_f32(Cl, 3)};
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    FBC[t + 1] = vslam_f2{ubyte_f32(Bm, t), ubyte_f32(Cm, t)};
                    FAD[t] = vslam_f2{ubyte_f32(A, t), ubyte_f32(D, t)};
                }
                FBC[5] = vslam_f2{ubyte_f32(Br, 0), ubyte_f32(Cr, 0)};
                vslam_f2 mg[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const vslam_f2 gx = FBC[t + 2] - FBC[t];
                    const vslam_f2 gy = pk_cross_diff(FBC[t + 1], FAD[t]);
                    const vslam_f2 xx = gx * gx;
                    mg[t] = sqrt_rn_small_pk(__builtin_elementwise_fma(gy, gy, xx));  // integers below 2^18: the sum is exact either way
                }
                vslam_f2* dst = M2 + rp * MP + 4 * gq;
                dst[0] = mg[0], dst[1] = mg[1], dst[2] = mg[2], dst[3] = mg[3];
            }
        } else {
            for (int it = tid; it < span * span; it += 128) {
                const int rr = it / span, cc = it - rr * span;
                const int sy = clampi(reflect101(y + rr - R, prows) - OR_PAD, 0, rows - 1);  // parent reflect-101, then padOctave's replicate
                const int sx = clampi(reflect101(x + cc - R, pcols) - OR_PAD, 0, cols - 1);
                reinterpret_cast<float*>(M2 + (rr >> 1) * MP + cc + 1)[rr & 1] = magnitude_at(G, gpitch, rows, cols, sy, sx);
            }
        }
        {
            const uint8_t* Pb = reinterpret_cast<const uint8_t*>(Pw + 1) + sh;
            const int pb = 4 * PP;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int p = tid + 128 * half, i = p >> 4, j = p & 15;
                float gx, gy;
                if (interior) {
                    const uint8_t* c0 = Pb + (i + R + 1) * pb + (j + R + 1);
                    gx = (float)((int)c0[1] - (int)c0[-1]), gy = (float)((int)c0[pb] - (int)c0[-pb]);
                } else {
                    gradient_at(G, gpitch, rows, cols, clampi(y + i - OR_PAD, 0, rows - 1), clampi(x + j - OR_PAD, 0, cols - 1), gx, gy);
                }
                const float reductionCoeff = (float)OR_BINS / 360.0f;            // :114
                const int index = (int)(fast_atan2_deg(gy, gx) * reductionCoeff);  // :126
                const int bin_of = min(max(index, 0), OR_BINS - 1);
                unsigned int mlo, mhi;
                bin_masks_from_planes(bin_of, nsel, mlo, mhi);
                if (lane < OR_BINS) binmask[lane * 4 + wave + 2 * half] = ((unsigned long long)mhi << 32) | mlo;  // pixels 64 (wave + 2 half) ..
            }
        }
        __syncthreads();
        // ---- [C] row pass: rb[rr][c] = sum_i k[i] * M[rr][c + i], products and sums rounded one by one, i ascending
        {
            const orient_taps4_p kr4 = (orient_taps4_p)(kt + orient_taps_row_off(kn));
            const int nsteps = kn + 3;
            for (int it = tid; it < hs * 4; it += 128) {
                const int cg = (it >= hs) + (it >= 2 * hs) + (it >= 3 * hs), rp = it - cg * hs;
                const vslam_f2* __restrict__ wb = M2 + rp * MP + 4 * cg + sh + 1;
                vslam_f2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
                for (int w0 = 0; w0 < nsteps; w0 += 4) {
                    const vslam_f4 ta = kr4[w0 >> 2], tb = kr4[(w0 >> 2) + 1];  // row[w0 .. w0 + 7]; column c at step w uses row[w - c + 3]
                    // all four words of the chunk at once (inside the region's allocation whatever nsteps is: column < MP);
                    // the ones past the row are not used
                    const vslam_f2 W0 = wb[w0], W1 = wb[w0 + 1], W2 = wb[w0 + 2], W3 = wb[w0 + 3];
                    // (a zero tap contributes +0 to a sum that never is -0 - the same value with or without fusing; the first
                    // non-zero tap of an accumulator lands on 0: k * w + 0 = fma(k, w, 0) = the rounded product, as the reference's s0 = k[0] * S[0])
                    a0 = tap_mac<3, FMA>(ta, tb, W0, a0), a1 = tap_mac<2, FMA>(ta, tb, W0, a1), a2 = tap_mac<1, FMA>(ta, tb, W0, a2), a3 = tap_mac<0, FMA>(ta, tb, W0, a3);
                    if (w0 + 1 < nsteps) a0 = tap_mac<4, FMA>(ta, tb, W1, a0), a1 = tap_mac<3, FMA>(ta, tb, W1, a1), a2 = tap_mac<2, FMA>(ta, tb, W1, a2), a3 = tap_mac<1, FMA>(ta, tb, W1, a3);
                    if (w0 + 2 < nsteps) a0 = tap_mac<5, FMA>(ta, tb, W2, a0), a1 = tap_mac<4, FMA>(ta, tb, W2, a1), a2 = tap_mac<3, FMA>(ta, tb, W2, a2), a3 = tap_mac<2, FMA>(ta, tb, W2, a3);
                    if (w0 + 3 < nsteps) a0 = tap_mac<6, FMA>(ta, tb, W3, a0), a1 = tap_mac<5, FMA>(ta, tb, W3, a1), a2 = tap_mac<4, FMA>(ta, tb, W3, a2), a3 = tap_mac<3, FMA>(ta, tb, W3, a3);
                }
                float* d0 = rb + (2 * rp) * OR_WIN + 4 * cg;
                *reinterpret_cast<vslam_f4*>(d0) = vslam_f4{a0.x, a1.x, a2.x, a3.x};
                *reinterpret_cast<vslam_f4*>(d0 + OR_WIN) = vslam_f4{a0.y, a1.y, a2.y, a3.y};
            }
        }
        __syncthreads();
        // ---- [D] symmetric column pass: mw[i][j] = k[R] rb[i + R][j] + sum_t k[R + t] (rb[i + R + t][j] + rb[i + R - t][j])
        {
            const int jp = tid & 7, i = tid >> 3;
            const vslam_f2* __restrict__ rc = reinterpret_cast<const vslam_f2*>(rb + (i + R) * OR_WIN + 2 * jp);
            const float kc = kt[R];
            vslam_f2 acc = rc[0] * kc;
            const orient_taps4_p kc4 = (orient_taps4_p)(kt + orient_taps_col_off(kn));
            for (int t0 = 1; t0 <= R; t0 += 4) {
                const vslam_f4 ta = kc4[(t0 - 1) >> 2];  // k[R + t0 .. R + t0 + 3]
                {
                    const vslam_f2 s2 = rc[t0 * 8] + rc[-t0 * 8];
                    acc = tap_mac<0, FMA>(ta, ta, s2, acc);
                }
                if (t0 + 1 <= R) {
                    const vslam_f2 s2 = rc[(t0 + 1) * 8] + rc[-(t0 + 1) * 8];
                    acc = tap_mac<1, FMA>(ta, ta, s2, acc);
                }
                if (t0 + 2 <= R) {
                    const vslam_f2 s2 = rc[(t0 + 2) * 8] + rc[-(t0 + 2) * 8];
                    acc = tap_mac<2, FMA>(ta, ta, s2, acc);
                }
                if (t0 + 3 <= R) {
                    const vslam_f2 s2 = rc[(t0 + 3) * 8] + rc[-(t0 + 3) * 8];
                    acc = tap_mac<3, FMA>(ta, ta, s2, acc);
                }
            }
            *reinterpret_cast<vslam_f2*>(mw + i * OR_WIN + 2 * jp) = acc;
        }
        __syncthreads();
        has_prev = true;
        k_prev = k;
    }
}

}  // namespace vslam
