// K-H: the whole Harris executable's per-pixel work in ONE pass over the frame (gfx950).
//
//   GaussianBlur 3x3 (Harris_corners.cpp:158) -> Sobel x/y ksize 1 (:163-164) -> replicate-
//   padded 3x3 structure tensor + response (HarrisCorner, :31-68) -> convertScaleAbs (:176) +
//   NonMaximumSuppression window 3 (:70-81) -> NMS2 window 5 (:83-129) -> keypoint test (:139)
//
// Structure: a WAVE owns a vertical strip.  Each lane owns 4 horizontally adjacent pixels (one
// dword of the 8-bit frame, so a wave row is a 256-byte coalesced load); rows stream top to
// bottom through registers (rolling 3-row / 4-row state per stage), and the only cross-lane
// traffic is the +-1 lane exchange of stage edges with DPP wave shifts (wave_shr:1 / wave_shl:1,
// folded into the consuming v_add_u32 / v_max_u32 where the consumer is a VOP2).  Nothing goes through LDS or back through HBM: the frame is read
// once (+ 4 % strip overlap) and response / mask / NMS2 map are written once, as 16-byte and
// 4-byte per-lane stores.  Keypoint flags leave as wave ballots (one 64-bit word per pixel
// slot k = 0..3 of the strip row), compacted in row-major order by kernels_compact.hip.h.
//
// Lanes 0,1 and 62,63 of a wave only feed their neighbours (each stage invalidates one more
// edge pixel), so a strip produces 60 lanes x 4 = 240 output columns; strips start every 240.
// ANYW = false requires cols % 4 == 0 (every row dword-aligned, the image ends on a lane
// boundary); ANYW = true takes any width: unaligned 4/16-byte accesses for the lanes that lie
// fully inside, per-pixel loads, replicate-edge selection and stores for the one lane that
// straddles the right edge.
//
// Exactness: all pre-response quantities are integers (16-bit lanes for
// the blur and the gradients, int32 for products and 3x3 sums < 2^24); det is formed exactly
// with one v_fma_f64 and rounded once to f32; the three f32 ops of :57 stay separate
// (-ffp-contract=off).  The 8-bit view of convertScaleAbs (:176) is NOT monotone on x86-64 --
// responses >= 2^31 read 0 (cvRound returns INT_MIN, kernels_generic.hip.h: cvt_abs_u8) -- so
// every response is converted once and the 3x3 NonMaximumSuppression runs on the converted
// values; NMS2 runs on the raw f32 response (:179) and its survivors are keypoints when their
// 8-bit view exceeds 253 (:139,181), i.e. for 253.5 <= max < 2^31.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels_generic.hip.h"

namespace vslam {

constexpr int HS_VALID_LANES = 60;
constexpr int HS_STRIP_W = 4 * HS_VALID_LANES;  // 240 output columns per wave strip

typedef short s2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (s2_t)(__builtin_bit_cast(s2_t, a) - __builtin_bit_cast(s2_t, b)));
}
struct HarrisStripArgs {
    const uint8_t* img;
    size_t frame;
    int rows, cols;
    float k;
    float* resp;  // may be null
    uint8_t* mask;
    float* nms2;
    unsigned long long* flags;  // [frame][rows][nstrips][4]
    size_t fframe;
    int nstrips, seg;
    uint8_t* dump;  // 64 writable bytes nobody reads (may be null): where the four margin lanes of a strip "store" in the steady rows
};

typedef unsigned short hs_us2_t __attribute__((ext_vector_type(2)));

// lane-1 / lane+1 value, 0 where the wave has no such lane (bound_ctrl): one v_mov_b32_dpp, and the
// DPP-combine pass folds it into a VOP2 consumer (v_add_u32_dpp, v_max_f32_dpp)
__device__ __forceinline__ uint32_t dpp_left(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t dpp_right(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xf, 0xf, true); }
__device__ __forceinline__ float dpp_left_f(float v) { return __uint_as_float(dpp_left(__float_as_uint(v))); }
__device__ __forceinline__ float dpp_right_f(float v) { return __uint_as_float(dpp_right(__float_as_uint(v))); }
__device__ __forceinline__ uint32_t pk_lshr4_u16(uint32_t a) {
    return __builtin_bit_cast(uint32_t, (hs_us2_t)(__builtin_bit_cast(hs_us2_t, a) >> (hs_us2_t)4));
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }

__device__ __forceinline__ uint32_t umax3(uint32_t a, uint32_t b, uint32_t c) { return max(max(a, b), c); }

// One wave: rows [y_begin, y_end) of one strip.  EDGE = the strip touches the left / right image
// border (or, any-width, ends inside a lane): only then do the replicate-column selects and the
// outside-the-image masking exist at all; interior strips run straight-line code.
//
// Responses are >= +0 and never NaN, so from the response on every maximum / comparison runs on the
// IEEE bit patterns as unsigned integers (same order, no canonicalisation instructions).
template <bool ANYW, bool EDGE>
__device__ __forceinline__ void harris_strip_rows(const HarrisStripArgs& a, const int lane, const int strip, const int y_begin,
                                                  const int y_end) {
    const int rows = a.rows, cols = a.cols;
    const size_t N = (size_t)rows * cols;
    const uint8_t* src = a.img + blockIdx.z * a.frame;
    const int x0 = strip * HS_STRIP_W + 4 * (lane - 2);
    const bool lane_in = x0 >= 0 && x0 < cols;                 // owns at least one image pixel
    const bool lane_full = x0 >= 0 && x0 + 4 <= cols;          // owns four (always, when !ANYW)
    const bool lane_out = lane >= 2 && lane < 62 && lane_in;
    const bool left_edge = x0 == 0, right_edge = x0 + 4 == cols;
    const int jedge = ANYW && lane_in && !lane_full ? cols - x0 : 0;  // 1..3: first pixel slot outside the image
    bool inter[4];  // NMS2 is evaluated for columns [2, cols-2)
#pragma unroll
    for (int k = 0; k < 4; ++k) inter[k] = lane_out && x0 + k >= 2 && x0 + k < cols - 2;
    unsigned long long inter_m[4];  // the same as wave masks: the keypoint words are built on the scalar unit
#pragma unroll
    for (int k = 0; k < 4; ++k) inter_m[k] = __builtin_amdgcn_ballot_w64(inter[k]);

    auto load_row = [&](int t) -> uint32_t {
        const uint8_t* row = src + (size_t)reflect101(t, rows) * cols;
        if (!EDGE) {  // interior strips: every lane lies inside the image; scalar row offset + 32-bit lane offset
            uint32_t w;   // (a frame is far below 4 GB: no 64-bit address arithmetic per row)
            const uint32_t roff = __builtin_amdgcn_readfirstlane((uint32_t)reflect101(t, rows) * (uint32_t)cols);  // scalar multiply
            __builtin_memcpy(&w, src + roff + (uint32_t)x0, 4);  // dword aligned when !ANYW
            return w;
        }
        if (ANYW ? lane_full : lane_in) {
            uint32_t w;
            __builtin_memcpy(&w, row + x0, 4);
            return w;
        }
        return (uint32_t)row[reflect101(x0, cols)] | ((uint32_t)row[reflect101(x0 + 1, cols)] << 8) |
               ((uint32_t)row[reflect101(x0 + 2, cols)] << 16) | ((uint32_t)row[reflect101(x0 + 3, cols)] << 24);
    };

    // rolling state (suffix = row relative to the row being loaded, t)
    uint32_t e1 = 0, o1 = 0, e2 = 0, o2 = 0;          // raw rows t-2, t-1 as 16-bit lanes (p0,p2)/(p1,p3)
    uint32_t be2 = 0, bo2 = 0, be1 = 0, bo1 = 0;      // blurred rows t-3, t-2
    int PA[3][4] = {}, PB[3][4] = {};                 // gradient products (xx, yy, xy) of rows t-4, t-3
    uint32_t h3a[4] = {}, h3b[4] = {};                // 3-max of the 8-bit view, rows y-1, y      (y = t-4)
    uint32_t h4a[4] = {}, h4b[4] = {}, h4c[4] = {};   // 4-max of the response, rows y-2, y-1, y
    uint32_t Ry[4] = {}, Cy[4] = {}, nby[4] = {};     // response row y, its 8-bit view and that view's horizontal neighbour max

    const int t_begin = y_begin - 5, t_end = y_end + 3;
    uint32_t nxt0 = load_row(t_begin), nxt1 = load_row(t_begin + 1);
    // Six rows per trip of the outer loop: the rolling state is 2 and 3 rows deep, so after 6 fully
    // unrolled rows every value is back in its own register and the per-row state copies disappear.
    // The last trip may run up to 5 rows past t_end: they load reflected rows and store nothing.
    // STEADY trips (interior strips of an aligned image with response + mask + keypoint flags and no NMS2 map - the batched
    // path's outputs - and all six rows of the trip far from the image's and the segment's first and last rows): no
    // conditional memory operation at all.  The four margin lanes store to a dump slot instead of being masked off and every
    // lane stores the (wave-uniform) flag words, so the trip is straight-line code and the compiler's s_waitcnt in front of a
    // prefetched row counts exactly the stores that may stay in flight.  Behind conditional stores it could not, and
    // drained the whole queue (vmcnt(0), twice per trip, right behind the newest prefetch): the kernel sat in s_waitcnt
    // for 47 % of its cycles waiting for its own response stores to be acknowledged.
    auto trip = [&](auto steady_tag, const int t0) {
    constexpr bool STEADY = decltype(steady_tag)::value;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int t = t0 + u;
        const uint32_t raw = nxt0;
        nxt0 = nxt1;
        if constexpr (STEADY) {  // 0 <= t + 2 < rows: no reflection (its loop would be a branch in front of every prefetch)
            const uint32_t roff = __builtin_amdgcn_readfirstlane((uint32_t)(t + 2) * (uint32_t)cols);
            __builtin_memcpy(&nxt1, src + roff + (uint32_t)x0, 4);
        } else {
            nxt1 = load_row(t + 2);
        }

        // ---- blur row t-1: vertical [1 2 1] on rows t-2,t-1,t, then horizontal -------------------
        const uint32_t e3 = raw & 0x00ff00ffu, o3 = (raw >> 8) & 0x00ff00ffu;
        const uint32_t ve = e1 + e3 + (e2 << 1), vo = o1 + o3 + (o2 << 1);  // (V0,V2), (V1,V3) <= 1020
        const uint32_t lvo = dpp_left(vo), rve = dpp_right(ve);
        const uint32_t he = __builtin_amdgcn_alignbit(vo, lvo, 16) + vo + (ve << 1) + 0x00080008u;  // (V-1,V1)+(V1,V3)+2(V0,V2)+8
        const uint32_t ho = __builtin_amdgcn_alignbit(rve, ve, 16) + ve + (vo << 1) + 0x00080008u;  // (V2,V4)+(V0,V2)+2(V1,V3)+8
        const uint32_t be0 = pk_lshr4_u16(he), bo0 = pk_lshr4_u16(ho);                             // blurred row t-1 (<= 255 per lane)
        e1 = e2, o1 = o2, e2 = e3, o2 = o3;

        // ---- gradients + products of row g = t-2 ---------------------------------------------------
        const uint32_t lbo = dpp_left(bo1), rbe = dpp_right(be1);
        const uint32_t ixe = pk_sub_i16(bo1, __builtin_amdgcn_alignbit(bo1, lbo, 16));  // (B1-B-1, B3-B1)
        const uint32_t ixo = pk_sub_i16(__builtin_amdgcn_alignbit(rbe, be1, 16), be1);  // (B2-B0, B4-B2)
        const uint32_t iye = pk_sub_i16(be0, be2), iyo = pk_sub_i16(bo0, bo2);          // B(g+1) - B(g-1)
        be2 = be1, bo2 = bo1, be1 = be0, bo1 = bo0;
        int ix[4], iy[4];  // sign-extended halves: folded into the multiplies as SDWA operands
        ix[0] = (int)(short)(ixe & 0xffff), ix[2] = (int)ixe >> 16, ix[1] = (int)(short)(ixo & 0xffff), ix[3] = (int)ixo >> 16;
        iy[0] = (int)(short)(iye & 0xffff), iy[2] = (int)iye >> 16, iy[1] = (int)(short)(iyo & 0xffff), iy[3] = (int)iyo >> 16;
        int PC[3][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            PC[0][k] = ix[k] * ix[k];
            PC[1][k] = iy[k] * iy[k];
            PC[2][k] = ix[k] * iy[k];
        }

        // ---- 3x3 sums of row b = t-3: vertical first, then horizontal on shared pair sums -----------
        // copyMakeBorder(..., BORDER_REPLICATE) of the gradients (:42-43): rows -1 / rows repeat rows
        // 0 / rows-1, columns likewise
        const int b = t - 3;
        uint32_t Rb[4];
        if (STEADY || (b >= 0 && b < rows)) {  // wave-uniform
            int V[3][4];
            if (!STEADY && (b - 1 < 0 || b + 1 >= rows)) {  // wave-uniform: first / last image row
                asm volatile("");              // keeps this a branch: if-converted it is 24 selects on every row
                const bool top_rep = b - 1 < 0, bot_rep = b + 1 >= rows;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int k = 0; k < 4; ++k) V[c][k] = (top_rep ? PB[c][k] : PA[c][k]) + PB[c][k] + (bot_rep ? PB[c][k] : PC[c][k]);
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int k = 0; k < 4; ++k) V[c][k] = PA[c][k] + PB[c][k] + PC[c][k];
            }
            int S[3][4];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                int vm1 = (int)dpp_left((uint32_t)V[c][3]), v4 = (int)dpp_right((uint32_t)V[c][0]);
                if (EDGE) {
                    if (ANYW) {  // the image ends inside this lane: column `cols` repeats column cols-1
#pragma unroll
                        for (int j = 1; j <= 3; ++j)
                            if (jedge == j) V[c][j] = V[c][j - 1];
                    }
                    vm1 = left_edge ? V[c][0] : vm1;
                    v4 = right_edge ? V[c][3] : v4;
                }
                const int a01 = V[c][0] + V[c][1], a23 = V[c][2] + V[c][3];
                S[c][0] = a01 + vm1;
                S[c][1] = a01 + V[c][2];
                S[c][2] = a23 + V[c][1];
                S[c][3] = a23 + v4;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double det_d = __builtin_fma(-(double)S[2][k], (double)S[2][k], (double)S[0][k] * (double)S[1][k]);
                const float det = (float)det_d;
                const float tr = (float)(S[0][k] + S[1][k]);
                const float trtr = tr * tr;
                const float ktr = a.k * trtr;
                Rb[k] = __float_as_uint(__builtin_fmaxf(det - ktr, 0.0f));  // :60-62 (det, ktr >= +0: never -0)
            }
            if (EDGE) {  // pixels outside the image carry no response
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool px_in = ANYW ? (lane_in && (lane_full || k < jedge)) : lane_in;
                    Rb[k] = px_in ? Rb[k] : 0u;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) Rb[k] = 0u;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) PA[c][k] = PB[c][k], PB[c][k] = PC[c][k];

        // ---- horizontal maxima of row b: NMS2 window columns x-2 .. x+1 (:100-101) -----------------
        uint32_t h3n[4], h4n[4], nbn[4], Cb[4];
        {
            const uint32_t l2 = dpp_left(Rb[2]), l3 = dpp_left(Rb[3]), r0 = dpp_right(Rb[0]);
            const uint32_t m01 = max(Rb[0], Rb[1]), m23 = max(Rb[2], Rb[3]);
            h4n[0] = umax3(l2, l3, m01);
            h4n[1] = umax3(l3, m01, Rb[2]);
            h4n[2] = max(m01, m23);
            h4n[3] = umax3(Rb[1], m23, r0);
        }
        // ---- 8-bit view (convertScaleAbs, :176): rint(min(R, 255)), 0 from 2^31 on -------------------
#pragma unroll
        for (int k = 0; k < 4; ++k) Cb[k] = __float_as_uint(__builtin_rintf(__uint_as_float(min(Rb[k], 0x437f0000u))));
        if (__builtin_amdgcn_ballot_w64(h4n[2] >= 0x4f000000u)) {
            // rare, wave-uniform: x86 cvRound wraps responses >= 2^31 to INT_MIN -> 0 in the view
            asm volatile("");  // not to be if-converted
#pragma unroll
            for (int k = 0; k < 4; ++k) Cb[k] = Rb[k] < 0x4f000000u ? Cb[k] : 0u;
        }
        nbn[0] = max(dpp_left(Cb[3]), Cb[1]);
        nbn[1] = max(Cb[0], Cb[2]);
        nbn[2] = max(Cb[1], Cb[3]);
        nbn[3] = max(dpp_right(Cb[0]), Cb[2]);
#pragma unroll
        for (int k = 0; k < 4; ++k) h3n[k] = max(nbn[k], Cb[k]);

        // ---- finalise row y = t-4 ------------------------------------------------------------------
        const int y = t - 4;
        if (STEADY || (y >= y_begin && y < y_end)) {  // wave-uniform
            uint32_t mword = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t mx = umax3(h3a[k], h3n[k], nby[k]);  // 8 neighbours
                mword |= Cy[k] > mx ? 0xffu << (8 * k) : 0u;
            }
            uint32_t n2[4] = {0u, 0u, 0u, 0u};
            unsigned long long kpw[4] = {0ull, 0ull, 0ull, 0ull};
            if (STEADY || (y >= 2 && y < rows - 2)) {  // wave-uniform: NMS2 rows [2, rows-2) (:94)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t w4 = max(umax3(h4a[k], h4b[k], h4c[k]), h4n[k]);  // rows y-2..y+1, cols x-2..x+1
                    const bool pass = inter[k] && Ry[k] >= w4;
                    n2[k] = pass ? w4 : 0u;
                    // cvt(w4) > 253 (Harris_corners.cpp:139): 253.5 <= w4 < 2^31 as one unsigned range test
                    kpw[k] = __builtin_amdgcn_ballot_w64(Ry[k] >= w4) & __builtin_amdgcn_ballot_w64(w4 - 0x437d8000u < 0x4f000000u - 0x437d8000u) & inter_m[k];
                }
            }
            if constexpr (STEADY) {
                const size_t off = blockIdx.z * N + __builtin_amdgcn_readfirstlane((uint32_t)y * (uint32_t)cols) + x0;
                // (one base pointer and a selected DISTANCE: a select between two pointers came back as two conditional stores)
                const ptrdiff_t dr = lane_out ? reinterpret_cast<const uint8_t*>(a.resp + off) - a.dump : 0;
                const ptrdiff_t dm = lane_out ? (a.mask + off) - a.dump : 16;
                *reinterpret_cast<float4*>(a.dump + dr) = make_float4(__uint_as_float(Ry[0]), __uint_as_float(Ry[1]), __uint_as_float(Ry[2]), __uint_as_float(Ry[3]));
                *reinterpret_cast<uint32_t*>(a.dump + dm) = mword;
                unsigned long long* F = a.flags + blockIdx.z * a.fframe + ((size_t)y * a.nstrips + strip) * 4;
                F[0] = kpw[0], F[1] = kpw[1], F[2] = kpw[2], F[3] = kpw[3];  // every lane, the same four words
            } else {
            if (lane_out) {
                // row start on the scalar unit (one scalar multiply instead of a v_mul_lo + v_mul_hi + add-with-carry per lane: fewer instructions, not a faster one; rows * cols < 2^32)
                const size_t off = blockIdx.z * N + __builtin_amdgcn_readfirstlane((uint32_t)y * (uint32_t)cols) + x0;
                const float4 rv = make_float4(__uint_as_float(Ry[0]), __uint_as_float(Ry[1]), __uint_as_float(Ry[2]), __uint_as_float(Ry[3]));
                const float4 nv = make_float4(__uint_as_float(n2[0]), __uint_as_float(n2[1]), __uint_as_float(n2[2]), __uint_as_float(n2[3]));
                if (!ANYW) {
                    if (a.resp) *reinterpret_cast<float4*>(a.resp + off) = rv;
                    if (a.mask) *reinterpret_cast<uint32_t*>(a.mask + off) = mword;
                    if (a.nms2) *reinterpret_cast<float4*>(a.nms2 + off) = nv;
                } else if (!EDGE || lane_full) {  // same stores, not 16 / 4-byte aligned
                    if (a.resp) __builtin_memcpy(a.resp + off, &rv, 16);
                    if (a.mask) __builtin_memcpy(a.mask + off, &mword, 4);
                    if (a.nms2) __builtin_memcpy(a.nms2 + off, &nv, 16);
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (k < jedge) {
                            if (a.resp) a.resp[off + k] = __uint_as_float(Ry[k]);
                            if (a.mask) a.mask[off + k] = (uint8_t)(mword >> (8 * k));
                            if (a.nms2) a.nms2[off + k] = __uint_as_float(n2[k]);
                        }
                }
            }
            if (a.flags && lane == 0) {
                unsigned long long* F = a.flags + blockIdx.z * a.fframe + ((size_t)y * a.nstrips + strip) * 4;
                F[0] = kpw[0], F[1] = kpw[1], F[2] = kpw[2], F[3] = kpw[3];
            }
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h3a[k] = h3b[k], h3b[k] = h3n[k];
            h4a[k] = h4b[k], h4b[k] = h4c[k], h4c[k] = h4n[k];
            Ry[k] = Rb[k], Cy[k] = Cb[k], nby[k] = nbn[k];
        }
    }
    };  // trip
    const bool fast_ok = !EDGE && !ANYW && a.resp && a.mask && a.flags && !a.nms2 && a.dump;  // wave-uniform
    const int y_lo = max(y_begin, 2), y_hi = min(y_end, rows - 2);
    int t0 = t_begin;
    const int t_hi = min(y_hi - 1, rows - 7);  // a steady trip starts at t0 < t_hi: its last row y = t0 + 1 < y_hi (the segment's own rows only: the next segment's wave stores its own), its last prefetch t0 + 7 < rows
    for (; t0 <= t_end && !(fast_ok && t0 - 4 >= y_lo && t0 < t_hi && t0 + 2 >= 0); t0 += 6) trip(std::false_type{}, t0);
    if constexpr (!EDGE && !ANYW) {
        // one steady trip peeled in front of the loop: the loop header's counter state is the join of the entry edge and the
        // back edge, and entering from guarded code would make every wait of the loop as strict as the entry's (vmcnt(2))
        if (fast_ok && t0 <= t_end && t0 < t_hi) {
            trip(std::true_type{}, t0);
            t0 += 6;
        }
        for (; fast_ok && t0 <= t_end && t0 < t_hi; t0 += 6) trip(std::true_type{}, t0);
    }
    for (; t0 <= t_end; t0 += 6) trip(std::false_type{}, t0);
}

// grid = (ceil(nstrips*nseg / 4), 1, frames), block = 256 (4 independent waves).
//
// Instruction budget (round 2): about half the VALU instructions per pixel of round 1.  What changed:
// the 3x3 structure-tensor sums run vertically first (one v_add3_u32 per pixel and channel on the
// rolling product rows) and then horizontally on shared pair sums; every image-border special case
// (replicate rows / columns of the padded gradients, pixels outside the image, responses >= 2^31 in
// the 8-bit view) sits behind a wave-uniform branch or a template flag, so interior strips run no
// select at all; lane exchanges are bound_ctrl DPP moves without an "old" operand; packed 16-bit
// shifts replace shift + mask; maxima and comparisons of the non-negative responses are integer ops.
template <bool ANYW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_harris_strip(const HarrisStripArgs a) {
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    const int nseg = (a.rows + a.seg - 1) / a.seg;
    if (wid >= a.nstrips * nseg) return;  // wave-uniform
    const int strip = wid % a.nstrips, sgi = wid / a.nstrips;
    const int y_begin = sgi * a.seg, y_end = min(y_begin + a.seg, a.rows);
    // margin lanes included, does this strip lie strictly inside the image?  (scalar)
    const bool edge_strip = strip == 0 || (strip + 1) * HS_STRIP_W + 8 > a.cols;
    if (edge_strip)
        harris_strip_rows<ANYW, true>(a, lane, strip, y_begin, y_end);
    else
        harris_strip_rows<ANYW, false>(a, lane, strip, y_begin, y_end);
}

}  // namespace vslam
