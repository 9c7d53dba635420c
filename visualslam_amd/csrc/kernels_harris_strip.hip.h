// K-H: the whole Harris executable's per-pixel work in ONE pass over the frame (gfx950).
//
//   GaussianBlur 3x3 (Harris_corners.cpp:158) -> Sobel x/y ksize 1 (:163-164) -> replicate-
//   padded 3x3 structure tensor + response (HarrisCorner, :31-68) -> convertScaleAbs (:176) +
//   NonMaximumSuppression window 3 (:70-81) -> NMS2 window 5 (:83-129) -> keypoint test (:139)
//
// Structure: a WAVE owns a vertical strip.  Each lane owns 4 horizontally adjacent pixels (one
// dword of the 8-bit frame, so a wave row is a 256-byte coalesced load); rows stream top to
// bottom through registers (rolling 3-row / 4-row state per stage), and the only cross-lane
// traffic is the +-1 lane exchange of stage edges with DPP wave shifts (v_mov_b32_dpp
// wave_shr:1 / wave_shl:1).  Nothing goes through LDS or back through HBM: the frame is read
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

#include "kernels_generic.hip.h"

namespace vslam {

constexpr int HS_VALID_LANES = 60;
constexpr int HS_STRIP_W = 4 * HS_VALID_LANES;  // 240 output columns per wave strip

typedef short s2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t from_left(uint32_t v) {  // value of lane-1
    return __builtin_amdgcn_update_dpp(0u, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t from_right(uint32_t v) {  // value of lane+1
    return __builtin_amdgcn_update_dpp(0u, v, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ float from_left_f(float v) { return __uint_as_float(from_left(__float_as_uint(v))); }
__device__ __forceinline__ float from_right_f(float v) { return __uint_as_float(from_right(__float_as_uint(v))); }
__device__ __forceinline__ uint32_t pk_sub_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, (s2_t)(__builtin_bit_cast(s2_t, a) - __builtin_bit_cast(s2_t, b)));
}
// 8-bit view of a non-negative response, kept as a float (0..255): see cvt_abs_u8
__device__ __forceinline__ float cvt8(float x) { return x < 2147483648.0f ? __builtin_rintf(fminf(x, 255.0f)) : 0.0f; }

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
};

// grid = (ceil(nstrips*nseg / 4), 1, frames), block = 256 (4 independent waves).
template <bool ANYW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_harris_strip(const HarrisStripArgs a) {
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nseg = (a.rows + a.seg - 1) / a.seg;
    if (wid >= a.nstrips * nseg) return;  // wave-uniform
    const int strip = wid % a.nstrips, sgi = wid / a.nstrips;
    const int rows = a.rows, cols = a.cols;
    const size_t N = (size_t)rows * cols;
    const uint8_t* src = a.img + blockIdx.z * a.frame;
    const int x0 = strip * HS_STRIP_W + 4 * (lane - 2);
    const int y_begin = sgi * a.seg, y_end = min(y_begin + a.seg, rows);
    const bool lane_in = x0 >= 0 && x0 < cols;                 // owns at least one image pixel
    const bool lane_full = x0 >= 0 && x0 + 4 <= cols;          // owns four (always, when !ANYW)
    const bool lane_out = lane >= 2 && lane < 62 && lane_in;
    const bool left_edge = x0 == 0, right_edge = x0 + 4 == cols;
    const int jedge = ANYW && lane_in && !lane_full ? cols - x0 : 0;  // 1..3: first pixel slot outside the image
    bool inter[4];  // NMS2 is evaluated for columns [2, cols-2)
#pragma unroll
    for (int k = 0; k < 4; ++k) inter[k] = lane_out && x0 + k >= 2 && x0 + k < cols - 2;

    auto load_row = [&](int t) -> uint32_t {
        const uint8_t* row = src + (size_t)reflect101(t, rows) * cols;
        if (ANYW ? lane_full : lane_in) {
            uint32_t w;
            __builtin_memcpy(&w, row + x0, 4);  // dword aligned when !ANYW
            return w;
        }
        return (uint32_t)row[reflect101(x0, cols)] | ((uint32_t)row[reflect101(x0 + 1, cols)] << 8) |
               ((uint32_t)row[reflect101(x0 + 2, cols)] << 16) | ((uint32_t)row[reflect101(x0 + 3, cols)] << 24);
    };

    // rolling state (suffix = row relative to the row being loaded, t)
    uint32_t e1 = 0, o1 = 0, e2 = 0, o2 = 0;          // raw rows t-2, t-1 as 16-bit lanes (p0,p2)/(p1,p3)
    uint32_t be2 = 0, bo2 = 0, be1 = 0, bo1 = 0;      // blurred rows t-3, t-2
    int hsA[3][4] = {}, hsB[3][4] = {};               // horizontal product sums of rows t-4, t-3
    float h3a[4] = {}, h3b[4] = {};                   // max3 of the 8-bit view, rows y-1, y      (y = t-4)
    float h4a[4] = {}, h4b[4] = {}, h4c[4] = {};      // max4 rows y-2, y-1, y
    float Ry[4] = {}, Cy[4] = {}, nby[4] = {};        // response row y, its 8-bit view and that view's horizontal neighbour max

    const int t_begin = y_begin - 5, t_end = y_end + 3;
    uint32_t nxt0 = load_row(t_begin), nxt1 = load_row(t_begin + 1);
    // Six rows per trip of the outer loop: the rolling state is 2 and 3 rows deep, so after 6 fully
    // unrolled rows every value is back in its own register and the per-row state copies (a
    // quarter of the loop's VALU instructions) disappear.  The last trip may run up to 5 rows past
    // t_end: they load reflected rows and store nothing.
    for (int t0 = t_begin; t0 <= t_end; t0 += 6)
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        const int t = t0 + u;
        const uint32_t raw = nxt0;
        nxt0 = nxt1;
        nxt1 = load_row(t + 2);

        // ---- blur row t-1: vertical [1 2 1] on rows t-2,t-1,t, then horizontal -------------------
        const uint32_t e3 = raw & 0x00ff00ffu, o3 = (raw >> 8) & 0x00ff00ffu;
        const uint32_t ve = e1 + e3 + (e2 << 1), vo = o1 + o3 + (o2 << 1);  // (V0,V2), (V1,V3) <= 1020
        const uint32_t lvo = from_left(vo), rve = from_right(ve);
        const uint32_t he = __builtin_amdgcn_alignbit(vo, lvo, 16) + vo + (ve << 1) + 0x00080008u;  // (V-1,V1)+(V1,V3)+2(V0,V2)+8
        const uint32_t ho = __builtin_amdgcn_alignbit(rve, ve, 16) + ve + (vo << 1) + 0x00080008u;  // (V2,V4)+(V0,V2)+2(V1,V3)+8
        const uint32_t be0 = (he >> 4) & 0x00ff00ffu, bo0 = (ho >> 4) & 0x00ff00ffu;              // blurred row t-1
        e1 = e2, o1 = o2, e2 = e3, o2 = o3;

        // ---- gradients + products + horizontal 3-sums of row g = t-2 ------------------------------
        const uint32_t lbo = from_left(bo1), rbe = from_right(be1);
        const uint32_t ixe = pk_sub_i16(bo1, __builtin_amdgcn_alignbit(bo1, lbo, 16));  // (B1-B-1, B3-B1)
        const uint32_t ixo = pk_sub_i16(__builtin_amdgcn_alignbit(rbe, be1, 16), be1);  // (B2-B0, B4-B2)
        const uint32_t iye = pk_sub_i16(be0, be2), iyo = pk_sub_i16(bo0, bo2);          // B(g+1) - B(g-1)
        be2 = be1, bo2 = bo1, be1 = be0, bo1 = bo0;
        int ix[4], iy[4];
        ix[0] = (int)(short)(ixe & 0xffff), ix[2] = (int)ixe >> 16, ix[1] = (int)(short)(ixo & 0xffff), ix[3] = (int)ixo >> 16;
        iy[0] = (int)(short)(iye & 0xffff), iy[2] = (int)iye >> 16, iy[1] = (int)(short)(iyo & 0xffff), iy[3] = (int)iyo >> 16;
        int P[3][6];  // [xx,yy,xy][x-1 .. x+4]
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            P[0][k + 1] = ix[k] * ix[k];
            P[1][k + 1] = iy[k] * iy[k];
            P[2][k + 1] = ix[k] * iy[k];
        }
        int hsC[3][4];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            // copyMakeBorder(..., BORDER_REPLICATE) (:42-43): outside columns repeat the edge column
            const int l = (int)from_left((uint32_t)P[c][4]), r = (int)from_right((uint32_t)P[c][1]);
            P[c][0] = left_edge ? P[c][1] : l;
            P[c][5] = right_edge ? P[c][4] : r;
            if (ANYW) {  // the image ends inside this lane: column `cols` repeats column cols-1
#pragma unroll
                for (int j = 1; j <= 3; ++j)
                    if (jedge == j) P[c][j + 1] = P[c][j];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) hsC[c][k] = P[c][k] + P[c][k + 1] + P[c][k + 2];
        }

        // ---- 3x3 sums and response of row b = t-3 (rows replicate too) ---------------------------
        const int b = t - 3;
        float Rb[4];
        {
            const bool top_rep = b - 1 < 0, bot_rep = b + 1 >= rows;  // wave-uniform
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int s[3];
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    s[c] = (top_rep ? hsB[c][k] : hsA[c][k]) + hsB[c][k] + (bot_rep ? hsB[c][k] : hsC[c][k]);
                const double det_d = __builtin_fma(-(double)s[2], (double)s[2], (double)s[0] * (double)s[1]);
                const float det = (float)det_d;
                const float tr = (float)(s[0] + s[1]);
                const float trtr = tr * tr;
                const float ktr = a.k * trtr;
                const float resp = det - ktr;
                const bool px_in = ANYW ? (lane_in && (lane_full || k < jedge)) : lane_in;
                Rb[k] = (b >= 0 && b < rows && px_in && resp > 0.0f) ? resp : 0.0f;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) hsA[c][k] = hsB[c][k], hsB[c][k] = hsC[c][k];

        // ---- horizontal maxima of row b ------------------------------------------------------------
        float h3n[4], h4n[4], nbn[4], Cb[4];
        {
            const float lm2 = from_left_f(Rb[2]), lm1 = from_left_f(Rb[3]), rp4 = from_right_f(Rb[0]);
            const float ext[7] = {lm2, lm1, Rb[0], Rb[1], Rb[2], Rb[3], rp4};  // x-2 .. x+4
#pragma unroll
            for (int k = 0; k < 4; ++k) h4n[k] = fmaxf(fmaxf(ext[k + 1], ext[k + 3]), fmaxf(ext[k + 2], ext[k]));
#pragma unroll
            for (int k = 0; k < 4; ++k) Cb[k] = cvt8(Rb[k]);
            const float cext[6] = {from_left_f(Cb[3]), Cb[0], Cb[1], Cb[2], Cb[3], from_right_f(Cb[0])};  // x-1 .. x+4
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                nbn[k] = fmaxf(cext[k], cext[k + 2]);
                h3n[k] = fmaxf(nbn[k], cext[k + 1]);
            }
        }

        // ---- finalise row y = t-4 ------------------------------------------------------------------
        const int y = t - 4;
        if (y >= y_begin && y < y_end) {  // wave-uniform
            const bool yin = y >= 2 && y < rows - 2;
            uint32_t mword = 0;
            float n2[4];
            unsigned long long kpw[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float mx = fmaxf(fmaxf(h3a[k], h3n[k]), nby[k]);  // 8 neighbours
                if (Cy[k] > mx) mword |= 0xffu << (8 * k);
                const float w4 = fmaxf(fmaxf(h4a[k], h4b[k]), fmaxf(h4c[k], h4n[k]));  // rows y-2..y+1, cols x-2..x+1
                const bool pass = yin && inter[k] && Ry[k] >= w4;
                n2[k] = pass ? w4 : 0.0f;
                kpw[k] = __ballot(pass && w4 >= 253.5f && w4 < 2147483648.0f);  // cvt(w4) > 253, Harris_corners.cpp:139
            }
            if (lane_out) {
                const size_t off = blockIdx.z * N + (size_t)y * cols + x0;
                if (!ANYW) {
                    if (a.resp) *reinterpret_cast<float4*>(a.resp + off) = make_float4(Ry[0], Ry[1], Ry[2], Ry[3]);
                    if (a.mask) *reinterpret_cast<uint32_t*>(a.mask + off) = mword;
                    if (a.nms2) *reinterpret_cast<float4*>(a.nms2 + off) = make_float4(n2[0], n2[1], n2[2], n2[3]);
                } else if (lane_full) {  // same stores, not 16 / 4-byte aligned
                    const float4 rv = make_float4(Ry[0], Ry[1], Ry[2], Ry[3]), nv = make_float4(n2[0], n2[1], n2[2], n2[3]);
                    if (a.resp) __builtin_memcpy(a.resp + off, &rv, 16);
                    if (a.mask) __builtin_memcpy(a.mask + off, &mword, 4);
                    if (a.nms2) __builtin_memcpy(a.nms2 + off, &nv, 16);
                } else {
#pragma unroll
                    for (int k = 0; k < 3; ++k)
                        if (k < jedge) {
                            if (a.resp) a.resp[off + k] = Ry[k];
                            if (a.mask) a.mask[off + k] = (uint8_t)(mword >> (8 * k));
                            if (a.nms2) a.nms2[off + k] = n2[k];
                        }
                }
            }
            if (a.flags && lane == 0) {
                unsigned long long* F = a.flags + blockIdx.z * a.fframe + ((size_t)y * a.nstrips + strip) * 4;
                F[0] = kpw[0], F[1] = kpw[1], F[2] = kpw[2], F[3] = kpw[3];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            h3a[k] = h3b[k], h3b[k] = h3n[k];
            h4a[k] = h4b[k], h4b[k] = h4c[k], h4c[k] = h4n[k];
            Ry[k] = Rb[k], Cy[k] = Cb[k], nby[k] = nbn[k];
        }
    }
}

}  // namespace vslam
