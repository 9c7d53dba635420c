// K-D1, matrix-core form (OPT-IN: VSLAM_MX=1 / vslam_ctx_set_matrix_path; never the default path).
//
// The same contract as k_pyr_octave (kernels_pyramid.hip.h): one launch turns the octave base (u8) into the
// octave's six Gaussian images and five DoG images -- GaussVector + Diff_of_Gauss of
// GaussPyramid.cpp:166-200 -- and the next octave's base (GaussPyramid.cpp:123-126), bit for bit.
//
// BASELINE.json's north star says "no MFMA (no dense contraction here)".  A separable blur IS a pair of
// banded-Toeplitz matrix products, and round 3's probe (tools/mfma_probe.hip) measured that form at 1.8-2.8x
// the packed-dot floor per level; the default path, bench.py's `value` and `roofline` therefore stay on the
// dot kernels and this kernel is reported beside them.
//
// Arithmetic: G = (sum_y sum_x ty*tx*p + 32768) >> 16, exact integers, any order.
// Per level and per 32 x 32 output block, as v_mfma_i32_32x32x32_i8 (A: lane l holds row l&31, K bytes
// 16*(l>>5)+j; B: column l&31, same K; C: column l&31, rows (v&3) + 8*(v>>2) + 4*(l>>5) in register v):
//
//   pass 1 (vertical)    C1[x, y'] = sum_y P'[x, y] * Tv[y, y']     A = pixels - 128 from the byte-transposed
//                        LDS image (lane = column x, 16 consecutive rows), B = the taps as a band matrix.
//                        C1 = H - 32768, H the 16-bit column sum; NS = 2 K steps for kernels up to 33 taps, 3 up to 65.
//   hand-off             C1 has the output row y' on the lane and 16 columns x in its registers: the shape of
//                        a B operand whose K index is x.  Operands are 8-bit, H has 16: split into a signed
//                        high-byte plane and a low-byte plane (4 v_perm + 1 v_xor per 4 values).
//   pass 2 (horizontal)  C2[x', y'] = sum_x Th[x', x] * H[x, y']    A = the band matrix with its K columns in
//                        the register order of C1 and its ROWS permuted so that register v of lane half h is
//                        output column 16 h + v: each lane ends up with 16 CONSECUTIVE bytes of its image
//                        row, no lane exchange.  One product per byte plane;
//                        G = ((C2hi << 8) + C2lo) >> 16, the biases and the rounding constant in C2lo's start value.
//   epilogue             as k_pyr_octave: byte 2 of each sum into 16-bit lanes, v_pk_sub_u16 clamp against the
//                        previous level (kept in registers), 16-byte stores of G and D.
//
// A wave owns a strip of 32 rows x SW columns of the tile and walks along it: input block ib goes through
// pass 1 and the split into a ring of NS converted blocks, and as soon as the ring holds output block
// ob = ib - (NS - 1)'s inputs its pass 2 runs.  After the tile is staged the waves never meet again (no barrier, no
// LDS writes).
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>

#include "../../include/vslam.h"

namespace vslam {

// (self-contained: this header is compiled in its own translation unit, the kernels of the other headers are not)
__device__ __forceinline__ int mx_reflect101(int p, int len) {  // cv::borderInterpolate(p, len, BORDER_REFLECT_101), repeated until inside
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
    return p;
}
typedef unsigned short mx_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mx_pk_sub_sat_u16(uint32_t a, uint32_t b) {  // v_pk_sub_u16 clamp
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(mx_us2, a), __builtin_bit_cast(mx_us2, b)));
}

constexpr int MX_SITE_PAIRS_32 = 6;  // 11 lattice rows per 32-row strip, two per register
typedef int mx_v4i __attribute__((ext_vector_type(4)));
typedef int mx_v16i __attribute__((ext_vector_type(16)));

template <int TW_, int TH_, int NWX_, int DBUF_, int N0, int N1, int N2, int N3, int N4, int N5>
struct MxCfg {
    static constexpr int TW = TW_, TH = TH_, NWX = NWX_;
    // 1: separate LDS buffers for a level's G and D strips; 0: one buffer, D waits in registers and follows G through it
    // (same box, octaves 0 + 1 of 256 x 1080p alone: 8.0 ms per step with two buffers, 9.35 ms with one - two where they fit)
    static constexpr int DBUF = DBUF_;
    static constexpr int SITE_PAIRS = MX_SITE_PAIRS_32;  // the fused scan: 11 lattice rows per 32-row strip, two per register
    static constexpr int SW = TW / NWX;   // columns of a wave's strip
    static constexpr int NWY = TH / 32;
    static constexpr int NW = NWX * NWY, NT = 64 * NW;
    static constexpr int NOB = SW / 32;   // output blocks per strip
    static constexpr int n(int l) { return l == 0 ? N0 : l == 1 ? N1 : l == 2 ? N2 : l == 3 ? N3 : l == 4 ? N4 : N5; }
    static constexpr int r(int l) { return n(l) / 2; }
    static constexpr int off(int l) { return (r(l) + 15) / 16 * 16; }   // the K window starts `off` before the output block
    static constexpr int ns(int l) { return 1 + off(l) / 16; }          // K steps of 32: window [-off, -off + 32 ns) covers [-r, 32 + r)
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int R = cmax(cmax(cmax(off(0), off(1)), cmax(off(2), off(3))), cmax(off(4), off(5)));  // staged halo
    static constexpr int NSMAX = 1 + R / 16;
    static constexpr int RQ = (TH + 2 * R) / 4;   // row quads of the staged tile
    static constexpr int RW = TW + 2 * R;         // its width = dwords per row quad
    static constexpr int RWP = RW + 4;
    // per-wave output buffer: a level's G rows of the strip, then its D rows; SW bytes + 16 per row (row pitch = 4 mod 32
    // dwords: the 16-byte writes of 8 consecutive rows and the 16-byte reads along a row are conflict-free)
    static constexpr int OBP = SW / 4 + 4;            // dwords per buffered row
    static constexpr int OBUF = 32 * OBP;             // dwords per wave
    static constexpr int STAGE_DWORDS = RQ * RWP;
    // (+ 8 rows of slack behind the last wave's buffers: the fused lattice scan reads up to 5 rows past a strip's D buffer
    // for lattice rows it does not own, and drops what it read)
    static constexpr int LDS_BYTES = (STAGE_DWORDS + NW * (1 + DBUF) * OBUF + 8 * OBP) * 4;
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup's LDS");
    static_assert(TW % (32 * NWX) == 0 && TH % 32 == 0 && NT <= 1024 && NT % 128 == 0, "strips of 32-row x 32-column blocks");
    static_assert(SW == 128, "the output flush maps a wave's 64 lanes to 8 rows x 128 bytes");
    static_assert(r(0) >= 1 && (N0 & 1) && (N1 & 1) && (N2 & 1) && (N3 & 1) && (N4 & 1) && (N5 & 1), "odd kernels");
};

// Operand fragments in lane order, one 16-byte fragment per K step: see mx_pack_taps.
template <class CFG>
struct MxTaps {
    mx_v4i b1[6][CFG::NSMAX][64];
    mx_v4i a2[6][CFG::NSMAX][64];
};

// Stages the TW x TH tile with halo R, byte-transposed (a dword = 4 vertically adjacent pixels of one column),
// BORDER_REFLECT_101 resolved at fill time, every byte ^ bias (0x80: pixels - 128 as signed bytes).
// Four pixels of one image row at columns x .. x+3 under BORDER_REFLECT_101: one dword load where the four lie inside the
// row, one dword load of the mirrored run with its bytes reversed where they lie wholly in the first reflection on either
// side, byte by byte (repeated reflection) only where they straddle an edge or the row is shorter than the halo.
__device__ __forceinline__ uint32_t mx_load4_reflect(const uint8_t* __restrict__ row, int x, int cols) {
    if (x >= 0 && x + 3 < cols) return *reinterpret_cast<const uint32_t*>(row + x);
    if (x + 3 < 0 && -x < cols) {  // columns x..x+3 mirror to -x, -x-1, -x-2, -x-3 (all >= 1)
        uint32_t v;
        __builtin_memcpy(&v, row + (-x - 3), 4);
        return __builtin_amdgcn_perm(0u, v, 0x00010203);
    }
    if (x >= cols && 2 * (cols - 1) - x - 3 >= 0) {  // mirror to 2(cols-1)-x, ... - 3 (all <= cols - 2)
        uint32_t v;
        __builtin_memcpy(&v, row + (2 * (cols - 1) - x - 3), 4);
        return __builtin_amdgcn_perm(0u, v, 0x00010203);
    }
    return (uint32_t)row[mx_reflect101(x, cols)] | ((uint32_t)row[mx_reflect101(x + 1, cols)] << 8) | ((uint32_t)row[mx_reflect101(x + 2, cols)] << 16) |
           ((uint32_t)row[mx_reflect101(x + 3, cols)] << 24);
}

template <int TW, int TH, int R, int RWP, int NT>
__device__ __forceinline__ void mx_stage_tile(const uint8_t* __restrict__ src, int rows, int cols, int pitch, int tile_x0, int tile_y0,
                                              uint32_t* __restrict__ rp, uint32_t bias) {
    constexpr int RW = TW + 2 * R, RQ = (TH + 2 * R) / 4;
    const int tid = threadIdx.x;
    const bool interior = tile_x0 - R >= 0 && tile_x0 + TW + R <= cols && tile_y0 - R >= 0 && tile_y0 + TH + R <= rows;
    if (interior) {
        for (int it = tid; it < RQ * (RW / 16); it += NT) {
            const int yq = it / (RW / 16), xs = it - yq * (RW / 16);
            const uint8_t* p = src + (size_t)(tile_y0 - R + 4 * yq) * pitch + (tile_x0 - R + 16 * xs);
            uint4 a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const uint4*>(p + (size_t)k * pitch);
            const uint32_t* aw[4] = {&a[0].x, &a[1].x, &a[2].x, &a[3].x};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t r0 = aw[0][q], r1 = aw[1][q], r2 = aw[2][q], r3 = aw[3][q];
                const uint32_t p01l = __builtin_amdgcn_perm(r1, r0, 0x05010400), p01h = __builtin_amdgcn_perm(r1, r0, 0x07030602);
                const uint32_t p23l = __builtin_amdgcn_perm(r3, r2, 0x05010400), p23h = __builtin_amdgcn_perm(r3, r2, 0x07030602);
                uint4 t;
                t.x = __builtin_amdgcn_perm(p23l, p01l, 0x05040100) ^ bias;
                t.y = __builtin_amdgcn_perm(p23l, p01l, 0x07060302) ^ bias;
                t.z = __builtin_amdgcn_perm(p23h, p01h, 0x05040100) ^ bias;
                t.w = __builtin_amdgcn_perm(p23h, p01h, 0x07060302) ^ bias;
                *reinterpret_cast<uint4*>(rp + yq * RWP + 16 * xs + 4 * q) = t;
            }
        }
    } else if (cols >= 4 && R < cols && tile_x0 + TW + R - 1 <= 2 * (cols - 1) && R < rows && tile_y0 + TH + R - 1 <= 2 * (rows - 1)) {
        // Border tiles whose halo reaches at most ONE reflection on either side (every tile of the coarse octaves at camera
        // sizes: 30 of 40 tiles of a 960 x 540 octave, all of a 480 x 270 one).  Round 5: branch-free.  Four pixels at columns
        // x .. x+3 under BORDER_REFLECT_101 always lie within four consecutive bytes of the row - a forward run, a mirrored
        // run, or a run folded around column 0 / cols-1 - so every case is ONE unaligned dword load at `base` and one v_perm
        // whose selector holds the four byte positions relative to base.  No divergent paths: the four row loads of an item
        // issue back to back and four items are in flight per thread (the loop is unrolled by four).  These kernels run with
        // one workgroup per CU beside the HBM-bound Harris chain and are bound by the latency of exactly these loads.
        auto f1 = [](int x, int n) { return x < 0 ? -x : (x >= n ? 2 * (n - 1) - x : x); };
#pragma unroll 4
        for (int it = tid; it < RQ * (RW / 4); it += NT) {
            const int yq = it / (RW / 4), xq = it - yq * (RW / 4);
            const int gy = tile_y0 - R + 4 * yq, gx = tile_x0 - R + 4 * xq;
            const int p0 = f1(gx, cols), p1 = f1(gx + 1, cols), p2 = f1(gx + 2, cols), p3 = f1(gx + 3, cols);
            const int base = min(min(min(p0, p1), min(p2, p3)), cols - 4);
            const uint32_t sel = (uint32_t)(p0 - base) | ((uint32_t)(p1 - base) << 8) | ((uint32_t)(p2 - base) << 16) | ((uint32_t)(p3 - base) << 24);
            uint32_t a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t v;
                __builtin_memcpy(&v, src + (size_t)f1(gy + k, rows) * pitch + base, 4);
                a[k] = __builtin_amdgcn_perm(0u, v, sel);
            }
            const uint32_t p01l = __builtin_amdgcn_perm(a[1], a[0], 0x05010400), p01h = __builtin_amdgcn_perm(a[1], a[0], 0x07030602);
            const uint32_t p23l = __builtin_amdgcn_perm(a[3], a[2], 0x05010400), p23h = __builtin_amdgcn_perm(a[3], a[2], 0x07030602);
            uint4 t;
            t.x = __builtin_amdgcn_perm(p23l, p01l, 0x05040100) ^ bias;
            t.y = __builtin_amdgcn_perm(p23l, p01l, 0x07060302) ^ bias;
            t.z = __builtin_amdgcn_perm(p23h, p01h, 0x05040100) ^ bias;
            t.w = __builtin_amdgcn_perm(p23h, p01h, 0x07060302) ^ bias;
            *reinterpret_cast<uint4*>(rp + yq * RWP + 4 * xq) = t;
        }
    } else {
        // tiny images (a halo wider than the image: repeated reflection): one dword column (4 pixels) x 4 rows per item, rows
        // reflected per row, columns per dword (mx_load4_reflect)
        for (int it = tid; it < RQ * (RW / 4); it += NT) {
            const int yq = it / (RW / 4), xq = it - yq * (RW / 4);
            const int gy = tile_y0 - R + 4 * yq, gx = tile_x0 - R + 4 * xq;
            uint32_t a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = mx_load4_reflect(src + (size_t)mx_reflect101(gy + k, rows) * pitch, gx, cols);
            const uint32_t p01l = __builtin_amdgcn_perm(a[1], a[0], 0x05010400), p01h = __builtin_amdgcn_perm(a[1], a[0], 0x07030602);
            const uint32_t p23l = __builtin_amdgcn_perm(a[3], a[2], 0x05010400), p23h = __builtin_amdgcn_perm(a[3], a[2], 0x07030602);
            uint4 t;
            t.x = __builtin_amdgcn_perm(p23l, p01l, 0x05040100) ^ bias;
            t.y = __builtin_amdgcn_perm(p23l, p01l, 0x07060302) ^ bias;
            t.z = __builtin_amdgcn_perm(p23h, p01h, 0x05040100) ^ bias;
            t.w = __builtin_amdgcn_perm(p23h, p01h, 0x07060302) ^ bias;
            *reinterpret_cast<uint4*>(rp + yq * RWP + 4 * xq) = t;
        }
    }
}

// ---- octave 0's base formed while the tile is staged: createPyramid's 2x bilinear upsample (GaussPyramid.cpp:110) -----------
// cv::resize(img, Size(), 2, 2, INTER_LINEAR) on CV_8U, bit for bit as k_resize_linear2x_slide (kernels_aux.hip.h): with
// A = s_i + 3 s_{i+1} (or 3 s_i + s_{i+1}) per source row, X = A >> 2, Y = 3A >> 2, the destination rows are
//   dst(2m) = (X(m-1) + Y(m) + 2) >> 2,   dst(2m+1) = (Y(m) + X(m+1) + 2) >> 2
// and clamped source reads reproduce OpenCV's border rule.  The default path writes that base to HBM (8.3 MB per 1080p
// frame) and the octave kernel reads it back with its halo (13 MB); here the tile's base pixels are computed from the
// SOURCE frame into the same byte-transposed LDS image, and the base never exists in HBM.
typedef unsigned short mx_us2c __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mx_pk_lshr2_u16(uint32_t a) {
    return __builtin_bit_cast(uint32_t, (mx_us2c)(__builtin_bit_cast(mx_us2c, a) >> (unsigned short)2));
}
// A of source row m (clamped) for base columns 2 c0 .. 2 c0 + 7 as (j0,j2) (j1,j3) (j4,j6) (j5,j7) 16-bit pairs; c0..c0+3 inside the row
__device__ __forceinline__ void mx_up_hrow(const uint8_t* __restrict__ s, int sstep, int rows_s, int cols_s, int m, int c0, uint32_t (&A)[4]) {
    const uint8_t* r = s + (size_t)min(max(m, 0), rows_s - 1) * sstep;
    uint32_t w;
    __builtin_memcpy(&w, r + c0, 4);
    const uint32_t s0 = r[max(c0 - 1, 0)], s5 = r[min(c0 + 4, cols_s - 1)];
    const uint32_t P = __builtin_amdgcn_perm(w, w, 0x0c010c00);   // (s1, s2)
    const uint32_t Q = __builtin_amdgcn_perm(w, w, 0x0c030c02);   // (s3, s4)
    const uint32_t Qm = __builtin_amdgcn_perm(w, w, 0x0c020c01);  // (s2, s3)
    const uint32_t Pm = s0 | ((w & 0xffu) << 16);                  // (s0, s1)
    const uint32_t Qp = (w >> 24) | (s5 << 16);                    // (s4, s5)
    const uint32_t P3 = P + (P << 1), Q3 = Q + (Q << 1);
    A[0] = Pm + P3, A[1] = P3 + Qm, A[2] = Qm + Q3, A[3] = Q3 + Qp;
}
// eight base pixels of one row from the two source rows' terms: bytes 0..7 in two dwords
__device__ __forceinline__ uint2 mx_up_emit(const uint32_t (&U)[4], const uint32_t (&V)[4]) {
    uint32_t v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = mx_pk_lshr2_u16(U[i] + V[i] + 0x00020002u);
    return make_uint2(v[0] | (v[1] << 8), v[2] | (v[3] << 8));
}
// one base pixel, the literal formula (k_resize_linear2x): dy, dx inside the upsampled image
__device__ __forceinline__ uint32_t mx_up_pixel(const uint8_t* __restrict__ s, int sstep, int rows_s, int cols_s, int dy, int dx) {
    int sx = (dx >> 1) - 1 + (dx & 1), a1 = (dx & 1) ? 512 : 1536;
    if (sx < 0) sx = 0, a1 = 0;
    if (sx >= cols_s - 1) sx = cols_s - 1, a1 = 0;
    const int a0 = 2048 - a1, sx1 = sx + 1 < cols_s ? sx + 1 : sx;
    const int sy = (dy >> 1) - 1 + (dy & 1), b1 = (dy & 1) ? 512 : 1536, b0 = 2048 - b1;
    const uint8_t* r0 = s + (size_t)min(max(sy, 0), rows_s - 1) * sstep;
    const uint8_t* r1 = s + (size_t)min(max(sy + 1, 0), rows_s - 1) * sstep;
    const int h0 = r0[sx] * a0 + r0[sx1] * a1, h1 = r1[sx] * a0 + r1[sx1] * a1;
    const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    return (uint32_t)min(max(v, 0), 255);
}

// Stages the tile of the UPSAMPLED image (2 rows_s x 2 cols_s) from the source frame.  Item = 16 base rows x 8 base columns
// (thread = row segment * (RW / 8) + column group): a sliding pass down 8 source rows (+ one above, one below) where the
// item's rows and columns lie inside the upsampled image; row by row (two source rows recomputed per base row) where some of
// its rows are BORDER_REFLECT_101 images of rows inside; pixel by pixel where its columns are.
// QU: unroll factor of the sliding pass over an item's four row quads (4: the 250-register kernels; 1: kernels held to 128 registers)
// SQ: row quads per staging item (4: items of 16 rows; 2: items of 8 rows - twice the items, for workgroups of 512 threads)
template <int TW, int TH, int R, int RWP, int NT, int QU = 4, int SQ = 4>
__device__ __forceinline__ void mx_stage_tile_up2(const uint8_t* __restrict__ s, int sstep, int rows_s, int cols_s, int tile_x0, int tile_y0,
                                                  uint32_t* __restrict__ rp, uint32_t bias) {
    constexpr int RW = TW + 2 * R, RH = TH + 2 * R, NG = RW / 8, NSEG = RH / (4 * SQ);
    static_assert(RW % 8 == 0 && RH % (4 * SQ) == 0 && (SQ == 2 || SQ == 4), "staging items of 4 SQ rows x 8 columns");
    const int H2 = 2 * rows_s, W2 = 2 * cols_s;
    // Items whose eight columns lie wholly left or right of the image are BORDER_REFLECT_101 images of columns this tile
    // stages anyway: they are skipped here and filled afterwards by dword copies inside LDS (a column of the byte-transposed
    // tile is one dword per row quad).  Computed pixel by pixel - four byte loads and ~25 operations each, 128 pixels per
    // item, on threads of all four waves - they made every left / right border tile (34 of 510 at 1080p) about twice as long.
    // (block-uniform; the mirrored columns must lie inside the image AND inside the staged range)
    const int x_lo = tile_x0 - R, x_hi = tile_x0 + TW + R;  // staged columns [x_lo, x_hi)
    const bool mir_l = x_lo < 0 && -x_lo < W2 && -x_lo < x_hi;
    const bool mir_r = x_hi > W2 && 2 * (W2 - 1) - (x_hi - 1) >= max(x_lo, 0);
    for (int it = threadIdx.x; it < NSEG * NG; it += NT) {
        const int seg = it / NG, g = it - seg * NG;
        const int by = tile_y0 - R + 4 * SQ * seg, bx = tile_x0 - R + 8 * g;  // both even
        if ((mir_l && bx + 7 < 0) || (mir_r && bx >= W2)) continue;
        // four base rows of the item (8 columns each): two 4 x 4 byte transposes, 8 columns x 4 vertical pixels = two 16-byte LDS stores
        auto put_quad = [&](int q, const uint2& a0, const uint2& a1, const uint2& a2, const uint2& a3) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const uint32_t r0 = hh ? a0.y : a0.x, r1 = hh ? a1.y : a1.x, r2 = hh ? a2.y : a2.x, r3 = hh ? a3.y : a3.x;
                const uint32_t p01l = __builtin_amdgcn_perm(r1, r0, 0x05010400), p01h = __builtin_amdgcn_perm(r1, r0, 0x07030602);
                const uint32_t p23l = __builtin_amdgcn_perm(r3, r2, 0x05010400), p23h = __builtin_amdgcn_perm(r3, r2, 0x07030602);
                uint4 t;
                t.x = __builtin_amdgcn_perm(p23l, p01l, 0x05040100) ^ bias;
                t.y = __builtin_amdgcn_perm(p23l, p01l, 0x07060302) ^ bias;
                t.z = __builtin_amdgcn_perm(p23h, p01h, 0x05040100) ^ bias;
                t.w = __builtin_amdgcn_perm(p23h, p01h, 0x07060302) ^ bias;
                *reinterpret_cast<uint4*>(rp + (SQ * seg + q) * RWP + 8 * g + 4 * hh) = t;
            }
        };
        if (bx >= 0 && bx + 7 < W2) {
            const int c0 = bx >> 1;
            if (by >= 0 && by + 4 * SQ - 1 < H2) {
                const int m0 = by >> 1;
                uint32_t Ap[4], Ac[4], An[4], Xp[4], Yc[4], Xc[4];
                mx_up_hrow(s, sstep, rows_s, cols_s, m0 - 1, c0, Ap);
                mx_up_hrow(s, sstep, rows_s, cols_s, m0, c0, Ac);
#pragma unroll
                for (int i = 0; i < 4; ++i) Xp[i] = mx_pk_lshr2_u16(Ap[i]), Xc[i] = mx_pk_lshr2_u16(Ac[i]), Yc[i] = mx_pk_lshr2_u16(Ac[i] + (Ac[i] << 1));
#pragma unroll QU
                for (int q = 0; q < SQ; ++q) {
                    uint2 row[4];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        mx_up_hrow(s, sstep, rows_s, cols_s, m0 + 2 * q + j + 1, c0, An);
                        uint32_t Xn[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) Xn[i] = mx_pk_lshr2_u16(An[i]);
                        row[2 * j] = mx_up_emit(Xp, Yc);      // rows (m-1, m), weights (512, 1536)
                        row[2 * j + 1] = mx_up_emit(Yc, Xn);  // rows (m, m+1), weights (1536, 512)
#pragma unroll
                        for (int i = 0; i < 4; ++i) Xp[i] = Xc[i], Xc[i] = Xn[i], Yc[i] = mx_pk_lshr2_u16(An[i] + (An[i] << 1));
                    }
                    put_quad(q, row[0], row[1], row[2], row[3]);
                }
            } else {
#pragma unroll 1
                for (int q = 0; q < SQ; ++q) {
                    uint2 row[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int yr = mx_reflect101(by + 4 * q + j, H2), m = yr >> 1;
                        uint32_t A0[4], A1[4], U[4], V[4];
                        // even row 2m: X(m-1) + Y(m); odd row 2m+1: Y(m) + X(m+1)
                        mx_up_hrow(s, sstep, rows_s, cols_s, (yr & 1) ? m : m - 1, c0, A0);
                        mx_up_hrow(s, sstep, rows_s, cols_s, (yr & 1) ? m + 1 : m, c0, A1);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t x0 = mx_pk_lshr2_u16(A0[i]), y0 = mx_pk_lshr2_u16(A0[i] + (A0[i] << 1));
                            const uint32_t x1 = mx_pk_lshr2_u16(A1[i]), y1 = mx_pk_lshr2_u16(A1[i] + (A1[i] << 1));
                            U[i] = (yr & 1) ? y0 : x0, V[i] = (yr & 1) ? x1 : y1;
                        }
                        row[j] = mx_up_emit(U, V);
                    }
                    put_quad(q, row[0], row[1], row[2], row[3]);
                }
            }
        } else if constexpr (QU < 4) {
            // (a 128-register kernel: the unrolled form below keeps 32 pixel evaluations in flight and took 114 registers more than
            // there are; this branch runs for the few items that straddle an image edge - byte by byte into the transposed tile)
#pragma unroll 1
            for (int rr = 0; rr < 4 * SQ; ++rr) {
                const int yr = mx_reflect101(by + rr, H2);
                uint8_t* dst = reinterpret_cast<uint8_t*>(rp + (SQ * seg + (rr >> 2)) * RWP + 8 * g) + (rr & 3);
#pragma unroll 1
                for (int k = 0; k < 8; ++k) dst[4 * k] = (uint8_t)(mx_up_pixel(s, sstep, rows_s, cols_s, yr, mx_reflect101(bx + k, W2)) ^ (bias & 0xffu));
            }
        } else {
#pragma unroll 1
            for (int q = 0; q < SQ; ++q) {
                uint2 row[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int yr = mx_reflect101(by + 4 * q + j, H2);
                    uint32_t lo = 0, hi = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lo |= mx_up_pixel(s, sstep, rows_s, cols_s, yr, mx_reflect101(bx + k, W2)) << (8 * k);
                        hi |= mx_up_pixel(s, sstep, rows_s, cols_s, yr, mx_reflect101(bx + 4 + k, W2)) << (8 * k);
                    }
                    row[j] = make_uint2(lo, hi);
                }
                put_quad(q, row[0], row[1], row[2], row[3]);
            }
        }
    }
    if (mir_l || mir_r) {
        __syncthreads();  // the columns the copies read are complete (every row of them: row reflection was resolved by their own items)
        const int nl = mir_l ? -x_lo : 0, nr = mir_r ? x_hi - W2 : 0;  // (x_lo, W2 even: whole 8-column items, all of them skipped above)
        for (int it = threadIdx.x; it < (RH / 4) * (nl + nr); it += NT) {
            const int yq = it / (nl + nr), k = it - yq * (nl + nr);
            const int x = k < nl ? x_lo + k : W2 + (k - nl);
            const int m = x < 0 ? -x : 2 * (W2 - 1) - x;
            rp[yq * RWP + (x - x_lo)] = rp[yq * RWP + (m - x_lo)];
        }
    }
}

// ---- fused lattice scan (initialKeypointDetection, Diff_of_Gauss.cpp:254-297, window 3) -------------------------------
// The default path's k_extrema_w3 reads two thirds of the five DoG planes back from HBM (43 MB per 1080p frame).  Here a
// wave evaluates the lattice sites of its own strip while the DoG rows are still in its LDS buffer: site (a, b) - padded
// (i, j) = (1 + 3a, 1 + 3b) - looks at image rows {3a-1, 3a} x columns {3b-1, 3b} (clamped at 0: padOctave's replicate
// border) of levels c-1, c, c+1 and is a candidate iff D_c(3a, 3b) equals the minimum or the maximum of the twelve.
// A strip owns the sites whose four pixels lie inside it; sites whose window straddles a strip's first row or first
// column (3a = 32 s, 3b = 128 t: one lattice row in 32, one lattice column in 128) are left to k_extrema_pack, which also
// turns this kernel's per-site bytes into the bitmask / list-flag words of include/vslam.h.
//   lane = lattice column b_first + lane (43 of a 128-column strip), 11 lattice rows per strip, two rows per register
//   (16-bit lanes): per DoG level and row pair 8 LDS dwords, 4 v_perm, 4 packed min / max for the rows, then the 2 x 2
//   minima / maxima / centre values of both rows by 7 more packed operations; three levels of them roll in registers.
struct MxExtArgs {
    uint8_t* sitemap;   // [frames][lat_rows][mpitch]: bit 2(c-1) = candidate at level c, bit 2(c-1)+1 = also value >= min_contrast
    size_t mframe;
    int lat_rows, lat_cols, mpitch, min_contrast;
    // seams: the image columns X = 3 SW s (s = 1..nseams) where a lattice column's window (X-1, X) straddles two strips.
    // The strips on either side leave those two DoG columns of every level here, [frames][5][nseams][rows][2] bytes, and
    // k_extrema_pack evaluates the seam sites from them (from the planes it cost a 128-byte line per byte: 18 MB per frame).
    uint8_t* colmap;
    size_t cframe;
    int nseams;
};

typedef unsigned short mx_us2b __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t mx_pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(mx_us2b, a), __builtin_bit_cast(mx_us2b, b)));
}
__device__ __forceinline__ uint32_t mx_pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(mx_us2b, a), __builtin_bit_cast(mx_us2b, b)));
}
__device__ __forceinline__ uint32_t mx_pk_sub_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(mx_us2b, a) - __builtin_bit_cast(mx_us2b, b));
}

constexpr int MX_SITE_PAIRS = MX_SITE_PAIRS_32;

template <int PAIRS>
struct MxSitesT {
    const uint8_t* slr;   // LDS: this lane's dword of column 3b-1 in row r0 of the wave's D buffer (r0 = image row 3 a_first - 1, may be -1)
    const uint8_t* sla0;  // ... in row max(r0, 0): the upper row of the strip's first lattice row
    uint32_t sel;         // v_perm selector: (column 3b-1, 0, column 3b, 0) of a dword pair
    int nk;               // lattice rows this strip owns (wave-uniform, <= 11)
    uint8_t* smap;        // this lane's byte of lattice row a_first (nullptr: the lane has no site)
    uint32_t mpitch;
    uint8_t* cdump;       // seam strips, lanes 0..31 with a row inside the image: this row's byte of DoG level 0 in the seam map
    const uint8_t* cl;    // ... and the byte it comes from: column 0 or SW-1 of this lane's row in the wave's D buffer
    uint32_t clevel;      // bytes per level of the seam map
    uint32_t mc2;         // min_contrast in both 16-bit lanes
    // 2 x 2 minima, maxima and centre values of the two previous DoG levels (slot = level % 2), two lattice rows per register
    uint32_t mn[2][PAIRS], mx[2][PAIRS], sf[2][PAIRS], out[PAIRS];
};
using MxSites = MxSitesT<MX_SITE_PAIRS>;

// DoG level l of the strip is in the wave's D buffer: the 2 x 2 minimum, maximum and the centre value of every owned site;
// from level 2 on, with the two levels before it, the candidate and list bits of centre level c = l - 1 (bits 2(c-1),
// 2(c-1)+1 of the site's byte); then level l takes the place of level l - 2.
template <class CFG, int l>
__device__ __forceinline__ void mx_sites_level(MxSitesT<CFG::SITE_PAIRS>& st) {
    constexpr int RB = CFG::OBP * 4;  // bytes per buffered row
    constexpr int old = l % 2, mid = (l + 1) % 2;  // slots of levels l - 2 and l - 1
#pragma unroll
    for (int p = 0; p < CFG::SITE_PAIRS; ++p) {
        uint32_t lo[2], hi[2], v1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = 2 * p + j;
            const uint32_t* qa = reinterpret_cast<const uint32_t*>(k == 0 ? st.sla0 : st.slr + 3 * k * RB);
            const uint32_t* qb = reinterpret_cast<const uint32_t*>(st.slr + (3 * k + 1) * RB);
            const uint32_t v0 = __builtin_amdgcn_perm(qa[1], qa[0], st.sel);  // (D(3a-1, 3b-1), D(3a-1, 3b)) in 16-bit lanes
            v1[j] = __builtin_amdgcn_perm(qb[1], qb[0], st.sel);             // (D(3a, 3b-1), D(3a, 3b))
            lo[j] = mx_pk_min_u16(v0, v1[j]);
            hi[j] = mx_pk_max_u16(v0, v1[j]);
        }
        // lattice rows 2p (low half) and 2p+1 (high half) side by side
        const uint32_t mn = mx_pk_min_u16(__builtin_amdgcn_perm(lo[1], lo[0], 0x05040100), __builtin_amdgcn_perm(lo[1], lo[0], 0x07060302));
        const uint32_t mx = mx_pk_max_u16(__builtin_amdgcn_perm(hi[1], hi[0], 0x05040100), __builtin_amdgcn_perm(hi[1], hi[0], 0x07060302));
        const uint32_t sf = __builtin_amdgcn_perm(v1[1], v1[0], 0x07060302);
        if constexpr (l >= 2) {
            constexpr int c = l - 1;
            const uint32_t lo3 = mx_pk_min_u16(mx_pk_min_u16(st.mn[old][p], st.mn[mid][p]), mn);
            const uint32_t hi3 = mx_pk_max_u16(mx_pk_max_u16(st.mx[old][p], st.mx[mid][p]), mx);
            const uint32_t self = st.sf[mid][p];
            const uint32_t z = mx_pk_min_u16(self ^ lo3, self ^ hi3);                            // a zero lane = candidate
            const uint32_t cand = mx_pk_sub_u16(0x00010001u, mx_pk_min_u16(z, 0x00010001u));    // 1 / 0 per lane
            const uint32_t below = mx_pk_min_u16(mx_pk_sub_sat_u16(st.mc2, self), 0x00010001u);  // 1 iff value < min_contrast
            const uint32_t listed = cand & ~below;
            const uint32_t bits = cand | (listed << 1);
            st.out[p] = c == 1 ? bits : (st.out[p] | (bits << (2 * (c - 1))));
        }
        st.mn[old][p] = mn, st.mx[old][p] = mx, st.sf[old][p] = sf;
    }
}

template <int PAIRS>
__device__ __forceinline__ void mx_sites_store(const MxSitesT<PAIRS>& st) {
    if (!st.smap) return;
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
        if (2 * p < st.nk) st.smap[(size_t)(2 * p) * st.mpitch] = (uint8_t)st.out[p];
        if (2 * p + 1 < st.nk) st.smap[(size_t)(2 * p + 1) * st.mpitch] = (uint8_t)(st.out[p] >> 16);
    }
}

// What one lane carries through the six levels.
template <class CFG>
struct MxLane {
    const uint32_t* lp;    // LDS: top-left of the staged halo as seen by this lane (its column, its K half)
    uint8_t* out;          // the frame's octave block
    size_t P;              // bytes per plane
    int nob_live;          // output blocks of the strip that start inside the image (wave-uniform)
    // output flush: results leave the MFMA with the image ROW on the lane (16 bytes of 32 different rows per store
    // instruction: measured 2.2 TB/s of a kernel whose arithmetic takes a third of that time).  Each wave therefore
    // passes a level's G and D strips through its own LDS buffer and stores them with the lanes ALONG the rows:
    // 8 rows x 128 contiguous bytes per instruction.
    uint32_t* wb;          // LDS: this lane's write position (its row, 16 h bytes in) in the wave's output buffer
    const uint32_t* rb;    // LDS: this lane's read position: row lane >> 3, bytes 16 (lane & 7)
    uint32_t off;          // byte offset inside a plane of (strip row lane >> 3, strip column 16 (lane & 7))
    uint32_t pitch8;       // 8 * pitch
    int rows_left;         // rows - that row: read i of this lane (strip row 8 i + (lane >> 3)) stores iff 8 i < rows_left
    bool col_ok;           // that column is inside the image
    uint8_t* nb;           // next octave's base row of this lane (nullptr: none, odd row, or outside)
    int ncols_left;        // next base: ncols - (strip origin + 16 h) / 2
};

// One Gaussian level of a wave's strip.
// tcur: this level's operand fragments (b1[0..NS), a2[0..NS) of level L), loaded by the previous level; tnext: the next level's,
// loaded here BEFORE this level's stores are issued.  vmcnt counts loads and stores together in issue order, so a
// fragment load issued behind the flush waited for every store of the level to be acknowledged by a saturated HBM - the
// waves spent 45 % of their cycles in s_waitcnt.
// FULL: the strip lies wholly inside the image (wave-uniform; all but the last tile row and column): every store is
// unconditional, so the level is straight-line code and the s_waitcnt in front of the next level's first MFMA counts exactly
// the stores that may stay in flight.  Behind conditional stores (exec-masked regions with skip branches) the counter state is
// a range and the wait degrades to vmcnt(0).
template <class CFG, int L, bool EXT, bool FULL>
__device__ __forceinline__ void mx_level(const MxTaps<CFG>* __restrict__ taps, const MxLane<CFG>& ln, uint32_t (&pe)[CFG::NOB][4],
                                         uint32_t (&po)[CFG::NOB][4], MxSites& st, const mx_v4i (&tcur)[2 * CFG::NSMAX],
                                         mx_v4i (&tnext)[2 * CFG::NSMAX]) {
    constexpr int OFF = CFG::off(L), NS = CFG::ns(L), NOB = CFG::NOB, NIN = NOB + NS - 1, R = CFG::R, RWP = CFG::RWP;
    // G = ((C2hi << 8) + C2lo) >> 16 with C2lo starting at 256 * (128 + 32768) + 32768: the biases of the two byte planes
    // (taps sum to 256) + the one round-half-up of A2-iv
    constexpr int kLoInit = 256 * (128 + 32768) + 32768;
    const int lane = threadIdx.x & 63;
    mx_v4i b1[NS], a2[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) b1[s] = tcur[s], a2[s] = tcur[CFG::NSMAX + s];
    mx_v4i hi[NS], lo[NS];
    uint32_t dd[NOB][4] = {};  // the level's D values wait in registers while G passes through the wave's LDS buffer
    // pass 1 on input block ib: columns [-OFF + 32 ib, +32) of the strip, rows [-OFF, -OFF + 32 NS)
    auto pass1 = [&](int ib) {
        mx_v16i c1 = {};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            mx_v4i a;
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = (int)ln.lp[((R - OFF + 32 * s) / 4 + k) * RWP + (R - OFF + 32 * ib)];
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b1[s], c1, 0, 0, 0);
        }
        return c1;
    };
    mx_v16i c1 = pass1(0);
#pragma unroll
    for (int ib = 0; ib < NIN; ++ib) {
        // C1 = H - 32768 in [-32768, 32512]: signed high byte as it is, low byte - 128 (x ^ 0x80)
        const int slot = ib % NS;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)c1[4 * d + 1], (uint32_t)c1[4 * d + 0], 0x05010400);  // (lo0, lo1, hi0, hi1)
            const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)c1[4 * d + 3], (uint32_t)c1[4 * d + 2], 0x05010400);
            lo[slot][d] = (int)(__builtin_amdgcn_perm(t23, t01, 0x05040100) ^ 0x80808080u);
            hi[slot][d] = (int)__builtin_amdgcn_perm(t23, t01, 0x07060302);
        }
        const int ob = ib - (NS - 1);
        const bool have_ob = ib >= NS - 1 && (FULL || ob < ln.nob_live);  // wave-uniform: the block exists and does not lie right of the image
        // ---- pass 2 on output block ob; the NEXT input block's pass 1 is issued right behind it, in front of this block's
        // epilogue: its MFMAs run in the matrix pipe while the vector unit packs and stores, and are finished when the
        // next iteration's split asks for them
        mx_v16i chi = {}, clo;
        if (have_ob) {
#pragma unroll
            for (int v = 0; v < 16; ++v) clo[v] = kLoInit;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                chi = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2[s], hi[(ob + s) % NS], chi, 0, 0, 0);
                clo = __builtin_amdgcn_mfma_i32_32x32x32_i8(a2[s], lo[(ob + s) % NS], clo, 0, 0, 0);
            }
        }
        if (ib + 1 < NIN) c1 = pass1(ib + 1);
        if (!have_ob) continue;
        // ---- epilogue: register v = column 16 h + v of this lane's row --------------------------------------------
        uint32_t g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = ((uint32_t)chi[4 * k + j] << 8) + (uint32_t)clo[4 * k + j];
            const uint32_t e = __builtin_amdgcn_perm(w[2], w[0], 0x0c060c02);  // (G0, G2) in 16-bit lanes
            const uint32_t o = __builtin_amdgcn_perm(w[3], w[1], 0x0c060c02);  // (G1, G3)
            g[k] = __builtin_amdgcn_perm(o, e, 0x06020400);
            if (L > 0)  // D_{L-1} = saturate_u8(G_L - G_{L-1}), GaussPyramid.cpp:197
                dd[ob][k] = __builtin_amdgcn_perm(mx_pk_sub_sat_u16(o, po[ob][k]), mx_pk_sub_sat_u16(e, pe[ob][k]), 0x06020400);
            pe[ob][k] = e;
            po[ob][k] = o;
        }
        *reinterpret_cast<uint4*>(ln.wb + 8 * ob) = make_uint4(g[0], g[1], g[2], g[3]);
        if (CFG::DBUF && L > 0) *reinterpret_cast<uint4*>(ln.wb + CFG::OBUF + 8 * ob) = make_uint4(dd[ob][0], dd[ob][1], dd[ob][2], dd[ob][3]);
        // next octave's base = Gaussian[3] decimated 2:1, INTER_NEAREST (GaussPyramid.cpp:123-126): the even
        // columns of the even rows; 16 ob < ncols_left keeps the 8-byte store inside the row (pitch: multiple of 16)
        if (L == 3 && ln.nb && 16 * ob < ln.ncols_left)
            *reinterpret_cast<uint2*>(ln.nb + 16 * ob) =
                make_uint2(__builtin_amdgcn_perm(pe[ob][1], pe[ob][0], 0x06040200), __builtin_amdgcn_perm(pe[ob][3], pe[ob][2], 0x06040200));
    }
    if constexpr (L < 5) {  // the next level's fragments, in front of this level's stores (see above)
#pragma unroll
        for (int s = 0; s < CFG::ns(L + 1); ++s) tnext[s] = taps->b1[L + 1][s][lane], tnext[CFG::NSMAX + s] = taps->a2[L + 1][s][lane];
    }
    if constexpr (EXT && L > 0) {  // fused lattice scan: DoG level L-1 of the strip is in the wave's D buffer now
        static_assert(CFG::DBUF != 0, "the fused scan reads the D buffer");
        if (st.cdump) st.cdump[(size_t)(L - 1) * st.clevel] = *st.cl;
        if (st.nk > 0) {  // wave-uniform
            mx_sites_level<CFG, L - 1>(st);
            if constexpr (L == 5) mx_sites_store(st);
        }
    }
    // ---- flush: the strip's G rows, then its D rows, from the wave's own LDS buffer with the lanes along the rows ---
    // (one wave's LDS operations execute in order: no barrier between the writes above and these reads, nor between
    // the G reads and the D writes that reuse the buffer, nor before the next level's writes)
    uint8_t* gp = ln.out + (size_t)L * ln.P;
    if (CFG::DBUF) {
        uint8_t* dp = ln.out + (size_t)(VSLAM_NUM_LEVELS + L - 1) * ln.P;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 gv = *reinterpret_cast<const uint4*>(ln.rb + 8 * i * CFG::OBP);
            uint4 dv;
            if (L > 0) dv = *reinterpret_cast<const uint4*>(ln.rb + CFG::OBUF + 8 * i * CFG::OBP);
            if (FULL || (ln.col_ok && 8 * i < ln.rows_left)) {
                *reinterpret_cast<uint4*>(gp + ln.off + i * ln.pitch8) = gv;
                if (L > 0) *reinterpret_cast<uint4*>(dp + ln.off + i * ln.pitch8) = dv;
            }
        }
        return;
    }
    // (four named values, not an array: the conditional stores made hipcc index an array of them dynamically - through scratch)
    const uint4 g0 = *reinterpret_cast<const uint4*>(ln.rb), g1 = *reinterpret_cast<const uint4*>(ln.rb + 8 * CFG::OBP);
    const uint4 g2 = *reinterpret_cast<const uint4*>(ln.rb + 16 * CFG::OBP), g3 = *reinterpret_cast<const uint4*>(ln.rb + 24 * CFG::OBP);
    if (L > 0) {
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) *reinterpret_cast<uint4*>(ln.wb + 8 * ob) = make_uint4(dd[ob][0], dd[ob][1], dd[ob][2], dd[ob][3]);
    }
    if (FULL || ln.col_ok) {
        if (FULL || 0 < ln.rows_left) *reinterpret_cast<uint4*>(gp + ln.off) = g0;
        if (FULL || 8 < ln.rows_left) *reinterpret_cast<uint4*>(gp + ln.off + ln.pitch8) = g1;
        if (FULL || 16 < ln.rows_left) *reinterpret_cast<uint4*>(gp + ln.off + 2 * ln.pitch8) = g2;
        if (FULL || 24 < ln.rows_left) *reinterpret_cast<uint4*>(gp + ln.off + 3 * ln.pitch8) = g3;
    }
    if (L > 0) {
        uint8_t* dp = ln.out + (size_t)(VSLAM_NUM_LEVELS + L - 1) * ln.P;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint4 dv = *reinterpret_cast<const uint4*>(ln.rb + 8 * i * CFG::OBP);
            if (FULL || (ln.col_ok && 8 * i < ln.rows_left)) *reinterpret_cast<uint4*>(dp + ln.off + i * ln.pitch8) = dv;
        }
    }
}

// grid = (ceil(cols/TW), ceil(rows/TH), frames); block = CFG::NT; dynamic LDS = CFG::LDS_BYTES.
// rows / cols arbitrary; `pitch` and `npitch` multiples of 16 (16-byte row stores), planes 16-byte aligned.
// UP2: `base` is the SOURCE frame (rows / 2 x cols / 2, dense rows of `pitch / 2`... see the launch: sstep) and the octave's base
// is its 2x bilinear upsample, formed while the tile is staged (octave 0 of createPyramid).
template <class CFG, bool EXT, bool UP2>
__global__ __launch_bounds__(CFG::NT) void k_pyr_octave_mx(const uint8_t* __restrict__ base, size_t bframe, uint8_t* __restrict__ oct_out,
                                                           size_t pframe, int rows, int cols, int pitch,
                                                           const MxTaps<CFG>* __restrict__ taps, uint8_t* __restrict__ next_base,
                                                           size_t nframe, int nrows, int ncols, int npitch, MxExtArgs ext, int sstep) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    // XCD-aware tile order, as k_pyr_octave: every XCD walks one contiguous run of tiles (neighbours share halo lines in its L2)
    unsigned int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned int per_xcd = (gridDim.x * gridDim.y * gridDim.z) >> 3;
    if (bid < (per_xcd << 3)) bid = (bid & 7u) * per_xcd + (bid >> 3);
    const unsigned int tiles_per_frame = gridDim.x * gridDim.y;
    const unsigned int fz = bid / tiles_per_frame, rem = bid - fz * tiles_per_frame;
    const unsigned int by = rem / gridDim.x, bx = rem - by * gridDim.x;
    const int tile_x0 = bx * CFG::TW, tile_y0 = by * CFG::TH;

    if constexpr (UP2)
        mx_stage_tile_up2<CFG::TW, CFG::TH, CFG::R, CFG::RWP, CFG::NT>(base + fz * bframe, sstep, rows / 2, cols / 2, tile_x0, tile_y0, smem, 0x80808080u);
    else
        mx_stage_tile<CFG::TW, CFG::TH, CFG::R, CFG::RWP, CFG::NT>(base + fz * bframe, rows, cols, pitch, tile_x0, tile_y0, smem, 0x80808080u);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int Yw = 32 * (wave / CFG::NWX), Xw = CFG::SW * (wave % CFG::NWX);
    if (tile_y0 + Yw >= rows || tile_x0 + Xw >= cols) return;  // the whole strip is outside (no barrier follows)
    MxLane<CFG> ln;
    ln.lp = smem + (Yw / 4 + 4 * h) * CFG::RWP + Xw + m;
    ln.out = oct_out + fz * pframe;
    ln.P = (size_t)rows * pitch;
    const int y = tile_y0 + Yw + m, x = tile_x0 + Xw + 16 * h;
    uint32_t* obuf = smem + CFG::STAGE_DWORDS + wave * (1 + CFG::DBUF) * CFG::OBUF;
    ln.wb = obuf + m * CFG::OBP + 4 * h;
    ln.rb = obuf + (lane >> 3) * CFG::OBP + 4 * (lane & 7);
    const int yr = tile_y0 + Yw + (lane >> 3), xr = tile_x0 + Xw + 16 * (lane & 7);
    ln.off = (uint32_t)yr * (uint32_t)pitch + (uint32_t)xr;
    ln.pitch8 = 8u * (uint32_t)pitch;
    ln.rows_left = rows - yr;
    ln.col_ok = xr < cols;
    ln.nob_live = min(CFG::NOB, (cols - (tile_x0 + Xw) + 31) / 32);
    ln.nb = (next_base && (m & 1) == 0 && (y >> 1) < nrows) ? next_base + fz * nframe + (size_t)(y >> 1) * npitch + (x >> 1) : nullptr;
    ln.ncols_left = ncols - (x >> 1);
    MxSites st;
    if (EXT) {
        // lattice rows / columns whose 2 x 2 window lies inside this strip (all wave-uniform but the lane's column)
        const int Y0 = tile_y0 + Yw, X0 = tile_x0 + Xw;
        const int a_lo = (Y0 + 2) / 3, b_lo = (X0 + 2) / 3;
        const int a_first = a_lo + ((Y0 > 0 && 3 * a_lo == Y0) ? 1 : 0);  // 3a = Y0: the window's upper row belongs to the strip above
        const int b_first = b_lo + ((X0 > 0 && 3 * b_lo == X0) ? 1 : 0);
        const int a_last = min((Y0 + 31) / 3, ext.lat_rows - 1);
        st.nk = max(0, a_last - a_first + 1);
        const int b = b_first + lane;
        const bool has = 3 * b <= X0 + CFG::SW - 1 && b < ext.lat_cols;
        const int xa = has ? max(3 * b - 1 - X0, 0) : 0;  // byte of column 3b-1 in a buffered row (column -1 = column 0)
        const uint32_t sbyte = xa & 3;
        st.sel = 0x0c000c00u | sbyte | ((sbyte + ((3 * b - 1 - X0 >= 0) ? 1u : 0u)) << 16);
        const int r0 = 3 * a_first - 1 - Y0;
        const uint8_t* dbuf = reinterpret_cast<const uint8_t*>(obuf + CFG::OBUF) + (xa & ~3);
        st.slr = dbuf + r0 * (CFG::OBP * 4);
        st.sla0 = dbuf + max(r0, 0) * (CFG::OBP * 4);
        st.smap = has ? ext.sitemap + fz * ext.mframe + (size_t)a_first * ext.mpitch + b : nullptr;
        st.mpitch = (uint32_t)ext.mpitch;
        // seam strips: X0 is a seam (this strip holds its right column, image column X0) or X0 + SW is (its left column, X0 + SW - 1)
        constexpr int SEAM = 3 * CFG::SW;
        const bool right_of = X0 > 0 && X0 % SEAM == 0 && X0 / SEAM <= ext.nseams;
        const bool left_of = (X0 + CFG::SW) % SEAM == 0 && (X0 + CFG::SW) / SEAM <= ext.nseams;
        st.cdump = nullptr;
        if ((right_of || left_of) && lane < 32 && Y0 + lane < rows) {
            const int seam = right_of ? X0 / SEAM : (X0 + CFG::SW) / SEAM;
            st.clevel = (uint32_t)ext.nseams * (uint32_t)rows * 2u;
            st.cdump = ext.colmap + fz * ext.cframe + ((size_t)(seam - 1) * rows + (Y0 + lane)) * 2 + (right_of ? 1 : 0);
            st.cl = reinterpret_cast<const uint8_t*>(obuf + CFG::OBUF) + lane * (CFG::OBP * 4) + (right_of ? 0 : CFG::SW - 1);
        }
        const uint32_t mc = (uint32_t)min(max(ext.min_contrast, 0), 256);
        st.mc2 = mc | (mc << 16);
    }
    uint32_t pe[CFG::NOB][4], po[CFG::NOB][4];
    mx_v4i ta[2 * CFG::NSMAX], tb[2 * CFG::NSMAX];
#pragma unroll
    for (int s = 0; s < CFG::ns(0); ++s) ta[s] = taps->b1[0][s][lane], ta[CFG::NSMAX + s] = taps->a2[0][s][lane];
    if (tile_y0 + Yw + 32 <= rows && tile_x0 + Xw + CFG::SW <= cols) {  // wave-uniform
        mx_level<CFG, 0, EXT, true>(taps, ln, pe, po, st, ta, tb);
        mx_level<CFG, 1, EXT, true>(taps, ln, pe, po, st, tb, ta);
        mx_level<CFG, 2, EXT, true>(taps, ln, pe, po, st, ta, tb);
        mx_level<CFG, 3, EXT, true>(taps, ln, pe, po, st, tb, ta);
        mx_level<CFG, 4, EXT, true>(taps, ln, pe, po, st, ta, tb);
        mx_level<CFG, 5, EXT, true>(taps, ln, pe, po, st, tb, ta);
    } else {
        mx_level<CFG, 0, EXT, false>(taps, ln, pe, po, st, ta, tb);
        mx_level<CFG, 1, EXT, false>(taps, ln, pe, po, st, tb, ta);
        mx_level<CFG, 2, EXT, false>(taps, ln, pe, po, st, ta, tb);
        mx_level<CFG, 3, EXT, false>(taps, ln, pe, po, st, tb, ta);
        mx_level<CFG, 4, EXT, false>(taps, ln, pe, po, st, ta, tb);
        mx_level<CFG, 5, EXT, false>(taps, ln, pe, po, st, tb, ta);
    }
}

// ---- the other half of the fused lattice scan ---------------------------------------------------------------------------
// One lane per EIGHT consecutive lattice sites of a lattice row (one byte of each of the six mask words), MX_PACK_ROWS
// lattice rows per workgroup (a thread per site was bound by wave launches: 3.7 M waves per 256-frame step; one wave per
// row piece by workgroup launches).  The lane reads its eight site bytes of every row first (independent loads), gathers
// bit k of the eight bytes into one byte (multiply trick per dword) and stores it at its place in mask word b / 64:
// byte (b mod 64) / 8 = the layout a wave ballot over 64 sites gives, i.e. exactly what k_extrema_w3 writes.
// Sites the octave kernel does not own:
//   * lattice rows whose windows straddle a strip's first image row (a > 0 a multiple of `sh`; sh, sw powers of two, so
//     3a = 0 mod sh <=> a = 0 mod sh) are SKIPPED here: the host runs k_extrema_w3 on exactly those rows;
//   * sites whose window straddles a strip's first column (b > 0 a multiple of `sw`: one lane in sw / 8) are evaluated
//     here from the seam map, the two DoG columns the strips on either side left for every level (MxExtArgs::colmap).
// grid = (ceil(8 wpr / 256), ceil(lat_rows / MX_PACK_ROWS), frames), block 256.
constexpr int MX_PACK_ROWS = 8;
#ifndef VSLAM_MX_OCT0_TU  // (a plain, non-template kernel: defined in one translation unit)
__global__ __launch_bounds__(256) void k_extrema_pack(const uint8_t* __restrict__ sitemap, size_t mframe, int mpitch, const uint8_t* __restrict__ colmap,
                                                      size_t cframe, int nseams, int rows, int lat_rows, int lat_cols, int wpr, int sh, int sw,
                                                      int min_contrast, unsigned long long* __restrict__ bits, unsigned long long* __restrict__ lflags,
                                                      size_t bframe) {
    const int f = blockIdx.z, a0 = blockIdx.y * MX_PACK_ROWS;
    const int g = blockIdx.x * 256 + threadIdx.x;  // byte of a lattice row's mask words
    if (g >= 8 * wpr) return;
    const int b0 = 8 * g, nvalid = min(8, lat_cols - b0);
    const uint8_t* map = sitemap + f * mframe + b0;
    uint2 v[MX_PACK_ROWS];
#pragma unroll
    for (int i = 0; i < MX_PACK_ROWS; ++i) {
        const int a = a0 + i;
        v[i] = make_uint2(0u, 0u);
        if (nvalid > 0 && a < lat_rows) v[i] = *reinterpret_cast<const uint2*>(map + (size_t)min(a, lat_rows - 1) * mpitch);
    }
    if (nvalid > 0 && b0 > 0 && (b0 & (sw - 1)) == 0) {  // site b0 of every row: the window straddles a strip's first column (a seam)
        const size_t lstride = (size_t)nseams * rows * 2;
        const uint8_t* S = colmap + f * cframe + (size_t)(b0 / sw - 1) * rows * 2;  // seam 3 b0 / (3 sw)
        uint32_t mn[MX_PACK_ROWS][2], mx[MX_PACK_ROWS][2], self[MX_PACK_ROWS] = {}, six[MX_PACK_ROWS] = {};
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            uint32_t m0[MX_PACK_ROWS], m1[MX_PACK_ROWS], sf[MX_PACK_ROWS];
#pragma unroll
            for (int i = 0; i < MX_PACK_ROWS; ++i) {
                const int a = min(a0 + i, lat_rows - 1);
                const uint8_t* r0 = S + l * lstride + (size_t)max(3 * a - 1, 0) * 2;
                const uint8_t* r1 = S + l * lstride + (size_t)(3 * a) * 2;
                const uint32_t v00 = r0[0], v01 = r0[1], v10 = r1[0], v11 = r1[1];
                m0[i] = min(min(v00, v01), min(v10, v11));
                m1[i] = max(max(v00, v01), max(v10, v11));
                sf[i] = v11;
            }
#pragma unroll
            for (int i = 0; i < MX_PACK_ROWS; ++i) {
                if (l >= 2) {  // levels l-2, l-1, l are known: centre level c = l-1
                    const uint32_t lo = min(mn[i][0], min(mn[i][1], m0[i])), hi = max(mx[i][0], max(mx[i][1], m1[i]));
                    const uint32_t cand = (self[i] == lo || self[i] == hi) ? 1u : 0u;
                    six[i] |= (cand | ((cand && (int)self[i] >= min_contrast) ? 2u : 0u)) << (2 * (l - 2));
                }
                mn[i][0] = mn[i][1], mx[i][0] = mx[i][1];
                mn[i][1] = m0[i], mx[i][1] = m1[i];
                self[i] = sf[i];  // the centre value of the NEXT test is this level's
            }
        }
#pragma unroll
        for (int i = 0; i < MX_PACK_ROWS; ++i) v[i].x = (v[i].x & ~0xffu) | six[i];
    }
    uint8_t* bb = bits ? reinterpret_cast<uint8_t*>(bits + f * bframe) : nullptr;
    uint8_t* lb = reinterpret_cast<uint8_t*>(lflags + f * bframe);
#pragma unroll
    for (int i = 0; i < MX_PACK_ROWS; ++i) {
        const int a = a0 + i;
        if (a >= lat_rows) break;
        if (a > 0 && (a & (sh - 1)) == 0) continue;  // k_extrema_w3 writes this row's words
        uint32_t xl = v[i].x, xh = v[i].y;  // sites b0..b0+3, b0+4..b0+7, one byte each
        if (nvalid < 8) {  // the row's last sites (or none): what lies behind them in the map is not a site
            const unsigned long long keep = nvalid > 0 ? ~0ull >> (8 * (8 - nvalid)) : 0ull;
            xl &= (uint32_t)keep, xh &= (uint32_t)(keep >> 32);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            // bit k of bytes 0..3 -> bits 28..31 of the product (the partial products below bit 28 sum to less than 2^24)
            const uint32_t lo4 = (((xl >> k) & 0x01010101u) * 0x10204080u) >> 28, hi4 = (((xh >> k) & 0x01010101u) * 0x10204080u) >> 28;
            const uint8_t byte = (uint8_t)(lo4 | (hi4 << 4));
            const size_t at = (((size_t)(k >> 1) * lat_rows + a) * wpr) * 8 + g;
            if (k & 1) lb[at] = byte;
            else if (bb) bb[at] = byte;
        }
    }
}

#endif  // !VSLAM_MX_OCT0_TU

// Host side: the quantised taps as MFMA operand fragments.
//   b1[l][s][lane] byte j: pass-1 B[k][y'] with y' = lane & 31 the output row and k = 16 (lane >> 5) + j the input row
//                          -off + 32 s + k relative to the output block: tap index (input - output + r).
//   a2[l][s][lane] byte j: pass-2 A[m][k]: row m = lane & 31 is output column 16 ((m >> 2) & 1) + (m & 3) + 4 (m >> 3) (so that
//                          C register v of lane half h is column 16 h + v), k = 16 (lane >> 5) + j is the input column
//                          -off + 32 s + rho(lane >> 5, j), rho(h, j) = (j & 3) + 8 (j >> 2) + 4 h: the row of C1 register j.
template <class CFG>
static bool mx_pack_taps(const uint16_t* const t[6], MxTaps<CFG>& out) {
    std::memset(&out, 0, sizeof(out));
    for (int l = 0; l < 6; ++l) {
        const int n = CFG::n(l), r = n / 2, off = CFG::off(l);
        auto tap = [&](int idx) -> int { return idx >= 0 && idx < n ? (int)t[l][idx] : 0; };
        for (int k = 0; k < n; ++k)
            if (t[l][k] > 127) return false;  // signed 8-bit operands
        for (int s = 0; s < CFG::ns(l); ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 31, h = lane >> 5;
                int8_t b[16], a[16];
                const int xout = 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3);
                for (int j = 0; j < 16; ++j) {
                    b[j] = (int8_t)tap((-off + 32 * s + 16 * h + j) - m + r);
                    a[j] = (int8_t)tap((-off + 32 * s + (j & 3) + 8 * (j >> 2) + 4 * h) - xout + r);
                }
                std::memcpy(&out.b1[l][s][lane], b, 16);
                std::memcpy(&out.a2[l][s][lane], a, 16);
            }
    }
    return true;
}

// The reference's fixed pyramid (sigma0 = 1.6): zero-trimmed widths as PyrCfgOct0 / PyrCfgOct1.
using MxCfgOct0 = MxCfg<128, 128, 1, 1, 9, 13, 15, 19, 23, 29>;
using MxCfgOct1 = MxCfg<128, 128, 1, 1, 19, 23, 29, 37, 45, 57>;
// octaves 2 and 3 (the default path runs them through the strip kernels and a u16 scratch in HBM): K windows of 3 - 5
// resp. 4 - 8 steps per level; octave 3's staged tile (halo 112) fills most of a CU's LDS
using MxCfgOct2 = MxCfg<128, 128, 1, 0, 37, 45, 57, 71, 89, 111>;
using MxCfgOct3 = MxCfg<128, 128, 1, 0, 71, 89, 111, 141, 177, 223>;

}  // namespace vslam
