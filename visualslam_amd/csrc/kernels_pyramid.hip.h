// K-D1: fused scale-space octave for gfx950.
//
// One launch turns the octave base (u8) into the octave's six Gaussian images and five DoG
// images -- GaussVector + Diff_of_Gauss of GaussPyramid.cpp:166-200 -- reading the base
// once per tile (plus halo) and writing each of the 11 images exactly once.  Nothing
// intermediate touches HBM: the separable passes live in LDS.
//
// Arithmetic: G = (sum_y sum_x ty*tx*p + 32768) >> 16 is an exact
// integer expression, so any evaluation order gives the reference result.  Per level:
//   pass 1 (vertical)   h(y,x) = sum_k t[k]*p(y-r+k, x)     u8 x u8 taps  -> v_dot4_u32_u8
//   pass 2 (horizontal) acc    = sum_k t[k]*h(y, x-r+k)     u16 x u16 taps -> v_dot2_u32_u16
// (all quantised taps are <= 64 for sigma >= 1.6, h <= 65280, acc < 2^24).  No MFMA.
//
// Data layout in LDS (one 256-thread workgroup = one TW x TH output tile, 128 x 64 or 256 x 32):
//   rp  [(TH+2R)/4][RWP]  dwords: the base tile with halo R, BYTE-TRANSPOSED so that one dword
//        holds 4 vertically adjacent pixels of one column -- the operand shape of a vertical
//        dot4.  Filled once per tile (v_perm 4x4 transposes); image borders are resolved at
//        fill time by storing the BORDER_REFLECT_101 extension (valid for every level because
//        the taps are symmetric).
//   hp  [TH][HPP]         dwords: pass-1 output of the current level as u16 PAIRS of
//        horizontally adjacent columns -- the operand shape of a horizontal dot2.
// Taps are wave-uniform scalar loads (SGPR operands) from a small device table: t4[l][o][m] =
// 4 taps packed as bytes for window alignment o, tp[l][e] = taps (e-1, e) packed as u16.  The
// loads of a pass are pinned behind an opaque zero defined at the start of that pass, so only
// one pass's taps are live at a time (letting the scheduler hoist all 6 levels' loads to the
// kernel entry spilled ~400 SGPRs through v_readlane).  Kernel widths are template parameters
// (the zero-trimmed effective widths), so every loop is fully unrolled and all-zero tap words
// are skipped at compile time.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_generic.hip.h"

namespace vslam {

typedef unsigned short us2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot4(a, b, c, false);
}
__device__ __forceinline__ uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b), c, false);
}
// packed u16 saturating subtract (v_pk_sub_u16 clamp)
__device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2_t, a), __builtin_bit_cast(us2_t, b)));
}

template <int TW_, int TH_, int N0, int N1, int N2, int N3, int N4, int N5>
struct PyrCfg {
    static constexpr int TW = TW_, TH = TH_;
    static constexpr int n(int l) { return l == 0 ? N0 : l == 1 ? N1 : l == 2 ? N2 : l == 3 ? N3 : l == 4 ? N4 : N5; }
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int NMAX = cmax(cmax(cmax(N0, N1), cmax(N2, N3)), cmax(N4, N5));
    static constexpr int R = (NMAX / 2 + 15) & ~15;  // halo, multiple of 16 (16-byte staging loads)
    static constexpr int RQ = (TH + 2 * R) / 4;    // row quads of the raw tile
    static constexpr int RW = TW + 2 * R;          // raw tile width (pixels == rp dwords per row quad)
    static constexpr int RWP = RW + 4;             // padded pitch (pass 1 may read 1 dword group past RW)
    static constexpr int r(int l) { return n(l) / 2; }
    static constexpr int delta(int l) { return (R - r(l)) & 3; }
    static constexpr int A(int l) { return (R - r(l)) & ~3; }
    static constexpr int ncg(int l) { return (TW + 2 * r(l) + delta(l) + 3) / 4; }   // h column groups of 4
    static constexpr int m1(int l) { return ((delta(l) + 3 + n(l) - 1) >> 2) + 1; } // rp dwords per pass-1 item column (4 rows)
    static constexpr int pmax(int l) { return (7 + delta(l) + 2 * r(l)) / 2; }      // last u16 pair a pass-2 item reads
    static constexpr int nb(int l) { return pmax(l) / 4 + 1; }                      // b128 reads per row
    static constexpr int hpp_l(int l) { return cmax(2 * ncg(l), TW / 2 - 4 + 4 * nb(l)); }
    static constexpr int HPP =
        (cmax(cmax(cmax(hpp_l(0), hpp_l(1)), cmax(hpp_l(2), hpp_l(3))), cmax(hpp_l(4), hpp_l(5))) + 15) & ~15;
    static constexpr int T4M = ((3 + NMAX - 1) >> 2) + 1;  // tap dwords per alignment
    static constexpr int TPM = NMAX + 1;                   // tap pairs
    static constexpr int LDS_BYTES = (RQ * RWP + TH * HPP) * 4;
    static constexpr int NT = (TH / 4) * (TW / 8);  // threads per workgroup: one pass-2 item (8 columns x 4 rows) each
    static constexpr int NW = NT / 64;
    static_assert(TW % 16 == 0 && TH % 8 == 0 && NT % 256 == 0 && NT <= 1024,
                  "a multiple of four waves, one per SIMD: 384-thread workgroups (384 x 32 tiles) ran 16 % slower, their six waves sit 2-2-1-1");
    static_assert((N0 & 1) && (N1 & 1) && (N2 & 1) && (N3 & 1) && (N4 & 1) && (N5 & 1), "odd kernels");
};

template <class CFG>
struct PyrTaps {
    uint32_t t4[6][4][CFG::T4M];
    uint32_t tp[6][CFG::TPM];
};

// One Gaussian level of the tile: pass 1 into hp, pass 2 into registers, pack, DoG, store.
template <class CFG, int L>
__device__ __forceinline__ void pyr_level(const PyrTaps<CFG>* __restrict__ taps, const uint32_t* __restrict__ rp,
                                          uint32_t* __restrict__ hp, uint8_t* __restrict__ out, size_t P, int cols,
                                          int pitch, int rows, int tile_x0, int tile_y0, uint32_t (&prev_e)[4][2],
                                          uint32_t (&prev_o)[4][2], uint8_t* __restrict__ next_base, int nrows,
                                          int ncols, int npitch, const uint32_t (&row_off)[4]) {
    constexpr int n = CFG::n(L), dl = CFG::delta(L), A = CFG::A(L);
    constexpr int NCG = CFG::ncg(L), M = CFG::m1(L), NB = CFG::nb(L);
    constexpr int RWP = CFG::RWP, HPP = CFG::HPP, TH = CFG::TH;
    const int tid = threadIdx.x;
    uint32_t z1;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z1));  // pins this pass's tap loads here
    const uint32_t* __restrict__ t4 = &taps->t4[L][0][0] + z1;
    const uint4* __restrict__ rp4 = reinterpret_cast<const uint4*>(rp);

    // ---- pass 1: vertical, item = 4 h-columns x 4 rows -------------------------------------
    // (4-row items: the (TW + 2r)/4 x TH/4 items of a level fill 8.3 - 10 waves, so the partly
    // filled last wave costs 0 - 8 % of the pass; with 8-row items it was 5 waves for 4.25 - 5)
    // The items do not fill a whole number of 256-thread rounds.  Left alone, the partial last round
    // always falls on wave 0 - 18 item rounds per tile against 12 for each other wave - and the waves of
    // a workgroup sit on different SIMDs, so one SIMD of the CU carries the excess of every resident
    // workgroup.  Rotating the partial round over the waves with the level evens it out (+1.1 % frames/s).
    constexpr int NT = CFG::NT;
    constexpr int N1 = NCG * (TH / 4), FULL1 = N1 / NT * NT;
    for (int k = 0; k < (N1 + NT - 1) / NT; ++k) {
        int it = tid + NT * k;
        if (k == N1 / NT) {
            it = FULL1 + (tid + NT - 64 * (L % CFG::NW)) % NT;
            if (it >= N1) break;
        }
        const int cg = it % NCG, rq = it / NCG;
        uint32_t acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] = 0;
        const uint4* col = rp4 + (rq + A / 4) * (RWP / 4) + cg + A / 4;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const uint4 v = col[m * (RWP / 4)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int s = dl + j, o = s & 3, mm = m - (s >> 2);
                if (mm >= 0 && mm <= ((o + n - 1) >> 2)) {
                    const uint32_t t = t4[o * CFG::T4M + mm];
                    acc[j][0] = udot4(v.x, t, acc[j][0]);
                    acc[j][1] = udot4(v.y, t, acc[j][1]);
                    acc[j][2] = udot4(v.z, t, acc[j][2]);
                    acc[j][3] = udot4(v.w, t, acc[j][3]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint2 w;
            w.x = __builtin_amdgcn_perm(acc[j][1], acc[j][0], 0x05040100);  // (acc0, acc1) as u16 pair: one v_perm, not shift + or
            w.y = __builtin_amdgcn_perm(acc[j][3], acc[j][2], 0x05040100);
            *reinterpret_cast<uint2*>(hp + (4 * rq + j) * HPP + 2 * cg) = w;
        }
    }
    __syncthreads();

    // ---- pass 2: horizontal, item = 8 columns x 4 rows (exactly one per thread) -------------
    uint32_t z2;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z2));
    const uint32_t* __restrict__ tp = &taps->tp[L][0] + z2;
    const int xg = tid % (CFG::TW / 8), rg = tid / (CFG::TW / 8);
    uint32_t acc[4][8];
#pragma unroll
    for (int jr = 0; jr < 4; ++jr)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[jr][j] = 32768u;  // the one round-half-up of A2-iv
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        // Wide kernels (octave 1: up to 58 tap pairs per level) do not fit the SGPR file next to the
        // rest of the pass and were reloaded through v_readlane (410 of them).  There the pairs
        // are pinned per 8-column block instead: block b reads tp[8b-7 .. 8b+7], loaded behind an
        // opaque zero defined one block earlier, so at most two blocks' pairs are live.
        uint32_t zb = 0;
        if (CFG::TPM > 40 && (b & 1) == 0) asm volatile("s_mov_b32 %0, 0" : "=s"(zb));
        const uint32_t* __restrict__ tpb = CFG::TPM > 40 ? tp + zb : tp;
#pragma unroll
        for (int jr = 0; jr < 4; ++jr) {
            const uint4 v = *reinterpret_cast<const uint4*>(hp + (4 * rg + jr) * HPP + 4 * xg + 4 * b);
            // keep the access one conflict-free ds_read_b128: if only part of v is used (window
            // edges) hipcc narrows it to ds_read2_b32, which at this 16-byte lane stride is a
            // 4-way bank conflict
            asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = 2 * (4 * b + pp) - j - dl + 1;  // tap pair (e-1, e)
                    if (e >= 0 && e <= n) acc[jr][j] = udot2(vv[pp], tpb[e], acc[jr][j]);
                }
            }
        }
    }
    __syncthreads();  // hp is free for the next level's pass 1

    // ---- pack (G = acc >> 16 is byte 2 of acc), DoG, store ----------------------------------
    const int x = tile_x0 + 8 * xg;
#pragma unroll
    for (int jr = 0; jr < 4; ++jr) {
        const int y = tile_y0 + 4 * rg + jr;
        uint32_t g[2], d[2];
#pragma unroll
        for (int hw = 0; hw < 2; ++hw) {
            // G = acc >> 16 is byte 2 of each accumulator.  Pick the even / odd pixels straight
            // into 16-bit lanes (the shape the saturating subtract wants), then interleave.
            const uint32_t e = __builtin_amdgcn_perm(acc[jr][4 * hw + 2], acc[jr][4 * hw + 0], 0x0c060c02);  // (G0, G2)
            const uint32_t o = __builtin_amdgcn_perm(acc[jr][4 * hw + 3], acc[jr][4 * hw + 1], 0x0c060c02);  // (G1, G3)
            g[hw] = __builtin_amdgcn_perm(o, e, 0x06020400);  // bytes (e0, o0, e1, o1): interleave in one v_perm
            if (L > 0)  // D_{L-1} = saturate_u8(G_L - G_{L-1}), GaussPyramid.cpp:197
                d[hw] = __builtin_amdgcn_perm(pk_sub_sat_u16(o, prev_o[jr][hw]), pk_sub_sat_u16(e, prev_e[jr][hw]), 0x06020400);
            prev_e[jr][hw] = e;
            prev_o[jr][hw] = o;
        }
        if (y < rows && x < cols) {
            // 32-bit offset inside a wave-uniform plane pointer (a frame's octave block is far below
            // 4 GB): scalar base + VGPR offset addressing, no 64-bit address arithmetic per store.
            // The last 8-column group may end in the row padding.
            const uint32_t off = row_off[jr];  // y * pitch + x: the same for every level, formed once per thread (k_pyr_octave)
            uint8_t* gp = out + (size_t)L * P;
            *reinterpret_cast<uint2*>(gp + off) = make_uint2(g[0], g[1]);
            if (L > 0) {
                uint8_t* dp = out + (size_t)(VSLAM_NUM_LEVELS + L - 1) * P;
                *reinterpret_cast<uint2*>(dp + off) = make_uint2(d[0], d[1]);
            }
            // next octave's base = Gaussian[3] decimated 2:1, INTER_NEAREST (GaussPyramid.cpp:123-126):
            // pixel (2y', 2x'); tile origins and (jr, x) are even, so it is the even bytes of even rows
            if (L == 3 && next_base && (jr & 1) == 0 && (y >> 1) < nrows && (x >> 1) < ncols)
                *reinterpret_cast<uint32_t*>(next_base + ((uint32_t)(y >> 1) * (uint32_t)npitch + (uint32_t)(x >> 1))) =
                    __builtin_amdgcn_perm(g[1], g[0], 0x06040200);
        }
    }
}

// grid = (ceil(cols/TW), ceil(rows/TH), frames); block = CFG::NT (256 for the 128 x 64 and 256 x 32 tiles);
// dynamic LDS = CFG::LDS_BYTES.
// rows / cols arbitrary; `pitch` (row pitch of the base and of every output plane) and `npitch`
// (next base) are multiples of 16 and >= the width rounded up to 8 (8-byte row stores).
template <class CFG>
__global__ __launch_bounds__(CFG::NT) void k_pyr_octave(const uint8_t* __restrict__ base, size_t bframe,
                                                     uint8_t* __restrict__ oct_out, size_t pframe, int rows, int cols,
                                                     int pitch, const PyrTaps<CFG>* __restrict__ taps,
                                                     uint8_t* __restrict__ next_base, size_t nframe, int nrows, int ncols,
                                                     int npitch) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* rp = smem;
    uint32_t* hp = smem + CFG::RQ * CFG::RWP;
    constexpr int R = CFG::R, RW = CFG::RW, RWP = CFG::RWP, RQ = CFG::RQ;
    const int tid = threadIdx.x;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an
    // L2), so the linear id is remapped to give every XCD one contiguous run of tiles - row-major
    // neighbours, which share their halo rows and columns, then meet in the same 4 MB L2 instead
    // of each fetching the halo from HBM.  Placement is a speed matter only.
    unsigned int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned int per_xcd = (gridDim.x * gridDim.y * gridDim.z) >> 3;
    if (bid < (per_xcd << 3)) bid = (bid & 7u) * per_xcd + (bid >> 3);
    const unsigned int tiles_per_frame = gridDim.x * gridDim.y;
    const unsigned int fz = bid / tiles_per_frame, rem = bid - fz * tiles_per_frame;
    const unsigned int by = rem / gridDim.x, bx = rem - by * gridDim.x;
    const int tile_x0 = bx * CFG::TW, tile_y0 = by * CFG::TH;
    const uint8_t* src = base + fz * bframe;
    uint8_t* out = oct_out + fz * pframe;
    const size_t P = (size_t)rows * pitch;

    // ---- stage the base tile, byte-transposed ------------------------------------------------
    const bool interior = tile_x0 - R >= 0 && tile_x0 + CFG::TW + R <= cols && tile_y0 - R >= 0 &&
                          tile_y0 + CFG::TH + R <= rows;
    if (interior) {
        // 16 pixels x 4 rows per item: 16-byte coalesced loads, four 4x4 byte transposes, four
        // 16-byte LDS stores (RW is a multiple of 16, tile origin - R is 16-byte aligned)
        for (int it = tid; it < RQ * (RW / 16); it += CFG::NT) {
            const int yq = it / (RW / 16), xs = it - yq * (RW / 16);
            const uint8_t* p = src + (size_t)(tile_y0 - R + 4 * yq) * pitch + (tile_x0 - R + 16 * xs);
            uint4 a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = *reinterpret_cast<const uint4*>(p + (size_t)k * pitch);
            const uint32_t* aw[4] = {&a[0].x, &a[1].x, &a[2].x, &a[3].x};
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // dword q of each row = pixels 4q..4q+3
                const uint32_t r0 = aw[0][q], r1 = aw[1][q], r2 = aw[2][q], r3 = aw[3][q];
                const uint32_t p01l = __builtin_amdgcn_perm(r1, r0, 0x05010400), p01h = __builtin_amdgcn_perm(r1, r0, 0x07030602);
                const uint32_t p23l = __builtin_amdgcn_perm(r3, r2, 0x05010400), p23h = __builtin_amdgcn_perm(r3, r2, 0x07030602);
                uint4 t;
                t.x = __builtin_amdgcn_perm(p23l, p01l, 0x05040100);
                t.y = __builtin_amdgcn_perm(p23l, p01l, 0x07060302);
                t.z = __builtin_amdgcn_perm(p23h, p01h, 0x05040100);
                t.w = __builtin_amdgcn_perm(p23h, p01h, 0x07060302);
                *reinterpret_cast<uint4*>(rp + yq * RWP + 16 * xs + 4 * q) = t;
            }
        }
    } else if (cols >= 4 && R < cols && tile_x0 + CFG::TW + R - 1 <= 2 * (cols - 1) && R < rows && tile_y0 + CFG::TH + R - 1 <= 2 * (rows - 1)) {
        // Border tiles whose halo reaches at most ONE reflection on either side (16 % of the tiles of a 3840 x 2160 octave,
        // 24 % of a 1920 x 1080 one).  Round 5, from the matrix path's staging: four pixels at columns x .. x+3 under
        // BORDER_REFLECT_101 always lie within four consecutive bytes of the row (a forward run, a mirrored run, or a run folded
        // around column 0 / cols-1), so every case is ONE unaligned dword load at `base` and one v_perm whose selector holds
        // the four byte positions relative to base - no divergent paths, four loads per item instead of sixteen byte loads.
        auto f1 = [](int x, int n) { return x < 0 ? -x : (x >= n ? 2 * (n - 1) - x : x); };
#pragma unroll 2
        for (int it = tid; it < RQ * (RW / 4); it += CFG::NT) {
            const int yq = it / (RW / 4), xq = it - yq * (RW / 4);
            const int gy = tile_y0 - R + 4 * yq, gx = tile_x0 - R + 4 * xq;
            const int p0 = f1(gx, cols), p1 = f1(gx + 1, cols), p2 = f1(gx + 2, cols), p3 = f1(gx + 3, cols);
            const int b0 = min(min(min(p0, p1), min(p2, p3)), cols - 4);
            const uint32_t sel = (uint32_t)(p0 - b0) | ((uint32_t)(p1 - b0) << 8) | ((uint32_t)(p2 - b0) << 16) | ((uint32_t)(p3 - b0) << 24);
            uint32_t a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                uint32_t v;
                __builtin_memcpy(&v, src + (size_t)f1(gy + k, rows) * pitch + b0, 4);
                a[k] = __builtin_amdgcn_perm(0u, v, sel);
            }
            const uint32_t p01l = __builtin_amdgcn_perm(a[1], a[0], 0x05010400), p01h = __builtin_amdgcn_perm(a[1], a[0], 0x07030602);
            const uint32_t p23l = __builtin_amdgcn_perm(a[3], a[2], 0x05010400), p23h = __builtin_amdgcn_perm(a[3], a[2], 0x07030602);
            uint4 t;
            t.x = __builtin_amdgcn_perm(p23l, p01l, 0x05040100);
            t.y = __builtin_amdgcn_perm(p23l, p01l, 0x07060302);
            t.z = __builtin_amdgcn_perm(p23h, p01h, 0x05040100);
            t.w = __builtin_amdgcn_perm(p23h, p01h, 0x07060302);
            *reinterpret_cast<uint4*>(rp + yq * RWP + 4 * xq) = t;
        }
    } else {
        // tiny images (a halo wider than the image: repeated reflection): one dword x 4 rows per item, BORDER_REFLECT_101 resolved per byte
        for (int it = tid; it < RQ * (RW / 4); it += CFG::NT) {
            const int yq = it / (RW / 4), xq = it - yq * (RW / 4);
            const int gy = tile_y0 - R + 4 * yq, gx = tile_x0 - R + 4 * xq;
            uint32_t a[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint8_t* row = src + (size_t)reflect101(gy + k, rows) * pitch;
                a[k] = (uint32_t)row[reflect101(gx, cols)] | ((uint32_t)row[reflect101(gx + 1, cols)] << 8) |
                       ((uint32_t)row[reflect101(gx + 2, cols)] << 16) | ((uint32_t)row[reflect101(gx + 3, cols)] << 24);
            }
            const uint32_t p01l = __builtin_amdgcn_perm(a[1], a[0], 0x05010400), p01h = __builtin_amdgcn_perm(a[1], a[0], 0x07030602);
            const uint32_t p23l = __builtin_amdgcn_perm(a[3], a[2], 0x05010400), p23h = __builtin_amdgcn_perm(a[3], a[2], 0x07030602);
            uint4 t;
            t.x = __builtin_amdgcn_perm(p23l, p01l, 0x05040100);
            t.y = __builtin_amdgcn_perm(p23l, p01l, 0x07060302);
            t.z = __builtin_amdgcn_perm(p23h, p01h, 0x05040100);
            t.w = __builtin_amdgcn_perm(p23h, p01h, 0x07060302);
            *reinterpret_cast<uint4*>(rp + yq * RWP + 4 * xq) = t;
        }
    }
    // the 4 pad dwords per row quad feed only h columns that pass 2 never reads (integers:
    // any value is harmless), so they are left uninitialised.
    __syncthreads();

    uint8_t* nb = next_base ? next_base + fz * nframe : nullptr;
    uint32_t prev_e[4][2], prev_o[4][2];
    // byte offsets of this thread's four output rows inside a plane (pass 2 / epilogue mapping of pyr_level):
    // formed once per thread, not per level: the offsets do not depend on the level (instruction count, not a slow multiply)
    uint32_t row_off[4];
    {
        const int xg = tid % (CFG::TW / 8), rg = tid / (CFG::TW / 8);
#pragma unroll
        for (int jr = 0; jr < 4; ++jr)
            row_off[jr] = (uint32_t)(tile_y0 + 4 * rg + jr) * (uint32_t)pitch + (uint32_t)(tile_x0 + 8 * xg);
    }
    pyr_level<CFG, 0>(taps, rp, hp, out, P, cols, pitch, rows, tile_x0, tile_y0, prev_e, prev_o, nb, nrows, ncols, npitch, row_off);
    pyr_level<CFG, 1>(taps, rp, hp, out, P, cols, pitch, rows, tile_x0, tile_y0, prev_e, prev_o, nb, nrows, ncols, npitch, row_off);
    pyr_level<CFG, 2>(taps, rp, hp, out, P, cols, pitch, rows, tile_x0, tile_y0, prev_e, prev_o, nb, nrows, ncols, npitch, row_off);
    pyr_level<CFG, 3>(taps, rp, hp, out, P, cols, pitch, rows, tile_x0, tile_y0, prev_e, prev_o, nb, nrows, ncols, npitch, row_off);
    pyr_level<CFG, 4>(taps, rp, hp, out, P, cols, pitch, rows, tile_x0, tile_y0, prev_e, prev_o, nb, nrows, ncols, npitch, row_off);
    pyr_level<CFG, 5>(taps, rp, hp, out, P, cols, pitch, rows, tile_x0, tile_y0, prev_e, prev_o, nb, nrows, ncols, npitch, row_off);
}

// Host side: pack quantised taps into the operand shapes described at the top.
template <class CFG>
static void pyr_pack_taps(const uint16_t* const t[6], PyrTaps<CFG>& out) {
    for (int l = 0; l < 6; ++l) {
        const int n = CFG::n(l);
        for (int o = 0; o < 4; ++o)
            for (int m = 0; m < CFG::T4M; ++m) {
                uint32_t w = 0;
                for (int b = 0; b < 4; ++b) {
                    const int k = 4 * m + b - o;
                    if (k >= 0 && k < n) w |= (uint32_t)(t[l][k] & 0xff) << (8 * b);
                }
                out.t4[l][o][m] = w;
            }
        for (int e = 0; e < CFG::TPM; ++e) {
            const uint32_t lo = (e - 1 >= 0 && e - 1 < n) ? t[l][e - 1] : 0, hi = e < n ? t[l][e] : 0;
            out.tp[l][e] = lo | (hi << 16);
        }
    }
}

// The reference's fixed pyramid (sigma0 = 1.6, Diff_of_Gauss.cpp:743): the zero-trimmed widths
// of the SURVEY.md Appendix C kernels (11,13,17,21,25,31 / 21,25,31,39,49,63) for octaves 0, 1.
// Two tile shapes per tap set: 256 x 32 halves the halo share of the vertical pass ((TW + 2r)/TW) and
// is taken when it covers the octave with no more tile area than 128 x 64 (3840 x 2160: 15 x 68 tiles
// either way); 1920 x 1080 is 7.5 tiles of 256 wide, so octave 1 of a 1080p frame stays at 128 x 64.
using PyrCfgOct0 = PyrCfg<128, 64, 9, 13, 15, 19, 23, 29>;
using PyrCfgOct1 = PyrCfg<128, 64, 19, 23, 29, 37, 45, 57>;
using PyrCfgOct0W = PyrCfg<256, 32, 9, 13, 15, 19, 23, 29>;
using PyrCfgOct1W = PyrCfg<256, 32, 19, 23, 29, 37, 45, 57>;

}  // namespace vslam
