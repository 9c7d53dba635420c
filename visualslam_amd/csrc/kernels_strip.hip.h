// K-D2 / K-D3: separable Gaussian levels for the coarse octaves (wide kernels on small images).
//
// Octaves >= 2 of the reference pyramid have 39..245-tap kernels on 960x540 / 480x270 images (and 155..977 taps on
// 240x135 / 120x68 when the octave count is the reference's automatic one)
// (SURVEY.md Appendix C): a 2-D tile would be mostly halo.  Here each pass keeps the WHOLE
// extent of its filtering axis in LDS, so there is no halo recomputation at all:
//   k_gauss_v_strip: one workgroup = a 64-column strip x all rows of the octave base, staged
//       byte-transposed (4 vertically adjacent pixels per dword); vertical pass with
//       v_dot4_u32_u8 for all 6 levels; writes the u16 row sums h[level] (scratch, L2/MALL).
//   k_gauss_h_strip: one workgroup = SH rows x all columns of h[level], staged as u16 pairs with
//       the BORDER_REFLECT_101 extension; horizontal pass with v_dot2_u32_u16; packs G, forms the
//       saturating DoG against the previous level and writes both.
// Kernel widths are runtime values: the tap operands are wave-uniform scalar loads from a
// table prepared on the host (StripTaps).  Arithmetic is the exact integer form of SURVEY A2.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_pyramid.hip.h"

namespace vslam {

constexpr int STRIP_W = 64;        // columns per vertical-pass workgroup
constexpr int STRIP_MAXN = 2047;   // widest kernel (taps, after zero-tail trimming)
// Kernels up to 2047 taps (round 5; 245 before): the reference's second constructor gives a 1080p frame six octaves, and
// octaves 4 and 5 (240 x 135, 120 x 68) have kernels of up to 489 and 977 taps - several times wider than the image (4K: a seventh octave, 1955 taps on 120 x 68), which
// the repeated BORDER_REFLECT_101 of the staging loops already handles.  On the one-thread-per-pixel generic kernels those two
// octaves cost 8 x octave 3 (4.1 + 3.8 ms per 64-frame batch against 0.5).
constexpr int STRIP_MAXM = 514;    // tap dword groups per level: M = (2047 + 6) >> 2 = 513
constexpr int STRIP_MAXTP = 2072;  // padded u16 tap-pair table: 8 + (2047+1) + 16, multiple of 8 (the last block of a level reads 8 (nb-1) + 15 <= 2063)
// The row sums the vertical pass hands to the horizontal pass carry +128 each: the taps of a level sum
// to exactly 256 (strip_pack_taps checks it), so sum tx * (h + 128) = sum tx * h + 32768 - the rounding
// constant of SURVEY A2-iv arrives with the data and the horizontal accumulators start from the first
// product instead of from a constant.  h <= 255 * 256 = 65280, so h + 128 still fits 16 bits.
constexpr uint32_t STRIP_HBIAS = 128;

// Device-resident tap tables of one octave (6 levels).
struct StripTaps {
    // v4[l][m][j]: bytes b = taps[4m + b - j]  (window alignment j = 0..3)
    uint32_t v4[VSLAM_NUM_LEVELS][STRIP_MAXM][4];
    // hp[l][8 + e] = (taps[e-1], taps[e]) as u16 pair, zero outside e in [0, n]
    uint32_t hp[VSLAM_NUM_LEVELS][STRIP_MAXTP];
    int n[VSLAM_NUM_LEVELS];
};

// ------------------------------------------------------------------------------ vertical pass
// grid = (ceil(cols/64), level split 1|2|3|6, frames); dynamic LDS = rhq * 64 * 4 bytes, rhq = (rows4 + 2*RM + 16)/4.
// Any width; `pitch` (a multiple of 16) is the row pitch of the base in bytes and of the scratch in
// elements, so the last 4-column group may run into the row padding.
// h: [frame][level][rows][pitch] u16.
__global__ __launch_bounds__(256) void k_gauss_v_strip(const uint8_t* __restrict__ base, size_t bframe,
                                                        uint16_t* __restrict__ h, size_t hframe, int rows, int cols,
                                                        int pitch, int RM, int rhq, const StripTaps* __restrict__ taps) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* rp = smem;  // [rhq][64]: dword (yq, c) = raw rows 4yq..4yq+3 (ry = y + RM) of column c
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * STRIP_W;
    const uint8_t* src = base + blockIdx.z * bframe;
    const size_t P = (size_t)rows * pitch;

    for (int it = tid; it < rhq * (STRIP_W / 4); it += 256) {
        const int yq = it >> 4, xq = it & 15;
        const int gx = x0 + 4 * xq;
        uint32_t a[4] = {0, 0, 0, 0};
        if (gx < cols) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                a[k] = *reinterpret_cast<const uint32_t*>(src + (size_t)reflect101(4 * yq + k - RM, rows) * pitch + gx);
        }
        const uint32_t p01l = __builtin_amdgcn_perm(a[1], a[0], 0x05010400), p01h = __builtin_amdgcn_perm(a[1], a[0], 0x07030602);
        const uint32_t p23l = __builtin_amdgcn_perm(a[3], a[2], 0x05010400), p23h = __builtin_amdgcn_perm(a[3], a[2], 0x07030602);
        uint4 t;
        t.x = __builtin_amdgcn_perm(p23l, p01l, 0x05040100);
        t.y = __builtin_amdgcn_perm(p23l, p01l, 0x07060302);
        t.z = __builtin_amdgcn_perm(p23h, p01h, 0x05040100);
        t.w = __builtin_amdgcn_perm(p23h, p01h, 0x07060302);
        *reinterpret_cast<uint4*>(rp + yq * STRIP_W + 4 * xq) = t;
    }
    __syncthreads();

    // gridDim.y splits the six levels over workgroups (each re-stages the strip): a single frame
    // has only cols/64 strips, far too few workgroups for 256 CUs
    uint32_t hbias = STRIP_HBIAS;
    asm volatile("" : "+v"(hbias));  // one VGPR for the whole kernel (not an inline constant; the taps occupy the scalar operand)
    for (int l = blockIdx.y; l < VSLAM_NUM_LEVELS; l += gridDim.y) {
        const int n = taps->n[l], r = n >> 1;
        const int M = (n + 3 + 3) >> 2;            // dwords covering bytes [0, 3 + n)
        const int phi = ((r - RM) % 4 + 4) % 4;    // item rows start where the window is dword aligned
        const int ty_start = phi ? phi - 4 : 0;
        const int nq = (rows - ty_start + 3) >> 2;
        const uint4* tab = reinterpret_cast<const uint4*>(&taps->v4[l][0][0]);
        uint16_t* hl = h + blockIdx.z * hframe + (size_t)l * P;
        // the partial last round of items rotates over the waves with the level (as in k_pyr_octave: it
        // would otherwise always load the SIMDs that hold waves 0 and 1)
        const int n_items = nq * (STRIP_W / 4), full = n_items & ~255;
        for (int base_it = 0; base_it < n_items; base_it += 256) {
            int it = base_it + tid;
            if (base_it == full) {
                it = full + ((tid - 64 * (l & 3)) & 255);
                if (it >= n_items) break;
            }
            const int cg = it & 15, q = it >> 4;
            const int ty0 = ty_start + 4 * q;
            uint32_t acc[4][4];
            const uint4* col = reinterpret_cast<const uint4*>(rp + ((ty0 - r + RM) >> 2) * STRIP_W + 4 * cg);
            {   // first tap group: starts the accumulators at STRIP_HBIAS (see k_gauss_h_strip), no zeroing pass
                const uint4 v = col[0];
                const uint4 t = tab[0];
                const uint32_t tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][0] = udot4(v.x, tt[j], hbias);
                    acc[j][1] = udot4(v.y, tt[j], hbias);
                    acc[j][2] = udot4(v.z, tt[j], hbias);
                    acc[j][3] = udot4(v.w, tt[j], hbias);
                }
            }
#pragma unroll 2
            for (int m = 1; m < M; ++m) {
                const uint4 v = col[m * (STRIP_W / 4)];
                const uint4 t = tab[m];  // wave-uniform: scalar load
                const uint32_t tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j][0] = udot4(v.x, tt[j], acc[j][0]);
                    acc[j][1] = udot4(v.y, tt[j], acc[j][1]);
                    acc[j][2] = udot4(v.z, tt[j], acc[j][2]);
                    acc[j][3] = udot4(v.w, tt[j], acc[j][3]);
                }
            }
            const int gx = x0 + 4 * cg;
            if (gx < cols) {
                const int off0 = ty0 * pitch + gx;  // one multiply per item, rows by addition (fewer instructions); a level's scratch is far below 2^31 elements
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = ty0 + j;
                    if (y >= 0 && y < rows)
                        *reinterpret_cast<uint2*>(hl + (off0 + j * pitch)) =
                            make_uint2(__builtin_amdgcn_perm(acc[j][1], acc[j][0], 0x05040100), __builtin_amdgcn_perm(acc[j][3], acc[j][2], 0x05040100));
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------- horizontal pass
// One level of one item (8 columns x RI rows): DL = r & 1 (window offset inside the first pair).
template <int DL, int RI>
__device__ __forceinline__ void h_item_level(const uint32_t* __restrict__ hrow, int pw, int tx0, int nb,
                                             const uint32_t* __restrict__ tp, uint32_t (&acc)[RI][8]) {
    // one block of 8 input columns (4 pairs) against the 8 outputs; FIRST: the accumulators start here
    // (the rounding constant is in the data: STRIP_HBIAS), so there is no initialisation pass
    auto block = [&](int b, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        // tap pairs e in [8b-8, 8b+8): 16 wave-uniform dwords
        uint32_t T[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) T[i] = tp[8 * b + i];
#pragma unroll
        for (int jr = 0; jr < RI; ++jr) {
            const uint4 v = *reinterpret_cast<const uint4*>(hrow + jr * pw + (tx0 >> 1) + 4 * b);
            asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));  // stay one ds_read_b128
            const uint32_t vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // e = 2(4b+pp) - j - DL + 1 ; table index = e + 8 - 8b
                    const int ti = 2 * pp - j - DL + 1 + 8;
                    acc[jr][j] = udot2(vv[pp], T[ti], (FIRST && pp == 0) ? 0u : acc[jr][j]);
                }
        }
    };
    block(0, std::true_type{});
    for (int b = 1; b < nb; ++b) block(b, std::false_type{});
}

// grid = (1, ceil(rows/SH), frames); dynamic LDS = SH * pw * 4 bytes, pw = (cols + 2*(rmax+1) + 8)/2
// rounded up to a multiple of 4.  Any width with ceil(cols/8)*(SH/RI) <= 512; `pitch` / `npitch` (multiples
// of 16) are the row pitches of the scratch (elements), of the output planes and of the next base.
// RI = rows per item: 4 amortises the tap loads (batches); 1 spreads a small launch over all
// 256 threads (a single frame is bound by the latency of one workgroup, not by throughput).
template <int SH, int RI>
__global__ __launch_bounds__(256) void k_gauss_h_strip(const uint16_t* __restrict__ h, size_t hframe,
                                                        uint8_t* __restrict__ oct_out, size_t pframe, int rows,
                                                        int cols, int pitch, int pw, const StripTaps* __restrict__ taps,
                                                        uint8_t* __restrict__ next_base, size_t nframe, int nrows,
                                                        int ncols, int npitch) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* hp = smem;  // [SH][pw] u16 pairs; LDS column cx = x + PL
    const int tid = threadIdx.x;
    const int y0 = blockIdx.y * SH;
    const size_t P = (size_t)rows * pitch;
    uint8_t* out = oct_out + blockIdx.z * pframe;
    const int ncg = (cols + 7) >> 3, items = ncg * (SH / RI);
    // items per thread: two of 8 columns x 4 rows, or four of fewer rows (the row count per item is chosen on the host so that the
    // items fill whole waves: 960 columns x 16 rows are 480 items of 4 rows = 7.5 waves, but 960 items of 2 rows = 15)
    constexpr int NI = RI == 4 ? 2 : 4;
    uint32_t prev_e[NI][RI][2], prev_o[NI][RI][2];
    // this thread's (at most two) items and the plane offset of each item's first row: the same for every
    // level, so the division and the multiply are done once (a dozen instructions saved per level)
    int item_cg[NI], item_rg[NI];
    uint32_t item_off[NI];
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) {
        const int it = tid + ii * 256;
        item_cg[ii] = it % ncg, item_rg[ii] = it / ncg;
        item_off[ii] = (uint32_t)(y0 + RI * item_rg[ii]) * (uint32_t)pitch + (uint32_t)(8 * item_cg[ii]);
    }

    for (int l = 0; l < VSLAM_NUM_LEVELS; ++l) {
        const int n = taps->n[l], r = n >> 1, dl = r & 1, PL = r + dl;
        const uint16_t* hl = h + blockIdx.z * hframe + (size_t)l * P;
        // ---- stage SH rows, reflect-101 extended, as u16 pairs -------------------------------
        __syncthreads();  // previous level's reads are done
        for (int jr = __builtin_amdgcn_readfirstlane(tid >> 6); jr < SH; jr += 4) {  // one wave per row: the row address is scalar
            const uint16_t* row = hl + (size_t)min(y0 + jr, rows - 1) * pitch;
            uint32_t* dstp = hp + jr * pw + (PL >> 1);  // pair index of image column 0
            // interior: 16-byte coalesced loads (8 columns), LDS side is only 4-byte aligned.  The
            // last group of a width that is not a multiple of 8 brings scratch padding along; the
            // halo loop below (same wave, later in program order) overwrites every pair with a
            // column >= cols that a valid output can reach.
            for (int x8 = tid & 63; x8 < ncg; x8 += 64) {
                const uint4 v = *reinterpret_cast<const uint4*>(row + 8 * x8);
                uint32_t* q = dstp + 4 * x8;
                q[0] = v.x, q[1] = v.y, q[2] = v.z, q[3] = v.w;
            }
            // reflect-101 halos: PL/2 pairs on the left, PL/2 + 1 on the right
            const int nh = (PL >> 1) + 1;
            for (int i = tid & 63; i < 2 * nh; i += 64) {
                const int pi = i < nh ? i - nh + (PL >> 1) : (cols >> 1) + (PL >> 1) + (i - nh);  // left: 0..PL/2-1 (i=0 unused), right
                if (pi < 0) continue;
                const int x = 2 * pi - PL;
                if (x >= 0 && x + 1 < cols) continue;
                hp[jr * pw + pi] = (uint32_t)row[reflect101(x, cols)] | ((uint32_t)row[reflect101(x + 1, cols)] << 16);
            }
        }
        __syncthreads();
        const int nb = (((7 + dl + 2 * r) >> 1) >> 2) + 1;
        const uint32_t* tp = &taps->hp[l][0];
#pragma unroll
        for (int ii = 0; ii < NI; ++ii) {
            const int it = tid + ii * 256;
            if (it < items) {
                const int cg = item_cg[ii], rg = item_rg[ii];
                uint32_t acc[RI][8];
                if (dl)
                    h_item_level<1, RI>(hp + (RI * rg) * pw, pw, 8 * cg, nb, tp, acc);
                else
                    h_item_level<0, RI>(hp + (RI * rg) * pw, pw, 8 * cg, nb, tp, acc);
                const int x = 8 * cg;
#pragma unroll
                for (int jr = 0; jr < RI; ++jr) {
                    const int y = y0 + RI * rg + jr;
                    uint32_t g[2], d[2] = {0, 0};
#pragma unroll
                    for (int hw = 0; hw < 2; ++hw) {
                        const uint32_t e = __builtin_amdgcn_perm(acc[jr][4 * hw + 2], acc[jr][4 * hw + 0], 0x0c060c02);
                        const uint32_t o = __builtin_amdgcn_perm(acc[jr][4 * hw + 3], acc[jr][4 * hw + 1], 0x0c060c02);
                        g[hw] = __builtin_amdgcn_perm(o, e, 0x06020400);  // bytes (e0, o0, e1, o1): interleave in one v_perm
                        if (l > 0) d[hw] = __builtin_amdgcn_perm(pk_sub_sat_u16(o, prev_o[ii][jr][hw]), pk_sub_sat_u16(e, prev_e[ii][jr][hw]), 0x06020400);
                        prev_e[ii][jr][hw] = e;
                        prev_o[ii][jr][hw] = o;
                    }
                    if (y < rows) {
                        const uint32_t off = item_off[ii] + (uint32_t)(jr * pitch);  // 32-bit offset in a uniform plane pointer
                        uint8_t* gp = out + (size_t)l * P;
                        *reinterpret_cast<uint2*>(gp + off) = make_uint2(g[0], g[1]);
                        if (l > 0) {
                            uint8_t* dp = out + (size_t)(VSLAM_NUM_LEVELS + l - 1) * P;
                            *reinterpret_cast<uint2*>(dp + off) = make_uint2(d[0], d[1]);
                        }
                        // next octave's base = Gaussian[3] decimated 2:1 (GaussPyramid.cpp:123-126)
                        if (l == 3 && next_base && (y & 1) == 0 && (y >> 1) < nrows && (x >> 1) < ncols)
                            *reinterpret_cast<uint32_t*>(next_base + blockIdx.z * nframe + (size_t)(y >> 1) * npitch + (x >> 1)) =
                                __builtin_amdgcn_perm(g[1], g[0], 0x06040200);
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------ fused band kernel
// Round 3: both passes of a coarse octave in ONE kernel, the 16-bit row sums never leave the CU.
//   One 512-thread workgroup = a band of SH output rows x ALL columns of one frame's octave.
//   LDS: rp [(SH + 2 RM) / 4][colsP] dwords - the band's rows plus the vertical halo RM of the octave base,
//        byte-transposed (4 vertically adjacent pixels per dword), reflect-101 resolved at fill time
//        (the whole 480 x 270 base of octave 3 is 127 KB; a band of octave 2 is 8 + 120 rows x 960);
//        hp [SH][pw] dwords - the current level's row sums of the band as u16 pairs with the
//        reflect-101 extension, exactly the image k_gauss_h_strip stages from the scratch.
//   Per level: vertical dot4 items (4 columns x 4 rows, window alignment by the per-level offset
//   delta as in k_pyr_octave) -> hp; reflect halos copied inside LDS; horizontal dot2 items
//   (8 columns x RI rows, h_item_level above) -> pack, saturating DoG, 8-byte stores.
// Against the two strip kernels this removes the [frame][6][P] u16 scratch (12 P bytes written and read
// again per frame: 15.5 MB of a 1080p frame's 24 MB strip traffic), the second launch and the
// horizontal kernel's staging loads.  The price is occupancy: one workgroup per CU (2 waves per SIMD).
// grid = (1, ceil(rows / SH), frames); block = 512; dynamic LDS = ((wq + 1) * colsP + SH * pw) * 4 bytes, wq = (SH + 2 RM) / 4.
template <int DELTA>
__device__ __forceinline__ void band_vertical_item(const uint4* __restrict__ col, int colsP4, const uint4* __restrict__ tab, int M,
                                                   uint32_t hbias, uint32_t (&acc)[4][4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[j][c] = hbias;
    uint4 tprev = make_uint4(0, 0, 0, 0);
#pragma unroll 2
    for (int m = 0; m < M; ++m) {
        const uint4 v = col[m * colsP4];
        const uint4 tcur = tab[m];  // wave-uniform: scalar load
        const uint32_t tc[4] = {tcur.x, tcur.y, tcur.z, tcur.w}, tp[4] = {tprev.x, tprev.y, tprev.z, tprev.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // row j of the item starts DELTA + j bytes into the first dword of the column window
            const uint32_t t = DELTA + j < 4 ? tc[DELTA + j] : tp[DELTA + j - 4];
            acc[j][0] = udot4(v.x, t, acc[j][0]);
            acc[j][1] = udot4(v.y, t, acc[j][1]);
            acc[j][2] = udot4(v.z, t, acc[j][2]);
            acc[j][3] = udot4(v.w, t, acc[j][3]);
        }
        tprev = tcur;
    }
}

template <int SH, int RI>
__global__ __launch_bounds__(512) void k_gauss_band(const uint8_t* __restrict__ base, size_t bframe, uint8_t* __restrict__ oct_out,
                                                     size_t pframe, int rows, int cols, int pitch, int RM, int colsP, int pw,
                                                     const StripTaps* __restrict__ taps, uint8_t* __restrict__ next_base,
                                                     size_t nframe, int nrows, int ncols, int npitch) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    const int wq = (SH + 2 * RM) >> 2;
    uint32_t* rp = smem;                     // [wq + 1][colsP]: one spare row quad, the last tap dword of a window may start in it (zero taps)
    uint32_t* hp = smem + (wq + 1) * colsP;  // [SH][pw]
    const int tid = threadIdx.x;
    const int y0 = blockIdx.y * SH;
    const uint8_t* src = base + blockIdx.z * bframe;
    uint8_t* out = oct_out + blockIdx.z * pframe;
    const size_t P = (size_t)rows * pitch;
    const int ncq = (cols + 3) >> 2;  // 4-column groups

    // ---- stage rows [y0 - RM, y0 + SH + RM) of the base, byte-transposed ------------------------------
    for (int it = tid; it < wq * ncq; it += 512) {
        const int yq = it / ncq, xq = it - yq * ncq;
        uint32_t a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)  // the dword may run into the row padding (pitch is a multiple of 16): those columns feed nothing
            a[k] = *reinterpret_cast<const uint32_t*>(src + (size_t)reflect101(y0 - RM + 4 * yq + k, rows) * pitch + 4 * xq);
        const uint32_t p01l = __builtin_amdgcn_perm(a[1], a[0], 0x05010400), p01h = __builtin_amdgcn_perm(a[1], a[0], 0x07030602);
        const uint32_t p23l = __builtin_amdgcn_perm(a[3], a[2], 0x05010400), p23h = __builtin_amdgcn_perm(a[3], a[2], 0x07030602);
        uint4 t;
        t.x = __builtin_amdgcn_perm(p23l, p01l, 0x05040100);
        t.y = __builtin_amdgcn_perm(p23l, p01l, 0x07060302);
        t.z = __builtin_amdgcn_perm(p23h, p01h, 0x05040100);
        t.w = __builtin_amdgcn_perm(p23h, p01h, 0x07060302);
        *reinterpret_cast<uint4*>(rp + yq * colsP + 4 * xq) = t;
    }

    uint32_t hbias = STRIP_HBIAS;
    asm volatile("" : "+v"(hbias));
    const int ncg = (cols + 7) >> 3, items = ncg * (SH / RI);
    uint32_t prev_e[2][RI][2], prev_o[2][RI][2];
    int item_cg[2], item_rg[2];
    uint32_t item_off[2];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
        const int it = tid + ii * 512;
        item_cg[ii] = it % ncg, item_rg[ii] = it / ncg;
        item_off[ii] = (uint32_t)(y0 + RI * item_rg[ii]) * (uint32_t)pitch + (uint32_t)(8 * item_cg[ii]);
    }
    const uint16_t* hp16 = reinterpret_cast<const uint16_t*>(hp);

    for (int l = 0; l < VSLAM_NUM_LEVELS; ++l) {
        const int n = taps->n[l], r = n >> 1, dl = r & 1, PL = r + dl;
        __syncthreads();  // staging done (l = 0) / the previous level's horizontal reads of hp are done
        // ---- vertical pass: band rows 4q .. 4q+3, columns 4cg .. 4cg+3 -> hp -------------------------
        {
            const int A = (RM - r) >> 2, delta = (RM - r) & 3;  // the window of band row 0 starts delta bytes into dword A of its column
            const int M = ((delta + 3 + n - 1) >> 2) + 1;
            const uint4* tab = reinterpret_cast<const uint4*>(&taps->v4[l][0][0]);
            for (int it = tid; it < ncq * (SH / 4); it += 512) {
                const int cg = it % ncq, q = it / ncq;
                const uint4* col = reinterpret_cast<const uint4*>(rp + (q + A) * colsP + 4 * cg);
                uint32_t acc[4][4];
                switch (delta) {  // wave-uniform
                    case 0: band_vertical_item<0>(col, colsP >> 2, tab, M, hbias, acc); break;
                    case 1: band_vertical_item<1>(col, colsP >> 2, tab, M, hbias, acc); break;
                    case 2: band_vertical_item<2>(col, colsP >> 2, tab, M, hbias, acc); break;
                    default: band_vertical_item<3>(col, colsP >> 2, tab, M, hbias, acc); break;
                }
                uint32_t* d = hp + (4 * q) * pw + (PL >> 1) + 2 * cg;  // pair index of image column 4cg (PL is even)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    d[j * pw] = __builtin_amdgcn_perm(acc[j][1], acc[j][0], 0x05040100);
                    d[j * pw + 1] = __builtin_amdgcn_perm(acc[j][3], acc[j][2], 0x05040100);
                }
            }
        }
        __syncthreads();
        // ---- reflect-101 halos of the band's rows, copied inside LDS: PL/2 pairs left, PL/2 + 1 right -
        {
            const int nh = (PL >> 1) + 1;
            for (int it = tid; it < SH * 2 * nh; it += 512) {
                const int jr = it / (2 * nh), i = it - jr * (2 * nh);
                const int pi = i < nh ? i - nh + (PL >> 1) : (cols >> 1) + (PL >> 1) + (i - nh);
                if (pi < 0) continue;
                const int x = 2 * pi - PL;
                if (x >= 0 && x + 1 < cols) continue;
                const uint16_t* row = hp16 + jr * 2 * pw + PL;  // element of image column 0
                hp[jr * pw + pi] = (uint32_t)row[reflect101(x, cols)] | ((uint32_t)row[reflect101(x + 1, cols)] << 16);
            }
        }
        __syncthreads();
        // ---- horizontal pass + epilogue (as k_gauss_h_strip) ------------------------------------------
        const int nb = (((7 + dl + 2 * r) >> 1) >> 2) + 1;
        const uint32_t* tp = &taps->hp[l][0];
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int it = tid + ii * 512;
            if (it < items) {
                const int cg = item_cg[ii], rg = item_rg[ii];
                uint32_t acc[RI][8];
                if (dl)
                    h_item_level<1, RI>(hp + (RI * rg) * pw, pw, 8 * cg, nb, tp, acc);
                else
                    h_item_level<0, RI>(hp + (RI * rg) * pw, pw, 8 * cg, nb, tp, acc);
                const int x = 8 * cg;
#pragma unroll
                for (int jr = 0; jr < RI; ++jr) {
                    const int y = y0 + RI * rg + jr;
                    uint32_t g[2], d[2] = {0, 0};
#pragma unroll
                    for (int hw = 0; hw < 2; ++hw) {
                        const uint32_t e = __builtin_amdgcn_perm(acc[jr][4 * hw + 2], acc[jr][4 * hw + 0], 0x0c060c02);
                        const uint32_t o = __builtin_amdgcn_perm(acc[jr][4 * hw + 3], acc[jr][4 * hw + 1], 0x0c060c02);
                        g[hw] = __builtin_amdgcn_perm(o, e, 0x06020400);
                        if (l > 0) d[hw] = __builtin_amdgcn_perm(pk_sub_sat_u16(o, prev_o[ii][jr][hw]), pk_sub_sat_u16(e, prev_e[ii][jr][hw]), 0x06020400);
                        prev_e[ii][jr][hw] = e;
                        prev_o[ii][jr][hw] = o;
                    }
                    if (y < rows) {
                        const uint32_t off = item_off[ii] + (uint32_t)(jr * pitch);
                        uint8_t* gp = out + (size_t)l * P;
                        *reinterpret_cast<uint2*>(gp + off) = make_uint2(g[0], g[1]);
                        if (l > 0) {
                            uint8_t* dp = out + (size_t)(VSLAM_NUM_LEVELS + l - 1) * P;
                            *reinterpret_cast<uint2*>(dp + off) = make_uint2(d[0], d[1]);
                        }
                        if (l == 3 && next_base && (y & 1) == 0 && (y >> 1) < nrows && (x >> 1) < ncols)
                            *reinterpret_cast<uint32_t*>(next_base + blockIdx.z * nframe + (size_t)(y >> 1) * npitch + (x >> 1)) =
                                __builtin_amdgcn_perm(g[1], g[0], 0x06040200);
                    }
                }
            }
        }
    }
}

// Host side: build the tables for one octave.
static bool strip_pack_taps(const uint16_t* const t[6], const int n[6], StripTaps& out) {
    memset(&out, 0, sizeof(out));
    for (int l = 0; l < VSLAM_NUM_LEVELS; ++l) {
        if (n[l] > STRIP_MAXN || (n[l] & 1) == 0) return false;
        out.n[l] = n[l];
        unsigned sum = 0;
        for (int k = 0; k < n[l]; ++k) {
            if (t[l][k] > 255) return false;
            sum += t[l][k];
        }
        if (sum != 256) return false;  // STRIP_HBIAS relies on it
        for (int m = 0; m < STRIP_MAXM; ++m)
            for (int j = 0; j < 4; ++j) {
                uint32_t w = 0;
                for (int b = 0; b < 4; ++b) {
                    const int k = 4 * m + b - j;
                    if (k >= 0 && k < n[l]) w |= (uint32_t)t[l][k] << (8 * b);
                }
                out.v4[l][m][j] = w;
            }
        for (int e = 0; e <= n[l]; ++e) {
            const uint32_t lo = e >= 1 ? t[l][e - 1] : 0, hi = e < n[l] ? t[l][e] : 0;
            out.hp[l][8 + e] = lo | (hi << 16);
        }
    }
    return true;
}

}  // namespace vslam
