// K-D1, matrix-core form on 16 x 16 x 64 tiles (OPT-IN with the rest of the matrix path: VSLAM_MX=1 / vslam_ctx_set_matrix_path).
//
// Same contract, same arithmetic and the same staging / fused-scan code as k_pyr_octave_mx (kernels_pyramid_mx.hip.h); what
// changes is the MFMA shape.  The 32 x 32 x 32 form keeps 48 accumulator registers per lane (C1, C2hi, C2lo: 16 each) and 32
// for the previous level's G of a 32-row strip: 229-250 vector registers, two waves per SIMD, 500 of the SIMD's 512 registers
// taken - nothing co-resides with it, not a third wave of its own and not a wave of the Harris chain or the scans, and the
// kernel sits at 40 % of its cycles in s_waitcnt with every unit half idle.  v_mfma_i32_16x16x64_i8
// produces 16 x 16 outputs over the SAME 64-deep K window with FOUR accumulator registers:
//
//   pass 1 (vertical)    C1[x, y'] = sum_y P'[x, y] Tv[y, y']   A: lane = (column x = l & 15, K group g = l >> 4) holds rows
//                        -off + 64 s + 16 g + (0..15) of its column: four dwords of the byte-transposed LDS image;
//                        B: the taps as a band, lane = (output row y', g).  C1: lane (y', gm) holds columns 4 gm + v, v = 0..3.
//   hand-off             the four values of a lane -> one dword of the high-byte plane, one of the low-byte plane (3 v_perm +
//                        1 v_xor): exactly one K dword of pass 2's B operand, for THIS lane - no lane exchange, the K index
//                        (g, 4 q + v) of pass 2 is DEFINED as column 16 q + 4 g + v of the 64-column window and the taps of
//                        its A operand are packed in that order (mx16_pack_taps).
//   pass 2 (horizontal)  C2[x', y'] = sum_x Th[x', x] H[x, y']   B = the plane dwords of input blocks ob .. ob + 3 (+ 4 s),
//                        A = the taps; lane (y', gm) ends with columns 16 ob + 4 gm + v of row y': one dword of G.
//   epilogue / flush     as the 32-wide form: byte 2 of (C2hi << 8) + C2lo, saturating DoG against the previous level, a dword
//                        per output block into the wave's LDS buffer (16 rows x 128 bytes), 8 rows x 128 contiguous bytes per
//                        store instruction out of it.
//
// A wave owns a strip of 16 rows x 128 columns (8 output blocks, 10-11 input blocks); a 128 x 128 tile is 8 waves, two tiles
// per CU = four waves per SIMD at <= 128 registers.  Per pixel the MFMA cycles, LDS bytes and vector instructions are those of
// the 32-wide form (twice the LDS read instructions); the point is occupancy.
#pragma once
#include "kernels_pyramid_mx.hip.h"

// Timing knock-outs (tools/mx16_knockout.sh builds libraries with -DVSLAM_MX16_KO=<mask> under lib/ab/: wrong results by
// construction, never the product): 1 = the planes' HBM stores become register sinks, 2 = no tile staging.
#ifndef VSLAM_MX16_KO
#define VSLAM_MX16_KO 0
#endif

namespace vslam {

template <int TH_, int N0, int N1, int N2, int N3, int N4, int N5>
struct Mx16Cfg {
    static constexpr int TW = 128, TH = TH_, SW = 128;
    static constexpr int NW = TH / 16, NT = 64 * NW;
    static constexpr int NOB = SW / 16;  // output blocks of a strip
    static constexpr int DBUF = 1;
    static constexpr int SITE_PAIRS = 3;  // the fused scan: 5-6 lattice rows per 16-row strip, two per register
    static constexpr int n(int l) { return l == 0 ? N0 : l == 1 ? N1 : l == 2 ? N2 : l == 3 ? N3 : l == 4 ? N4 : N5; }
    static constexpr int r(int l) { return n(l) / 2; }
    static constexpr int off(int l) { return (r(l) + 15) / 16 * 16; }              // the K window starts `off` before the output block
    static constexpr int ns(int l) { return (off(l) + 16 + r(l) + 63) / 64; }      // K steps of 64: [-off, -off + 64 ns) covers [-r, 16 + r)
    static constexpr int needb(int l) { return (off(l) + SW + r(l) + 15) / 16; }   // input blocks of a strip that meet a non-zero tap
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int R = cmax(cmax(cmax(off(0), off(1)), cmax(off(2), off(3))), cmax(off(4), off(5)));  // staged halo
    static constexpr int NSMAX = cmax(cmax(cmax(ns(0), ns(1)), cmax(ns(2), ns(3))), cmax(ns(4), ns(5)));
    static constexpr int RMAX = cmax(cmax(cmax(r(0), r(1)), cmax(r(2), r(3))), cmax(r(4), r(5)));
    // every level's window is one K step and no wider than three input blocks (r <= 16): two neighbouring output blocks then read
    // the SAME four input blocks (mx16_level_paired: the B operands stay in place, no sliding-window register copies)
    static constexpr bool PAIRED = NSMAX == 1 && RMAX <= 16 && R == 16;
    static constexpr int RQ = (TH + 2 * R) / 4, RW = TW + 2 * R, RWP = RW + 4;
    static constexpr int OBP = SW / 4 + 4;  // dwords per buffered row (pitch = 4 mod 32: the dword writes of 16 rows and the 16-byte reads along a row spread over the banks)
    static constexpr int OBUF = 16 * OBP;   // dwords per plane of a wave's output buffer
    static constexpr int STAGE_DWORDS = RQ * RWP;
    // (+ 8 rows of slack behind the last wave's buffers: the fused scan reads a few rows past a strip's D buffer for lattice rows it
    // does not own; pass 1 reads up to 64 ns - 16 - r rows past the staged tile for K rows whose taps are zero: both land in here)
    // PAIRED configurations keep the operand fragments of levels 1..5 in LDS (3 fragments x 64 lanes x 16 bytes per level): a
    // fragment fetched from memory inside the level loop shares vmcnt with the planes' stores, and the wait in front of its first
    // use then waits for those stores too - the waves spent most of a tile's time on exactly that (mx16_level_paired)
    static constexpr int TAPL_DWORDS = (NSMAX == 1 && cmax(cmax(cmax(r(0), r(1)), cmax(r(2), r(3))), cmax(r(4), r(5))) <= 16 && R == 16) ? 5 * 3 * 64 * 4 : 0;
    static constexpr int TAPL_OFF = STAGE_DWORDS + NW * 2 * OBUF + 8 * OBP;  // dword offset of that table
    static constexpr int LDS_BYTES = (TAPL_OFF + TAPL_DWORDS) * 4;
    // the last row pass 1 reads (a zero-tap row of the last strip) lies inside the allocation
    static constexpr int last_quad(int l) { return (R - off(l) + (TH - 16) + 64 * ns(l) - 1) / 4; }
    static constexpr int LASTQ = cmax(cmax(cmax(last_quad(0), last_quad(1)), cmax(last_quad(2), last_quad(3))), cmax(last_quad(4), last_quad(5)));
    static_assert((LASTQ + 1) * RWP <= LDS_BYTES / 4, "pass 1's zero-tap rows must stay inside the workgroup's LDS");
    static_assert(LDS_BYTES <= 160 * 1024 && TH % 16 == 0 && NT <= 1024, "one workgroup");
    static_assert(r(0) >= 1 && (N0 & 1) && (N1 & 1) && (N2 & 1) && (N3 & 1) && (N4 & 1) && (N5 & 1), "odd kernels");
};

// Operand fragments in lane order, one 16-byte fragment per K step (mx16_pack_taps).
template <class CFG>
struct Mx16Taps {
    mx_v4i b1[6][CFG::NSMAX][64];
    mx_v4i a2[6][CFG::NSMAX][64];
    mx_v4i a2o[6][64];  // PAIRED configurations: pass 2's taps for the ODD output block of a pair (its window starts one block earlier)
};

template <class CFG>
struct Mx16Lane {
    const uint32_t* lp;    // LDS: row quad (strip row 0 - R) + 4 g, column (strip column 0 - R) + x of the staged tile
    uint8_t* out;          // the frame's octave block
    size_t P;              // bytes per plane
    int nob_live;          // output blocks of the strip that start inside the image (wave-uniform)
    uint32_t* wb;          // LDS: this lane's dword of output block 0 in the wave's G buffer: row y', byte 4 gm
    const uint32_t* rb;    // LDS: this lane's read position: row lane >> 3, bytes 16 (lane & 7)
    uint32_t off;          // byte offset inside a plane of (strip row lane >> 3, strip column 16 (lane & 7))
    uint32_t pitch8;       // 8 * pitch
    int rows_left;         // rows - that row
    bool col_ok;           // that column is inside the image
    uint8_t* nb;           // next octave's base: the 8 bytes this lane's first flush row contributes (nullptr: none - odd row, outside)
    uint32_t npitch4;      // 4 * npitch: the second flush row is 8 image rows = 4 base rows further down
    int nrows_left;        // nrows - (that base row)
};

// The strip's G rows, then its D rows, out of the wave's LDS buffer with the lanes along the rows: 8 rows x 128 contiguous bytes
// per store instruction (one wave's LDS operations execute in order: no barrier between the writes of the level and these reads).
template <class CFG, int L, bool FULL>
__device__ __forceinline__ void mx16_flush(const Mx16Lane<CFG>& ln) {
    uint8_t* gp = ln.out + (size_t)L * ln.P;
    uint8_t* dp = ln.out + (size_t)(VSLAM_NUM_LEVELS + L - 1) * ln.P;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const uint4 gv = *reinterpret_cast<const uint4*>(ln.rb + 8 * i * CFG::OBP);
        uint4 dv;
        if (L > 0) dv = *reinterpret_cast<const uint4*>(ln.rb + CFG::OBUF + 8 * i * CFG::OBP);
#if VSLAM_MX16_KO & 1
        asm volatile("" ::"v"(gv.x), "v"(gv.y), "v"(gv.z), "v"(gv.w));
        if (L > 0) asm volatile("" ::"v"(dv.x), "v"(dv.y), "v"(dv.z), "v"(dv.w));
        (void)gp, (void)dp;
#else
        if (FULL || (ln.col_ok && 8 * i < ln.rows_left)) {
#if VSLAM_MX16_KO & 4  // A/B: streaming (non-temporal) stores of the planes
            typedef uint32_t mx_u4v __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(mx_u4v{gv.x, gv.y, gv.z, gv.w}, reinterpret_cast<mx_u4v*>(gp + ln.off + i * ln.pitch8));
            if (L > 0) __builtin_nontemporal_store(mx_u4v{dv.x, dv.y, dv.z, dv.w}, reinterpret_cast<mx_u4v*>(dp + ln.off + i * ln.pitch8));
#else
            *reinterpret_cast<uint4*>(gp + ln.off + i * ln.pitch8) = gv;
            if (L > 0) *reinterpret_cast<uint4*>(dp + ln.off + i * ln.pitch8) = dv;
#endif
        }
#endif
        // next octave's base = Gaussian[3] decimated 2:1, INTER_NEAREST (GaussPyramid.cpp:123-126): the even bytes of the even rows
        if (L == 3 && ln.nb && 4 * i < ln.nrows_left)
            *reinterpret_cast<uint2*>(ln.nb + i * ln.npitch4) = make_uint2(__builtin_amdgcn_perm(gv.y, gv.x, 0x06040200), __builtin_amdgcn_perm(gv.w, gv.z, 0x06040200));
    }
}

// One Gaussian level of a wave's 16-row strip (see mx_level for the FULL / tcur / tnext conventions).
template <class CFG, int L, bool EXT, bool FULL>
__device__ __forceinline__ void mx16_level(const Mx16Taps<CFG>* __restrict__ taps, const Mx16Lane<CFG>& ln, uint32_t (&pe)[CFG::NOB], uint32_t (&po)[CFG::NOB],
                                           MxSitesT<CFG::SITE_PAIRS>& st, const mx_v4i (&tcur)[2 * CFG::NSMAX], mx_v4i (&tnext)[2 * CFG::NSMAX]) {
    constexpr int OFF = CFG::off(L), NS = CFG::ns(L), NOB = CFG::NOB, NIN = NOB + 4 * NS - 1, NEEDB = CFG::needb(L), R = CFG::R, RWP = CFG::RWP;
    constexpr int RING = 4 * NS;
    constexpr int kLoInit = 256 * (128 + 32768) + 32768;  // the biases of the two byte planes (taps sum to 256) + the round-half-up of A2-iv
    const int lane = threadIdx.x & 63;
    mx_v4i b1[NS], a2[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) b1[s] = tcur[s], a2[s] = tcur[CFG::NSMAX + s];
    uint32_t hi[RING], lo[RING];  // the byte planes of the last 4 ns input blocks: slot = block % RING
    // pass 1 on input block ib: columns [-OFF + 16 ib, +16) of the strip, rows [-OFF, -OFF + 64 NS)
    auto pass1 = [&](int ib) {
        mx_v4i c1 = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            mx_v4i a;
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] = (int)ln.lp[((R - OFF + 64 * s) / 4 + k) * RWP + (R - OFF + 16 * ib)];
            c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b1[s], c1, 0, 0, 0);
        }
        return c1;
    };
    mx_v4i c1 = pass1(0);
#pragma unroll
    for (int ib = 0; ib < NIN; ++ib) {
        const int slot = ib % RING;
        if (ib < NEEDB) {  // C1 = H - 32768: signed high byte as it is, low byte - 128 (x ^ 0x80)
            const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)c1[1], (uint32_t)c1[0], 0x05010400);  // (lo0, lo1, hi0, hi1)
            const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)c1[3], (uint32_t)c1[2], 0x05010400);
            lo[slot] = __builtin_amdgcn_perm(t23, t01, 0x05040100) ^ 0x80808080u;
            hi[slot] = __builtin_amdgcn_perm(t23, t01, 0x07060302);
        } else {  // a block of the K window that meets zero taps only (right of the strip's halo): any finite value does
            lo[slot] = 0u, hi[slot] = 0u;
        }
        const int ob = ib - (RING - 1);
        const bool have_ob = ib >= RING - 1 && (FULL || ob < ln.nob_live);  // wave-uniform
        mx_v4i chi = {0, 0, 0, 0}, clo = {kLoInit, kLoInit, kLoInit, kLoInit};
        if (have_ob) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const mx_v4i bh = {(int)hi[(ob + 4 * s) % RING], (int)hi[(ob + 4 * s + 1) % RING], (int)hi[(ob + 4 * s + 2) % RING], (int)hi[(ob + 4 * s + 3) % RING]};
                const mx_v4i bl = {(int)lo[(ob + 4 * s) % RING], (int)lo[(ob + 4 * s + 1) % RING], (int)lo[(ob + 4 * s + 2) % RING], (int)lo[(ob + 4 * s + 3) % RING]};
                chi = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2[s], bh, chi, 0, 0, 0);
                clo = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2[s], bl, clo, 0, 0, 0);
            }
        }
        if (ib + 1 < NIN && ib + 1 < NEEDB) c1 = pass1(ib + 1);  // behind pass 2, in front of the epilogue: runs in the matrix pipe while the vector unit packs
        if (!have_ob) continue;
        // ---- epilogue: register v = column 16 ob + 4 gm + v of this lane's row ----------------------------------------------
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = ((uint32_t)chi[j] << 8) + (uint32_t)clo[j];
        const uint32_t e = __builtin_amdgcn_perm(w[2], w[0], 0x0c060c02);  // (G0, G2) in 16-bit lanes
        const uint32_t o = __builtin_amdgcn_perm(w[3], w[1], 0x0c060c02);  // (G1, G3)
        ln.wb[4 * ob] = __builtin_amdgcn_perm(o, e, 0x06020400);
        if (L > 0)  // D_{L-1} = saturate_u8(G_L - G_{L-1}), GaussPyramid.cpp:197
            ln.wb[CFG::OBUF + 4 * ob] = __builtin_amdgcn_perm(mx_pk_sub_sat_u16(o, po[ob]), mx_pk_sub_sat_u16(e, pe[ob]), 0x06020400);
        pe[ob] = e;
        po[ob] = o;
    }
    if constexpr (L < 5) {  // the next level's fragments, in front of this level's stores (mx_level)
#pragma unroll
        for (int s = 0; s < CFG::ns(L + 1); ++s) tnext[s] = taps->b1[L + 1][s][lane], tnext[CFG::NSMAX + s] = taps->a2[L + 1][s][lane];
    }
    if constexpr (EXT && L > 0) {  // fused lattice scan: DoG level L-1 of the strip is in the wave's D buffer now
        if (st.cdump) st.cdump[(size_t)(L - 1) * st.clevel] = *st.cl;
        if (st.nk > 0) {  // wave-uniform
            mx_sites_level<CFG, L - 1>(st);
            if constexpr (L == 5) mx_sites_store(st);
        }
    }
    mx16_flush<CFG, L, FULL>(ln);
}

// The same level for PAIRED configurations (octave 0: one K step, r <= 16).  Output blocks go in pairs (2m, 2m + 1): both lie
// inside the four input blocks 2m .. 2m + 3 (block 2m + 1's window is those of 2m shifted by one block: its taps are packed one
// block later, Mx16Taps::a2o), so the two byte-plane operands are ONE register quad each for the whole level - after a pair its
// upper half moves down (one v_mov_b64 per plane) and the next two blocks land in the upper half.  The generic form above
// rebuilds a sliding 4-register operand per output block: 8-10 v_mov each, a quarter more vector instructions than the 32-wide
// kernel.  Pass 1's LDS reads run one pair AHEAD of their MFMAs (the generic form waited for every read right where it issued it).
// tf: this level's fragments on entry ([0] = b1, [1] = a2 (even block), [2] = a2o (odd block)), the next level's on exit, read from the
// workgroup's LDS table `tapl` (levels 1..5; the kernel fills it while the tile is staged).
template <class CFG, int L, bool EXT, bool FULL>
__device__ __forceinline__ void mx16_level_paired(const Mx16Taps<CFG>* __restrict__ taps, const Mx16Lane<CFG>& ln, uint32_t (&pg)[CFG::NOB],
                                                  MxSitesT<CFG::SITE_PAIRS>& st, mx_v4i (&tf)[3], const mx_v4i* __restrict__ tapl) {
    static_assert(CFG::PAIRED && CFG::off(L) == 16 && CFG::ns(L) == 1 && CFG::needb(L) <= CFG::NOB + 2, "one K step, three-block windows");
    constexpr int NOB = CFG::NOB, NP = NOB / 2, RWP = CFG::RWP;
    constexpr int kLoInit = 256 * (128 + 32768) + 32768;
    const int lane = threadIdx.x & 63;
    const mx_v4i b1 = tf[0], a2e = tf[1], a2o = tf[2];
    const mx_v4i kinit = {kLoInit, kLoInit, kLoInit, kLoInit}, zero = {0, 0, 0, 0};
    // A operand of pass 1 for input block ib: rows [-16, 48) of the strip (R = off = 16: the staged tile's own origin), columns 16 ib ..
    auto load_a = [&](int ib) {
        mx_v4i a;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = (int)ln.lp[k * RWP + 16 * ib];
        return a;
    };
    auto split = [&](const mx_v4i& c1, uint32_t& h, uint32_t& l) {  // C1 = H - 32768: signed high bytes, low bytes - 128
        const uint32_t t01 = __builtin_amdgcn_perm((uint32_t)c1[1], (uint32_t)c1[0], 0x05010400);
        const uint32_t t23 = __builtin_amdgcn_perm((uint32_t)c1[3], (uint32_t)c1[2], 0x05010400);
        l = __builtin_amdgcn_perm(t23, t01, 0x05040100) ^ 0x80808080u;
        h = __builtin_amdgcn_perm(t23, t01, 0x07060302);
    };
#if VSLAM_MX16_KO & 8  // timing knock-out: no arithmetic, the flush stores whatever the LDS buffer holds
    if constexpr (L < 5) tf[0] = tapl[(3 * L + 0) * 64 + lane], tf[1] = tapl[(3 * L + 1) * 64 + lane], tf[2] = tapl[(3 * L + 2) * 64 + lane];
    (void)b1, (void)a2e, (void)a2o, (void)kinit, (void)zero, (void)pg, (void)st, (void)taps;
    mx16_flush<CFG, L, FULL>(ln);
    return;
#endif
    mx_v4i bh, bl;  // the byte planes of input blocks 2m .. 2m + 3
    {   // blocks 0 .. 3
        mx_v4i a0 = load_a(0), a1 = load_a(1), a2 = load_a(2), a3 = load_a(3);
        const mx_v4i c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b1, zero, 0, 0, 0), c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, zero, 0, 0, 0);
        const mx_v4i c2 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2, b1, zero, 0, 0, 0), c3 = __builtin_amdgcn_mfma_i32_16x16x64_i8(a3, b1, zero, 0, 0, 0);
        uint32_t h, l;
        split(c0, h, l), bh[0] = (int)h, bl[0] = (int)l;
        split(c1, h, l), bh[1] = (int)h, bl[1] = (int)l;
        split(c2, h, l), bh[2] = (int)h, bl[2] = (int)l;
        split(c3, h, l), bh[3] = (int)h, bl[3] = (int)l;
    }
    mx_v4i an0 = load_a(4), an1 = load_a(5);  // the next pair's two blocks, one pair ahead
#pragma unroll
    for (int m = 0; m < NP; ++m) {
        const bool have_e = FULL || 2 * m < ln.nob_live, have_o = FULL || 2 * m + 1 < ln.nob_live;  // wave-uniform
        // ---- pass 2 on blocks 2m and 2m + 1, pass 1 on the two blocks after the window, the reads of the pair after that
        const mx_v4i chi_e = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2e, bh, zero, 0, 0, 0), clo_e = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2e, bl, kinit, 0, 0, 0);
        const mx_v4i chi_o = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2o, bh, zero, 0, 0, 0), clo_o = __builtin_amdgcn_mfma_i32_16x16x64_i8(a2o, bl, kinit, 0, 0, 0);
        mx_v4i c4 = zero, c5 = zero;
        if (m + 1 < NP) {
            c4 = __builtin_amdgcn_mfma_i32_16x16x64_i8(an0, b1, zero, 0, 0, 0);
            if (2 * m + 5 < CFG::needb(L)) c5 = __builtin_amdgcn_mfma_i32_16x16x64_i8(an1, b1, zero, 0, 0, 0);
        }
        if (m + 2 < NP) {
            an0 = load_a(2 * m + 6);
            if (2 * m + 7 < CFG::needb(L)) an1 = load_a(2 * m + 7);
        }
        // ---- epilogue of the pair: register v = column 16 ob + 4 gm + v of this lane's row
        auto finish = [&](const mx_v4i& chi, const mx_v4i& clo, int ob) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = ((uint32_t)chi[j] << 8) + (uint32_t)clo[j];
            const uint32_t e = __builtin_amdgcn_perm(w[2], w[0], 0x0c060c02);  // (G0, G2) in 16-bit lanes
            const uint32_t o = __builtin_amdgcn_perm(w[3], w[1], 0x0c060c02);  // (G1, G3)
            const uint32_t gq = __builtin_amdgcn_perm(o, e, 0x06020400);
            ln.wb[4 * ob] = gq;
            if (L > 0) {  // D_{L-1} = saturate_u8(G_L - G_{L-1}), GaussPyramid.cpp:197; the previous level's G waits PACKED (a register per block, not two)
                const uint32_t pe = __builtin_amdgcn_perm(0u, pg[ob], 0x0c020c00), po = __builtin_amdgcn_perm(0u, pg[ob], 0x0c030c01);
                ln.wb[CFG::OBUF + 4 * ob] = __builtin_amdgcn_perm(mx_pk_sub_sat_u16(o, po), mx_pk_sub_sat_u16(e, pe), 0x06020400);
            }
            pg[ob] = gq;
        };
        if (have_e) finish(chi_e, clo_e, 2 * m);
        if (have_o) finish(chi_o, clo_o, 2 * m + 1);
        // ---- the window moves on by two blocks
        if (m + 1 < NP) {
            bh[0] = bh[2], bh[1] = bh[3], bl[0] = bl[2], bl[1] = bl[3];
            uint32_t h, l;
            split(c4, h, l), bh[2] = (int)h, bl[2] = (int)l;
            if (2 * m + 5 < CFG::needb(L)) split(c5, h, l), bh[3] = (int)h, bl[3] = (int)l;
            else bh[3] = 0, bl[3] = 0;  // a block that meets zero taps only
        }
    }
    if constexpr (L < 5) {  // the next level's fragments, from the workgroup's LDS table (no vmcnt traffic inside the level loop)
        tf[0] = tapl[(3 * L + 0) * 64 + lane], tf[1] = tapl[(3 * L + 1) * 64 + lane], tf[2] = tapl[(3 * L + 2) * 64 + lane];
    }
    if constexpr (EXT && L > 0) {  // fused lattice scan: DoG level L-1 of the strip is in the wave's D buffer now
        if (st.cdump) st.cdump[(size_t)(L - 1) * st.clevel] = *st.cl;
        if (st.nk > 0) {  // wave-uniform
            mx_sites_level<CFG, L - 1>(st);
            if constexpr (L == 5) mx_sites_store(st);
        }
    }
    mx16_flush<CFG, L, FULL>(ln);
    (void)taps;
}

// grid = (ceil(cols/128), ceil(rows/TH), frames); block = CFG::NT; dynamic LDS = CFG::LDS_BYTES.  Arguments as k_pyr_octave_mx.
template <class CFG, bool EXT, bool UP2>
__global__ __launch_bounds__(CFG::NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_pyr_octave_mx16(const uint8_t* __restrict__ base, size_t bframe, uint8_t* __restrict__ oct_out, size_t pframe,
                                                             int rows, int cols, int pitch, const Mx16Taps<CFG>* __restrict__ taps,
                                                             uint8_t* __restrict__ next_base, size_t nframe, int nrows, int ncols, int npitch, MxExtArgs ext,
                                                             int sstep) {
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    unsigned int bid = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned int per_xcd = (gridDim.x * gridDim.y * gridDim.z) >> 3;
    if (bid < (per_xcd << 3)) bid = (bid & 7u) * per_xcd + (bid >> 3);  // XCD-aware tile order (k_pyr_octave)
    const unsigned int tiles_per_frame = gridDim.x * gridDim.y;
    const unsigned int fz = bid / tiles_per_frame, rem = bid - fz * tiles_per_frame;
    const unsigned int by = rem / gridDim.x, bx = rem - by * gridDim.x;
    const int tile_x0 = bx * CFG::TW, tile_y0 = by * CFG::TH;

#if !(VSLAM_MX16_KO & 2)
    if constexpr (UP2) {
        // The fused upsample reads ~10 source rows per staging item through dependent, clamped 4-byte loads, and while the other
        // workgroups' plane stores fill the memory pipeline every one of them takes microseconds (staging: 0.8 ms of the kernel with
        // the stores in, 0.5 without).  Interior tiles therefore fetch the source rectangle they need ONCE, as 16-byte chunks by all
        // threads (one round trip), into the output buffers' LDS (idle until the levels start), and the upsample runs out of LDS.
        constexpr int SRC_ROWS = (CFG::TH + 2 * CFG::R) / 2 + 2, SRC_CH = ((CFG::TW + 2 * CFG::R) / 2 + 2 + 15 + 15) / 16, SRC_PITCH = 16 * SRC_CH;
        static_assert(SRC_ROWS * SRC_PITCH <= CFG::NW * 2 * CFG::OBUF * 4, "the source rectangle fits the output buffers");
        const int rows_s = rows / 2, cols_s = cols / 2;
        const uint8_t* src = base + fz * bframe;
        const int m_lo = (tile_y0 - CFG::R) / 2 - 1, c_lo = (((tile_x0 - CFG::R) / 2 - 1) & ~15);
        const bool src_interior = tile_y0 - CFG::R >= 2 && tile_x0 - CFG::R >= 2 && m_lo + SRC_ROWS <= rows_s && c_lo + SRC_PITCH <= cols_s && c_lo >= 0 &&
                                  (sstep & 15) == 0 && ((bframe | reinterpret_cast<uintptr_t>(base)) & 15) == 0;  // block-uniform
#if VSLAM_MX16_KO & 16
        if (false) {
#else
        if (src_interior) {
#endif
            uint32_t* lsrc = smem + CFG::STAGE_DWORDS;  // (the output buffers)
            for (int it = threadIdx.x; it < SRC_ROWS * SRC_CH; it += CFG::NT) {
                const int rr = it / SRC_CH, ch = it - rr * SRC_CH;
                *reinterpret_cast<uint4*>(lsrc + rr * (SRC_PITCH / 4) + 4 * ch) = *reinterpret_cast<const uint4*>(src + (size_t)(m_lo + rr) * sstep + c_lo + 16 * ch);
            }
            __syncthreads();
            // the same staging code on a "frame" whose rows m_lo .. and columns c_lo .. live in LDS (no clamp is active on an interior tile)
            const uint8_t* vsrc = reinterpret_cast<const uint8_t*>(lsrc) - ((ptrdiff_t)m_lo * SRC_PITCH + c_lo);
            mx_stage_tile_up2<CFG::TW, CFG::TH, CFG::R, CFG::RWP, CFG::NT, 2, 2>(vsrc, SRC_PITCH, rows_s, cols_s, tile_x0, tile_y0, smem, 0x80808080u);
        } else {
            mx_stage_tile_up2<CFG::TW, CFG::TH, CFG::R, CFG::RWP, CFG::NT, 2, 2>(src, sstep, rows_s, cols_s, tile_x0, tile_y0, smem, 0x80808080u);
        }
    } else
        mx_stage_tile<CFG::TW, CFG::TH, CFG::R, CFG::RWP, CFG::NT>(base + fz * bframe, rows, cols, pitch, tile_x0, tile_y0, smem, 0x80808080u);
#endif
    if constexpr (CFG::PAIRED) {  // the operand fragments of levels 1..5 into LDS: [level - 1][b1, a2, a2o][lane]
        uint4* tl = reinterpret_cast<uint4*>(smem + CFG::TAPL_OFF);
        for (int it = threadIdx.x; it < 5 * 3 * 64; it += CFG::NT) {
            const int l = 1 + it / 192, k = (it / 64) % 3, ln64 = it & 63;
            const mx_v4i* src = k == 0 ? &taps->b1[l][0][ln64] : k == 1 ? &taps->a2[l][0][ln64] : &taps->a2o[l][ln64];
            tl[it] = *reinterpret_cast<const uint4*>(src);
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 15, g = lane >> 4;
    const int Yw = 16 * wave;
    if (tile_y0 + Yw >= rows) return;  // the whole strip is below the image (no barrier follows)
    Mx16Lane<CFG> ln;
    ln.lp = smem + (Yw / 4 + 4 * g) * CFG::RWP + q;
    ln.out = oct_out + fz * pframe;
    ln.P = (size_t)rows * pitch;
    uint32_t* obuf = smem + CFG::STAGE_DWORDS + wave * 2 * CFG::OBUF;
    ln.wb = obuf + q * CFG::OBP + g;
    ln.rb = obuf + (lane >> 3) * CFG::OBP + 4 * (lane & 7);
    const int yr = tile_y0 + Yw + (lane >> 3), xr = tile_x0 + 16 * (lane & 7);
    ln.off = (uint32_t)yr * (uint32_t)pitch + (uint32_t)xr;
    ln.pitch8 = 8u * (uint32_t)pitch;
    ln.rows_left = rows - yr;
    ln.col_ok = xr < cols;
    ln.nob_live = min(CFG::NOB, (cols - tile_x0 + 15) / 16);
    ln.nb = (next_base && (yr & 1) == 0 && (yr >> 1) < nrows && (xr >> 1) < ncols && xr < cols) ? next_base + fz * nframe + (size_t)(yr >> 1) * npitch + (xr >> 1) : nullptr;
    ln.npitch4 = 4u * (uint32_t)npitch;
    ln.nrows_left = nrows - (yr >> 1);
    MxSitesT<CFG::SITE_PAIRS> st;
    if (EXT) {
        // lattice rows / columns whose 2 x 2 window lies inside this strip (as k_pyr_octave_mx, with 16-row strips)
        const int Y0 = tile_y0 + Yw, X0 = tile_x0;
        const int a_lo = (Y0 + 2) / 3, b_lo = (X0 + 2) / 3;
        const int a_first = a_lo + ((Y0 > 0 && 3 * a_lo == Y0) ? 1 : 0);  // 3a = Y0: the window's upper row belongs to the strip above
        const int b_first = b_lo + ((X0 > 0 && 3 * b_lo == X0) ? 1 : 0);
        const int a_last = min((Y0 + 15) / 3, ext.lat_rows - 1);
        st.nk = max(0, a_last - a_first + 1);
        const int b = b_first + lane;
        const bool has = 3 * b <= X0 + CFG::SW - 1 && b < ext.lat_cols;
        const int xa = has ? max(3 * b - 1 - X0, 0) : 0;
        const uint32_t sbyte = xa & 3;
        st.sel = 0x0c000c00u | sbyte | ((sbyte + ((3 * b - 1 - X0 >= 0) ? 1u : 0u)) << 16);
        const int r0 = 3 * a_first - 1 - Y0;
        const uint8_t* dbuf = reinterpret_cast<const uint8_t*>(obuf + CFG::OBUF) + (xa & ~3);
        st.slr = dbuf + r0 * (CFG::OBP * 4);
        st.sla0 = dbuf + max(r0, 0) * (CFG::OBP * 4);
        st.smap = has ? ext.sitemap + fz * ext.mframe + (size_t)a_first * ext.mpitch + b : nullptr;
        st.mpitch = (uint32_t)ext.mpitch;
        constexpr int SEAM = 3 * CFG::SW;
        const bool right_of = X0 > 0 && X0 % SEAM == 0 && X0 / SEAM <= ext.nseams;
        const bool left_of = (X0 + CFG::SW) % SEAM == 0 && (X0 + CFG::SW) / SEAM <= ext.nseams;
        st.cdump = nullptr;
        if ((right_of || left_of) && lane < 16 && Y0 + lane < rows) {
            const int seam = right_of ? X0 / SEAM : (X0 + CFG::SW) / SEAM;
            st.clevel = (uint32_t)ext.nseams * (uint32_t)rows * 2u;
            st.cdump = ext.colmap + fz * ext.cframe + ((size_t)(seam - 1) * rows + (Y0 + lane)) * 2 + (right_of ? 1 : 0);
            st.cl = reinterpret_cast<const uint8_t*>(obuf + CFG::OBUF) + lane * (CFG::OBP * 4) + (right_of ? 0 : CFG::SW - 1);
        }
        const uint32_t mc = (uint32_t)min(max(ext.min_contrast, 0), 256);
        st.mc2 = mc | (mc << 16);
    }
    const bool full = tile_y0 + Yw + 16 <= rows && tile_x0 + CFG::SW <= cols;  // wave-uniform
    if constexpr (CFG::PAIRED) {
        uint32_t pg[CFG::NOB];
        mx_v4i tf[3];
        tf[0] = taps->b1[0][0][lane], tf[1] = taps->a2[0][0][lane], tf[2] = taps->a2o[0][lane];
        const mx_v4i* tapl = reinterpret_cast<const mx_v4i*>(smem + CFG::TAPL_OFF);
        if (full) {
            mx16_level_paired<CFG, 0, EXT, true>(taps, ln, pg, st, tf, tapl);
            mx16_level_paired<CFG, 1, EXT, true>(taps, ln, pg, st, tf, tapl);
            mx16_level_paired<CFG, 2, EXT, true>(taps, ln, pg, st, tf, tapl);
            mx16_level_paired<CFG, 3, EXT, true>(taps, ln, pg, st, tf, tapl);
            mx16_level_paired<CFG, 4, EXT, true>(taps, ln, pg, st, tf, tapl);
            mx16_level_paired<CFG, 5, EXT, true>(taps, ln, pg, st, tf, tapl);
        } else {
            mx16_level_paired<CFG, 0, EXT, false>(taps, ln, pg, st, tf, tapl);
            mx16_level_paired<CFG, 1, EXT, false>(taps, ln, pg, st, tf, tapl);
            mx16_level_paired<CFG, 2, EXT, false>(taps, ln, pg, st, tf, tapl);
            mx16_level_paired<CFG, 3, EXT, false>(taps, ln, pg, st, tf, tapl);
            mx16_level_paired<CFG, 4, EXT, false>(taps, ln, pg, st, tf, tapl);
            mx16_level_paired<CFG, 5, EXT, false>(taps, ln, pg, st, tf, tapl);
        }
    } else {
        uint32_t pe[CFG::NOB], po[CFG::NOB];
        mx_v4i ta[2 * CFG::NSMAX], tb[2 * CFG::NSMAX];
#pragma unroll
        for (int s = 0; s < CFG::ns(0); ++s) ta[s] = taps->b1[0][s][lane], ta[CFG::NSMAX + s] = taps->a2[0][s][lane];
        if (full) {
            mx16_level<CFG, 0, EXT, true>(taps, ln, pe, po, st, ta, tb);
            mx16_level<CFG, 1, EXT, true>(taps, ln, pe, po, st, tb, ta);
            mx16_level<CFG, 2, EXT, true>(taps, ln, pe, po, st, ta, tb);
            mx16_level<CFG, 3, EXT, true>(taps, ln, pe, po, st, tb, ta);
            mx16_level<CFG, 4, EXT, true>(taps, ln, pe, po, st, ta, tb);
            mx16_level<CFG, 5, EXT, true>(taps, ln, pe, po, st, tb, ta);
        } else {
            mx16_level<CFG, 0, EXT, false>(taps, ln, pe, po, st, ta, tb);
            mx16_level<CFG, 1, EXT, false>(taps, ln, pe, po, st, tb, ta);
            mx16_level<CFG, 2, EXT, false>(taps, ln, pe, po, st, ta, tb);
            mx16_level<CFG, 3, EXT, false>(taps, ln, pe, po, st, tb, ta);
            mx16_level<CFG, 4, EXT, false>(taps, ln, pe, po, st, ta, tb);
            mx16_level<CFG, 5, EXT, false>(taps, ln, pe, po, st, tb, ta);
        }
    }
}

// Host side: the quantised taps as 16 x 16 x 64 operand fragments (the K order of pass 2 is the one the hand-off produces).
//   b1[l][s][lane] byte j: pass-1 B[k][y'], y' = lane & 15, k = 64 s + 16 (lane >> 4) + j: input row -off + k relative to the output
//                          block's first row; tap index (-off + k) - y' + r.
//   a2o[l][lane]: as a2 with the window one block earlier (the odd block of a pair, mx16_level_paired).
//   a2[l][s][lane] byte j: pass-2 A[x'][k], x' = lane & 15, K slot (g = lane >> 4, j): input column -off + 16 (4 s + (j >> 2)) + 4 g + (j & 3)
//                          relative to the output block's first column (block 4 s + (j >> 2) of the window, the column C1's lane group g
//                          holds in register j & 3); tap index (that) - x' + r.
template <class CFG>
static bool mx16_pack_taps(const uint16_t* const t[6], Mx16Taps<CFG>& out) {
    std::memset(&out, 0, sizeof(out));
    for (int l = 0; l < 6; ++l) {
        const int n = CFG::n(l), r = n / 2, off = CFG::off(l);
        auto tap = [&](int idx) -> int { return idx >= 0 && idx < n ? (int)t[l][idx] : 0; };
        for (int k = 0; k < n; ++k)
            if (t[l][k] > 127) return false;  // signed 8-bit operands
        for (int s = 0; s < CFG::ns(l); ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int q = lane & 15, g = lane >> 4;
                int8_t b[16], a[16];
                for (int j = 0; j < 16; ++j) {
                    b[j] = (int8_t)tap((-off + 64 * s + 16 * g + j) - q + r);
                    a[j] = (int8_t)tap((-off + 16 * (4 * s + (j >> 2)) + 4 * g + (j & 3)) - q + r);
                }
                std::memcpy(&out.b1[l][s][lane], b, 16);
                std::memcpy(&out.a2[l][s][lane], a, 16);
            }
        // the odd output block of a pair (PAIRED configurations): its window starts one block (16 columns) before the even block's
        for (int lane = 0; lane < 64; ++lane) {
            const int q = lane & 15, g = lane >> 4;
            int8_t a[16];
            for (int j = 0; j < 16; ++j) a[j] = (int8_t)tap((-off - 16 + 16 * (j >> 2) + 4 * g + (j & 3)) - q + r);
            std::memcpy(&out.a2o[l][lane], a, 16);
        }
    }
    return true;
}

// The reference's fixed pyramid (sigma0 = 1.6), octaves 0 and 1: zero-trimmed widths as MxCfgOct0 / MxCfgOct1.
#ifndef VSLAM_MX16_TH
#define VSLAM_MX16_TH 128  // (tools/mx16_knockout.sh th64: 128 x 64 tiles of four waves, three workgroups per CU - an A/B build)
#endif
using Mx16CfgOct0 = Mx16Cfg<VSLAM_MX16_TH, 9, 13, 15, 19, 23, 29>;
using Mx16CfgOct1 = Mx16Cfg<128, 19, 23, 29, 37, 45, 57>;

}  // namespace vslam
