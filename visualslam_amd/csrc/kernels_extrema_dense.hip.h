// Extension (note under the table; not on the parity path): the dense 3x3x3
// scale-space test the north star's wording names.  The reference's initialKeypointDetection
// (Diff_of_Gauss.cpp:254-297) tests a half-open 2x2x3 window on a stride-3 lattice; this kernel
// applies the same rule (:282-287: candidate iff the value equals the minimum or the maximum of its
// window, ties included) and the same replicate border (padOctave, :260) to EVERY pixel of DoG
// levels 1..3 with the full 3x3x3 neighbourhood.
//
// One lane = 4 adjacent pixels (one dword of each DoG row), walking down a segment of rows:
//   per new row and level: the dword and its two neighbours (3 coalesced loads), the bytes x-1 / x+1
//   by v_alignbit, even / odd pixels widened into 16-bit lanes by v_perm, horizontal min / max by
//   v_pk_min_u16 / v_pk_max_u16;
//   a three-row ring of those per level gives the vertical pass, three levels the scale pass;
//   value == max or value == min as (v ^ max) and (v ^ min) having a zero 16-bit lane; the list flag
//   adds value >= min_contrast as a saturating subtraction;
//   a lane's four flags form a nibble, 16 lanes (one DPP row) OR their shifted nibbles together
//   (row_shr 1, 2, 4, 8) and lane 15 of the row stores the 64-pixel word.
// Bit layout: bit (x & 63) of word [((level-1)*rows + y)*words_per_row + x/64], words_per_row =
// ceil(cols/64).  Row pitch a multiple of 16 (the pyramid's), any width.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_aux.hip.h"
#include "kernels_compact.hip.h"
#include "kernels_pyramid.hip.h"

namespace vslam {

constexpr int XD_SEG_MAX = 32;  // output rows per lane: seg + 2 rows loaded for seg written; the host shortens it for small launches

struct DenseGeom {
    int rows, cols, pitch, wpr, min_contrast;
    int seg;                 // output rows per lane
    unsigned int P;          // bytes per plane
    unsigned int dog_off;    // byte offset of DoG level 0 of the octave inside the frame's pyramid block
};

__device__ __forceinline__ uint32_t pk_xor_zero_lanes_to_nibble(uint32_t tE, uint32_t tO) {
    // tE / tO: 16-bit lanes (pixel 0, pixel 2) / (pixel 1, pixel 3), zero = flag set.  Returns the nibble
    // of flags, bit k = pixel k.
    const uint32_t zE = pk_min_u16(tE, 0x00010001u), zO = pk_min_u16(tO, 0x00010001u);  // 0 = set, 1 = clear
    const uint32_t z = zE | (zO << 1);                                                   // bits 0, 1, 16, 17
    return ((z & 3u) | ((z >> 14) & 12u)) ^ 15u;
}

__device__ __forceinline__ unsigned long long row16_or(uint32_t nib, int lane) {
    // OR of nib << 4*(lane & 15) over the 16 lanes of a DPP row; valid in lane 15 of each row
    const uint32_t sh = nib << (4 * (lane & 7));
    uint32_t lo = (lane & 8) ? 0u : sh, hi = (lane & 8) ? sh : 0u;
#define VSLAM_ROW_OR(ctrl)                                                   \
    lo |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, ctrl, 0xf, 0xf, true); \
    hi |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, ctrl, 0xf, 0xf, true);
    VSLAM_ROW_OR(0x111)  // row_shr:1
    VSLAM_ROW_OR(0x112)  // row_shr:2
    VSLAM_ROW_OR(0x114)  // row_shr:4
    VSLAM_ROW_OR(0x118)  // row_shr:8
#undef VSLAM_ROW_OR
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}

// grid = (ceil(ceil(cols/4) / 256), ceil(rows / g.seg), frames); block = 256.
// bits (may be null) / lflags: [frame][3][rows][wpr] words.
__global__ __launch_bounds__(256) void k_extrema_dense(const uint8_t* __restrict__ pyr, size_t pframe, DenseGeom g,
                                                        unsigned long long* __restrict__ bits,
                                                        unsigned long long* __restrict__ lflags, size_t bframe) {
    const int lane = threadIdx.x & 63;
    const int nquads = (g.cols + 3) >> 2;
    const int quad_raw = blockIdx.x * 256 + threadIdx.x;
    const bool lane_in = quad_raw < nquads;
    const int quad = min(quad_raw, nquads - 1);  // lanes past the row compute on the last quad and contribute no bits
    const int x0 = 4 * quad;
    const int y0 = blockIdx.y * g.seg, y1 = min(y0 + g.seg, g.rows);
    const uint8_t* dog = pyr + blockIdx.z * pframe + g.dog_off;
    const int jedge = g.cols - x0;  // valid pixels of this quad when it is the row's last one (1..4)
    const bool last_quad = x0 + 4 >= g.cols;
    const uint32_t mc = (uint32_t)g.min_contrast * 0x00010001u;
    // pixels of the quad that exist (the last quad of a width that is not a multiple of 4)
    const uint32_t live = !lane_in ? 0u : (last_quad && jedge < 4 ? (1u << jedge) - 1u : 15u);

    // ring[l][slot] = horizontal (max even, max odd, min even, min odd) of a row; cen[l-1][slot] = its own even / odd pixels
    uint32_t hxE[5][3], hxO[5][3], hnE[5][3], hnO[5][3], cE[3][3], cO[3][3];

    auto load_row = [&](int y, int slot) {
        const int yc = min(max(y, 0), g.rows - 1);  // replicate rows
        const uint32_t roff = (uint32_t)yc * (uint32_t)g.pitch + (uint32_t)x0;
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            const uint8_t* p = dog + (size_t)l * g.P + roff;
            uint32_t C = *reinterpret_cast<const uint32_t*>(p);
            uint32_t Lf = x0 > 0 ? *reinterpret_cast<const uint32_t*>(p - 4) : C << 24;  // column -1 = column 0
            uint32_t Rt;
            if (last_quad) {  // columns >= cols repeat column cols-1 (the row padding holds anything)
                const uint32_t rep = ((C >> (8 * (jedge - 1))) & 0xffu) * 0x01010101u;
                const uint32_t keep = jedge >= 4 ? 0xffffffffu : (1u << (8 * jedge)) - 1u;
                C = (C & keep) | (rep & ~keep);
                Rt = rep;
            } else {
                Rt = *reinterpret_cast<const uint32_t*>(p + 4);
            }
            const uint32_t Lw = __builtin_amdgcn_alignbit(C, Lf, 24);  // pixels x0-1 .. x0+2
            const uint32_t Rw = __builtin_amdgcn_alignbit(Rt, C, 8);   // pixels x0+1 .. x0+4
            const uint32_t Ce = __builtin_amdgcn_perm(0u, C, 0x0c020c00u), Co = __builtin_amdgcn_perm(0u, C, 0x0c030c01u);
            const uint32_t Le = __builtin_amdgcn_perm(0u, Lw, 0x0c020c00u), Lo = __builtin_amdgcn_perm(0u, Lw, 0x0c030c01u);
            const uint32_t Re = __builtin_amdgcn_perm(0u, Rw, 0x0c020c00u), Ro = __builtin_amdgcn_perm(0u, Rw, 0x0c030c01u);
            hxE[l][slot] = pk_max_u16(pk_max_u16(Le, Ce), Re);
            hxO[l][slot] = pk_max_u16(pk_max_u16(Lo, Co), Ro);
            hnE[l][slot] = pk_min_u16(pk_min_u16(Le, Ce), Re);
            hnO[l][slot] = pk_min_u16(pk_min_u16(Lo, Co), Ro);
            if (l >= 1 && l <= 3) cE[l - 1][slot] = Ce, cO[l - 1][slot] = Co;
        }
    };

    // rows y-1, y, y+1 live in slots (y-1)%3, y%3, (y+1)%3 of the ring; the loop is unrolled by 3 so
    // that the slot indices are compile-time constants
    load_row(y0 - 1, 0);
    load_row(y0, 1);
    for (int yb = y0; yb < y1; yb += 3) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int y = yb + s;
            if (y >= y1) break;                       // wave-uniform
            const int sp = s, sc = (s + 1) % 3, sn = (s + 2) % 3;  // slots of rows y-1, y, y+1
            load_row(y + 1, sn);
            uint32_t vxE[5], vxO[5], vnE[5], vnO[5];
#pragma unroll
            for (int l = 0; l < 5; ++l) {
                vxE[l] = pk_max_u16(pk_max_u16(hxE[l][sp], hxE[l][sc]), hxE[l][sn]);
                vxO[l] = pk_max_u16(pk_max_u16(hxO[l][sp], hxO[l][sc]), hxO[l][sn]);
                vnE[l] = pk_min_u16(pk_min_u16(hnE[l][sp], hnE[l][sc]), hnE[l][sn]);
                vnO[l] = pk_min_u16(pk_min_u16(hnO[l][sp], hnO[l][sc]), hnO[l][sn]);
            }
#pragma unroll
            for (int L = 1; L <= 3; ++L) {
                const uint32_t mxE = pk_max_u16(pk_max_u16(vxE[L - 1], vxE[L]), vxE[L + 1]);
                const uint32_t mxO = pk_max_u16(pk_max_u16(vxO[L - 1], vxO[L]), vxO[L + 1]);
                const uint32_t mnE = pk_min_u16(pk_min_u16(vnE[L - 1], vnE[L]), vnE[L + 1]);
                const uint32_t mnO = pk_min_u16(pk_min_u16(vnO[L - 1], vnO[L]), vnO[L + 1]);
                const uint32_t vE = cE[L - 1][sc], vO = cO[L - 1][sc];
                const uint32_t tE = pk_min_u16(vE ^ mxE, vE ^ mnE), tO = pk_min_u16(vO ^ mxO, vO ^ mnO);  // zero lane = extremum
                const uint32_t uE = tE | pk_sub_sat_u16(mc, vE), uO = tO | pk_sub_sat_u16(mc, vO);       // ... and value >= min_contrast
                const uint32_t ncand = pk_xor_zero_lanes_to_nibble(tE, tO) & live;
                const uint32_t nlist = pk_xor_zero_lanes_to_nibble(uE, uO) & live;
                const unsigned long long wc = row16_or(ncand, lane), wl = row16_or(nlist, lane);
                const int word = quad_raw >> 4;
                if ((lane & 15) == 15 && word < g.wpr) {
                    const size_t w = blockIdx.z * bframe + ((size_t)(L - 1) * g.rows + y) * g.wpr + word;
                    if (bits) bits[w] = wc;
                    lflags[w] = wl;
                }
            }
        }
    }
}

// List emission through kernels_compact.hip.h: entry = one 64-pixel word of lflags, records in
// (level, row, col) order with the reference's padded coordinates (Diff_of_Gauss.cpp:289).
struct DenseDogEntries {
    static constexpr int WORDS = 1;
    struct Info {
        int level, y, x0;
        unsigned int row_off;  // byte offset of DoG row (level, y) in the frame's pyramid block
    };
    const unsigned long long* lflags;
    size_t bframe;
    const uint8_t* pyr;
    size_t pframe;
    DenseGeom g;
    int octave;
    vslam_point* out;
    __device__ size_t count() const { return (size_t)3 * g.rows * g.wpr; }
    __device__ unsigned int load(int f, size_t e, unsigned long long (&w)[4]) const {
        w[0] = lflags[f * bframe + e];
        return __popcll(w[0]);
    }
    __device__ Info info(int, size_t e) const {
        const unsigned int per_level = (unsigned int)g.rows * (unsigned int)g.wpr;
        const int level = (int)(e / per_level) + 1;
        const unsigned int r = (unsigned int)(e % per_level);
        const int y = (int)(r / (unsigned int)g.wpr), x0 = (int)(r % (unsigned int)g.wpr) * 64;
        return Info{level, y, x0, g.dog_off + (unsigned int)level * g.P + (unsigned int)y * (unsigned int)g.pitch};
    }
    __device__ void emit(int f, const Info& in, const unsigned long long (&w)[4], unsigned int k, size_t slot) const {
        const int x = in.x0 + select64(w[0], k);
        vslam_point pt;
        pt.row = in.y + 1;
        pt.col = x + 1;
        pt.value = pyr[f * pframe + in.row_off + x];
        pt.padding = 1;
        pt.octave = octave;
        pt.level = in.level;
        out[slot] = pt;
    }
};

}  // namespace vslam
