// Vectorised memory-side kernels of the DoG path for gfx950: 2x bilinear upsample (K-D0),
// 2:1 nearest decimation (K-D4) and the scale-space extrema scan for windowSize 3 (K-D3).
// All are HBM/L2-bound byte movers: coalesced 4/8/16-byte accesses, LDS row staging where a
// row is shared, one ballot word per 64 lattice sites.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_generic.hip.h"
#include "kernels_localize.hip.h"

namespace vslam {

// ---- K-D0: cv::resize(img, Size(), 2, 2, INTER_LINEAR) on CV_8U (GaussPyramid.cpp:110) ----
// Sliding form, ~3.4 VALU instructions per pixel.  A thread owns 4 source columns (= 8 destination columns) and walks down a segment of source
// rows; per source row the horizontal pass is done ONCE in 16-bit lanes and reused by the four
// destination rows it feeds.  With s = source pixels, the 11-bit fixed-point steps collapse to
//   A = s_i + 3 s_{i+1}  (or 3 s_i + s_{i+1})            h = 512 A,  h >> 4 = 32 A
//   (512*32A) >> 16 = A >> 2,   (1536*32A) >> 16 = (3A) >> 2
//   dst(2m)   = ((A(m-1) >> 2) + ((3 A(m)) >> 2) + 2) >> 2
//   dst(2m+1) = (((3 A(m)) >> 2) + (A(m+1) >> 2) + 2) >> 2
// which is bit-identical to the literal formula (k_resize_linear2x in kernels_generic.hip.h;
// clamped reads at the borders reproduce OpenCV's border rule because the weights sum to 2048).  Any width and source step; dpitch >= 2*cols rounded up to 8.  grid = (ceil(ceil(cols/4)/256), ceil(rows/seg), frames).
typedef unsigned short us2r_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_lshr_u16(uint32_t a, int sh) {
    return __builtin_bit_cast(uint32_t, (us2r_t)(__builtin_bit_cast(us2r_t, a) >> (unsigned short)sh));
}

__global__ __launch_bounds__(256) void k_resize_linear2x_slide(const uint8_t* __restrict__ src, size_t sstep, size_t sframe,
                                                                uint8_t* __restrict__ dst, size_t dframe, int dpitch,
                                                                int rows, int cols, int seg) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (4 * k >= cols) return;
    const bool whole = 4 * k + 3 < cols;  // last group of a width that is not a multiple of 4: clamp per byte
    const int m0 = blockIdx.y * seg, m1 = min(m0 + seg, rows);
    const uint8_t* s = src + blockIdx.z * sframe;
    uint8_t* d = dst + blockIdx.z * dframe + 8 * k;
    const int xl = max(4 * k - 1, 0), xr = min(4 * k + 4, cols - 1);
    // X = A >> 2 and Y = 3A >> 2 of one source row, as (j0,j2) (j1,j3) (j4,j6) (j5,j7) 16-bit pairs
    auto hrow = [&](int m, uint32_t (&X)[4], uint32_t (&Y)[4]) {
        const uint8_t* r = s + (size_t)min(max(m, 0), rows - 1) * sstep;
        uint32_t w;  // s1 s2 s3 s4; rows of an arbitrary width / step are not dword aligned: 4-byte copy
        if (whole)
            __builtin_memcpy(&w, r + 4 * k, 4);
        else
            w = (uint32_t)r[4 * k] | ((uint32_t)r[min(4 * k + 1, cols - 1)] << 8) | ((uint32_t)r[min(4 * k + 2, cols - 1)] << 16) |
                ((uint32_t)r[cols - 1] << 24);
        const uint32_t s0 = r[xl], s5 = r[xr];
        const uint32_t P = __builtin_amdgcn_perm(w, w, 0x0c010c00);   // (s1, s2)
        const uint32_t Q = __builtin_amdgcn_perm(w, w, 0x0c030c02);   // (s3, s4)
        const uint32_t Qm = __builtin_amdgcn_perm(w, w, 0x0c020c01);  // (s2, s3)
        const uint32_t Pm = s0 | ((w & 0xffu) << 16);                  // (s0, s1)
        const uint32_t Qp = (w >> 24) | (s5 << 16);                    // (s4, s5)
        const uint32_t P3 = P + (P << 1), Q3 = Q + (Q << 1);          // <= 765 per lane
        const uint32_t A[4] = {Pm + P3, P3 + Qm, Qm + Q3, Q3 + Qp};    // (j0,j2) (j1,j3) (j4,j6) (j5,j7)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            X[i] = pk_lshr_u16(A[i], 2);
            Y[i] = pk_lshr_u16(A[i] + (A[i] << 1), 2);  // 3A <= 3060 per lane
        }
    };
    auto emit = [&](int dy, const uint32_t (&U)[4], const uint32_t (&V)[4]) {
        uint32_t v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = pk_lshr_u16(U[i] + V[i] + 0x00020002u, 2);
        *reinterpret_cast<uint2*>(d + (size_t)dy * dpitch) = make_uint2(v[0] | (v[1] << 8), v[2] | (v[3] << 8));
    };
    uint32_t Xp[4], Yp[4], Xc[4], Yc[4];
    hrow(m0 - 1, Xp, Yp);
    hrow(m0, Xc, Yc);
    for (int m = m0; m < m1; ++m) {
        uint32_t Xn[4], Yn[4];
        hrow(m + 1, Xn, Yn);
        emit(2 * m, Xp, Yc);      // rows (m-1, m), weights (512, 1536)
        emit(2 * m + 1, Yc, Xn);  // rows (m, m+1), weights (1536, 512)
#pragma unroll
        for (int i = 0; i < 4; ++i) Xp[i] = Xc[i], Xc[i] = Xn[i], Yc[i] = Yn[i];
    }
}

// ---- K-D4: cv::resize(src, Size(), 0.5, 0.5, INTER_NEAREST) (GaussPyramid.cpp:126) ----------
// One thread = 4 destination pixels from 8 source bytes; spitch / dpitch are multiples of 16, so
// the 8-byte load and the 4-byte store may run into the row padding but never out of the row.
__global__ __launch_bounds__(256) void k_resize_nearest_half_v4(const uint8_t* __restrict__ src, size_t sframe, int spitch,
                                                                 uint8_t* __restrict__ dst, size_t dframe, int dpitch,
                                                                 int rows, int drows, int dcols) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if (4 * k >= dcols) return;
    const int sy = min(2 * y, rows - 1);
    const uint2 w = *reinterpret_cast<const uint2*>(src + blockIdx.z * sframe + (size_t)sy * spitch + 8 * k);
    *reinterpret_cast<uint32_t*>(dst + blockIdx.z * dframe + (size_t)y * dpitch + 4 * k) =
        __builtin_amdgcn_perm(w.y, w.x, 0x06040200);
}

// ---- K-D3: initialKeypointDetection for windowSize 3 (Diff_of_Gauss.cpp:254-297) -----------
// One workgroup = 256 consecutive sites (4 ballot words) of one lattice row of one frame, all
// three levels.  The 768-column span those sites touch, for the two image rows of the row's
// 2x2x3 windows (unpadded rows 3li-1 and 3li, clamped) and all five DoG levels, is staged with
// 16-byte loads (7.8 KB of LDS: small enough to co-reside with the pyramid kernel when the two
// run on different streams).  One thread per site: its two columns (3lj-1, 3lj) are adjacent
// bytes, fetched per staged row by one ds_read2_b32 of the enclosing dword pair and one v_perm
// with a per-site selector, landing as (a, b) in 16-bit lanes for packed min/max.  A wave's 64
// candidate flags leave as one ballot word = the bitmask layout of include/vslam.h.
// Row pitch a multiple of 16 (any width).  grid = (ceil(words_per_row/4), lattice rows to scan, frames).
typedef unsigned short us2e_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2e_t, a), __builtin_bit_cast(us2e_t, b)));
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2e_t, a), __builtin_bit_cast(us2e_t, b)));
}

constexpr int EXT_SPAN = 768;            // image columns per workgroup (256 sites x stride 3)
constexpr int EXT_PITCH = EXT_SPAN + 16; // 16 guard bytes in front: column -1 of the span

// LOC: the list flags are FeaturePointLocalization's verdict (Diff_of_Gauss.cpp:290) instead of
// the min_contrast threshold; that needs padded (i+1, j) and (i, j+1) of the three middle levels
// as well, so three more rows (unpadded 3li+1, clamped) are staged: 13 rows, 10 KB.
template <bool LOC>
__global__ __launch_bounds__(256) void k_extrema_w3(const uint8_t* __restrict__ pyr, size_t pframe, ExtGeom g, int o,
                                                     unsigned long long* __restrict__ bits,
                                                     unsigned long long* __restrict__ lflags, size_t bframe, int li0, int li_step) {
    constexpr int NROW = LOC ? 13 : 10;
    __shared__ __attribute__((aligned(16))) uint8_t srow[NROW * EXT_PITCH + 16];  // [level 0..4][row 0..1], then [level 1..3] row 2
    __shared__ uint2 queue[LOC ? 768 : 1];       // per wave: the sites whose localization needs the full inverse
    __shared__ uint8_t qkeep[LOC ? 768 : 1];     // per wave: their verdicts, by (level-1)*64 + lane
    // lattice row li0 + blockIdx.y * li_step: (0, 1) = every row; the matrix path's fused scan leaves only every 32nd row here
    const int li = li0 + (int)blockIdx.y * li_step, f = blockIdx.z;
    const int rows = g.rows[o], cols = g.cols[o], pitch = g.pitch[o];
    const uint32_t P = (uint32_t)rows * (uint32_t)pitch;  // 11 planes of an octave stay below 2^31 bytes
    const uint8_t* dog = pyr + f * pframe + g.oct_off[o] + (size_t)VSLAM_NUM_LEVELS * P;
    const int ya = max(3 * li - 1, 0), yb = 3 * li;  // padded rows i-1, i with i = 1 + 3li -> unpadded 3li-1, 3li
    const int yc = min(3 * li + 1, rows - 1);        // padded row i+1
    const int c0 = blockIdx.x * EXT_SPAN - 16;       // image column of staged byte 0
    // Staging: a wave takes whole staged rows (wave w: rows w, w+4, ...), lane = 16-byte piece, so
    // the level / image row / row base are scalar and a piece costs one load and one LDS write.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int cpiece = c0 + 16 * lane;
    const bool piece_ok = lane < EXT_PITCH / 16 && cpiece >= 0 && cpiece < cols;
#pragma unroll
    for (int i = 0; i < (NROW + 3) / 4; ++i) {
        const int rl = wave + 4 * i;  // rl = level*2 + row
        if (rl < NROW) {
            const int lev = rl < 10 ? (rl >> 1) : rl - 9, y = rl < 10 ? ((rl & 1) ? yb : ya) : yc;
            const uint8_t* rowp = dog + (ptrdiff_t)(int)((uint32_t)lev * P + (uint32_t)y * (uint32_t)pitch + (uint32_t)c0);  // scalar
            if (piece_ok)
                *reinterpret_cast<uint4*>(srow + rl * EXT_PITCH + 16 * lane) =
                    *reinterpret_cast<const uint4*>(rowp + (uint32_t)(16 * lane));  // may end in the row padding
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 10)  // column -1 replicates column 0 (padOctave)
        srow[threadIdx.x * EXT_PITCH + 15] = dog[(threadIdx.x >> 1) * P + (uint32_t)((threadIdx.x & 1) ? yb : ya) * (uint32_t)pitch];
    __syncthreads();
    const int lc = g.lat_cols[o], wpr = g.wpr[o], lr = g.lat_rows[o];
    const int lj = blockIdx.x * 256 + threadIdx.x;
    bool cand[3] = {false, false, false}, listed[3] = {false, false, false};
    unsigned long long wcand[3], wlist[3];
    // Every lane runs the window test (lanes past the last site read staged bytes that are there
    // but mean nothing) and the ballots are masked by the lanes that own a site: the masks stay
    // in scalar registers, built from the compares themselves.
    const unsigned long long owns = __builtin_amdgcn_ballot_w64(lj < lc);
    const int xa = 3 * (int)threadIdx.x - 1 + 16;       // byte offset of column 3lj-1 in a staged row
    uint32_t self[5];
    uint32_t up[5], left[5];  // LOC: padded (i-1, j) and (i, j-1) of each level, already in the window
    {
        const uint32_t s = xa & 3;
        const uint32_t sel = 0x0c000c00u | s | ((s + 1) << 16);  // (byte s, 0, byte s+1, 0)
        uint32_t mn[5], mx[5];
#pragma unroll
        for (int l = 0; l < 5; ++l) {
            // one address per level: the level's second row is within ds_read2_b32's offset range of the first
            uint32_t off = (uint32_t)(xa & ~3) + (uint32_t)(2 * l * EXT_PITCH);
            if (l) asm volatile("" : "+v"(off));
            const uint32_t* q0 = reinterpret_cast<const uint32_t*>(srow + off);  // 4-byte aligned: ds_read2_b32
            const uint32_t* q1 = q0 + EXT_PITCH / 4;
            const uint32_t v0 = __builtin_amdgcn_perm(q0[1], q0[0], sel), v1 = __builtin_amdgcn_perm(q1[1], q1[0], sel);  // (a,b), (c,d)
            const uint32_t lo = pk_min_u16(v0, v1), hi = pk_max_u16(v0, v1);
            mn[l] = min(lo & 0xffffu, lo >> 16);
            mx[l] = max(hi & 0xffffu, hi >> 16);
            self[l] = v1 >> 16;  // (i, j) itself = d
            if (LOC) up[l] = v0 >> 16, left[l] = v1 & 0xffffu;
        }
#pragma unroll
        for (int L = 1; L <= 3; ++L) {
            const uint32_t lo = min(mn[L - 1], min(mn[L], mn[L + 1])), hi = max(mx[L - 1], max(mx[L], mx[L + 1]));
            wcand[L - 1] = (__builtin_amdgcn_ballot_w64(self[L] == lo) | __builtin_amdgcn_ballot_w64(self[L] == hi)) & owns;
            if (LOC) cand[L - 1] = lj < lc && (self[L] == lo || self[L] == hi);
            else wlist[L - 1] = wcand[L - 1] & __builtin_amdgcn_ballot_w64((int)self[L] >= g.min_contrast);
        }
    }
    if (LOC) {
        // Candidates with three non-zero differences go through a queue OF THIS WAVE (slots from the
        // ballot, no atomics, no workgroup barrier: LDS serves a wave's accesses in order) and are
        // evaluated on dense lanes; a workgroup-wide queue made every workgroup wait twice for its
        // slowest wave and for one wave's table loads - the scan was bound by that latency chain.
        uint2* wq = queue + wave * 192;
        uint8_t* wk = qkeep + wave * 192;
        const int xj = xa + 1;                                            // byte offset of column 3lj
        const int xr = xj + ((3 * lj + 1 < cols) ? 1 : 0);                // padded (., j+1): replicate at the edge
        unsigned int nw = 0;                                              // wave-uniform queue length
        bool queued[3];
#pragma unroll
        for (int L = 1; L <= 3; ++L) {
            bool nz = false;
            uint32_t dpack = 0;
            if (cand[L - 1]) {
                const uint8_t* r1 = srow + (2 * L + 1) * EXT_PITCH;       // padded row i
                const uint8_t* r2 = srow + (9 + L) * EXT_PITCH;           // padded row i+1
                const int d_x = (int)left[L] - (int)r1[xr];               // Diff_of_Gauss.cpp:226
                const int d_y = (int)up[L] - (int)r2[xj];                 // :227
                const int d_s = (int)self[L - 1] - (int)self[L + 1];      // :228
                nz = d_x != 0 && d_y != 0 && d_s != 0;
                // exactly singular otherwise: the quadratic term is +-0 and the test is
                // value/255.0f > 0.03f, which holds from 8 on (8/255 = 0.03137, 7/255 = 0.02745; checked
                // over 0..255 in tests/test_oracle_kat.py::test_feature_point_localization_oracle)
                listed[L - 1] = !nz && self[L] >= 8u;
                dpack = (uint32_t)(d_x + 256) | ((uint32_t)(d_y + 256) << 10) | ((uint32_t)(d_s + 256) << 20);
            }
            const unsigned long long m = __builtin_amdgcn_ballot_w64(nz);
            if (nz) {
                const unsigned int slot = nw + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                wq[slot] = make_uint2(dpack, self[L] | ((uint32_t)((L - 1) * 64 + lane) << 8));
            }
            queued[L - 1] = nz;
            nw += (unsigned int)__builtin_popcountll(m);
        }
        if (nw) {  // wave-uniform
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (unsigned int q = lane; q < nw; q += 64) {
                const uint2 e = wq[q];
                int nv;
                // table lookup for small differences (all of them on ordinary frames), closed form otherwise
                wk[e.y >> 8] = feature_point_localization((int)(e.x & 1023u) - 256, (int)((e.x >> 10) & 1023u) - 256,
                                                          (int)(e.x >> 20) - 256, (int)(e.y & 255u), nv, g.loc_lut);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int L = 0; L < 3; ++L)
                if (queued[L]) listed[L] = wk[L * 64 + lane] != 0;
        }
    }
    const int word = blockIdx.x * 4 + wave;  // = lj >> 6, wave-uniform
    if (word < wpr) {
        const size_t w0 = f * bframe + g.bits_off[o] + (size_t)li * wpr + word;
#pragma unroll
        for (int L = 0; L < 3; ++L) {
            const unsigned long long wc = wcand[L], wl = LOC ? __builtin_amdgcn_ballot_w64(listed[L]) : wlist[L];
            if (lane == 0) {
                const size_t w = w0 + (size_t)L * lr * wpr;
                if (bits) bits[w] = wc;
                lflags[w] = wl;
            }
        }
    }
}


// ---- processGradients for one Gaussian level (GaussPyramid.cpp:65-104) ----
// Sobel x / y with ksize 1 (reflect-101), cv::magnitude and cv::phase(..., angleInDegrees=true)
// = OpenCV's fastAtan2 polynomial.  Materialised on demand (96 bytes per pyramid pixel if all
// levels were kept); any output pointer may be null.  One thread per pixel; every f32 operation
// is individually rounded (-ffp-contract=off) so the result equals the oracle's bit for bit.
// One f32 multiply-add of the two stages that run inside OpenCV's run-time-dispatched SIMD code (fastAtan32f's polynomial, the
// separable f32 filter's passes).  FMA = false: product and sum rounded separately - OpenCV's SSE2 baseline, the default, the
// oracle's variant 0.  FMA = true: one fused operation - its AVX2 + FMA3 dispatch, oracle.fma_variant (vslam_ctx_set_f32_fused;
// profiles/r06_fma_risk.json: what changes between the two).  The file is compiled with -ffp-contract=off, so a * b + c stays two
// operations.
template <bool FMA>
__device__ __forceinline__ float mad_f32(float a, float b, float c) {
    if constexpr (FMA) return __builtin_fmaf(a, b, c);
    else return a * b + c;
}

// FMA: the polynomial's three multiply-adds fused (its last bit only: for integer gradients in [-255, 255] NO 36-bin or 8-bin
// histogram index depends on it - tests/test_pin_readiness_cpu.py checks all 511 x 511 pairs - so the kernels that only BIN the
// angle call the default form; k_level_gradients, which returns the angle itself, carries the switch)
template <bool FMA = false>
__device__ __forceinline__ float fast_atan2_deg(float y, float x) {
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    const float eps = (float)2.2204460492503131e-16;
    const float ax = fabsf(x), ay = fabsf(y);
    float a;
    if (ax >= ay) {
        const float c = ay / (ax + eps), c2 = c * c;
        a = mad_f32<FMA>(mad_f32<FMA>(mad_f32<FMA>(p7, c2, p5), c2, p3), c2, p1) * c;
    } else {
        const float c = ax / (ay + eps), c2 = c * c;
        a = 90.f - mad_f32<FMA>(mad_f32<FMA>(mad_f32<FMA>(p7, c2, p5), c2, p3), c2, p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

template <bool FMA>
__global__ __launch_bounds__(256) void k_level_gradients(const uint8_t* __restrict__ g, int gpitch, int rows, int cols,
                                                          float* __restrict__ gx, float* __restrict__ gy,
                                                          float* __restrict__ mag, float* __restrict__ orient) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;
    if (c >= cols) return;
    const uint8_t* row = g + (size_t)r * gpitch;
    const float x = (float)((int)row[reflect101(c + 1, cols)] - (int)row[reflect101(c - 1, cols)]);
    const float y = (float)((int)g[(size_t)reflect101(r + 1, rows) * gpitch + c] - (int)g[(size_t)reflect101(r - 1, rows) * gpitch + c]);
    const size_t o = (size_t)r * cols + c;
    if (gx) gx[o] = x;
    if (gy) gy[o] = y;
    if (mag) {
        const float xx = x * x, yy = y * y;
        mag[o] = sqrt_rn_small_nr(xx + yy);  // IEEE-correct f32 square root (the bare v_sqrt_f32 is not)
    }
    if (orient) orient[o] = fast_atan2_deg<FMA>(y, x);
}

}  // namespace vslam
