// Deterministic ordered compaction of ballot flag words into keypoint lists, any batch size:
//   k_flag_count   : one 256-thread block per chunk of 256 flag entries -> chunk totals
//   k_chunk_scan   : one block per frame: exclusive scan of its chunk totals, frame total
//   k_flag_scatter : per chunk again: in-block exclusive scan (wave shuffles + LDS), then every
//                    non-empty entry is emitted by a whole wave - lane b owns bit b of the entry's
//                    words, its list position is the entry's offset plus the popcount of the lower
//                    bits (v_mbcnt-style), so the records of one entry leave as one coalesced burst
//                    instead of one thread walking its bits while 63 lanes idle
// List order = entry order, and inside an entry ascending bit order, which is the reference's
// loop order (row-major pixels for Harris; octave, level, i, j for the DoG lattice).  Three
// short launches instead of one serial workgroup per frame: a single 1080p frame compacts in
// ~20 us instead of 250 us, and a 256-frame batch spreads over the whole chip.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_generic.hip.h"
#include "kernels_harris_strip.hip.h"

namespace vslam {

constexpr int CMP_CHUNK = 256;  // entries per workgroup

// Entry kinds: what one flag entry is and how its set bits turn into list records.
struct HarrisStripEntries {  // entry = (row, strip): 4 ballot words (pixel slot k = 0..3), lane-major order
    const unsigned long long* flags;
    size_t fframe;
    int rows, cols, nstrips;
    const float* resp;
    size_t rframe;
    vslam_kp* out;
    __device__ size_t count() const { return (size_t)rows * nstrips; }
    __device__ unsigned int load(int f, size_t e, unsigned long long (&w)[4]) const {
        const unsigned long long* F = flags + f * fframe + e * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = F[k];
        return __popcll(w[0]) + __popcll(w[1]) + __popcll(w[2]) + __popcll(w[3]);
    }
    // whole wave, uniform (f, e, w, pos): lane l owns pixel slots k = 0..3 of its dword; records in
    // (lane, k) order = ascending columns
    __device__ void emit_wave(int f, size_t e, const unsigned long long (&w)[4], unsigned int pos, unsigned int cap, int lane) const {
        const int r = (int)(e / nstrips), strip = (int)(e % nstrips);
        const unsigned long long below = (1ull << lane) - 1ull;
        pos += __popcll(w[0] & below) + __popcll(w[1] & below) + __popcll(w[2] & below) + __popcll(w[3] & below);
        const int c0 = strip * HS_STRIP_W + 4 * (lane - 2);
        const float* rp = resp + f * rframe + (size_t)r * cols + c0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((w[k] >> lane) & 1ull) {
                if (pos < cap) {
                    vslam_kp kp;
                    kp.row = r;
                    kp.col = c0 + k;
                    kp.response = rp[k];
                    out[(size_t)f * cap + pos] = kp;
                }
                ++pos;
            }
    }
};

struct DogEntries {  // entry = one 64-site word of the (octave, level, lattice row) bitmask layout
    const unsigned long long* lflags;
    size_t bframe;
    const uint8_t* pyr;
    size_t pframe;
    ExtGeom g;
    int o_begin, o_end;
    vslam_point* out;
    __device__ size_t first() const { return g.bits_off[o_begin]; }
    __device__ size_t count() const {
        return g.bits_off[o_end - 1] + (size_t)3 * g.lat_rows[o_end - 1] * g.wpr[o_end - 1] - g.bits_off[o_begin];
    }
    __device__ unsigned int load(int f, size_t e, unsigned long long (&w)[4]) const {
        w[0] = lflags[f * bframe + first() + e];
        return __popcll(w[0]);
    }
    __device__ void emit_wave(int f, size_t e, const unsigned long long (&w)[4], unsigned int pos, unsigned int cap, int lane) const {
        const size_t wi = first() + e;
        int o = o_begin;
        while (o + 1 < o_end && wi >= g.bits_off[o + 1]) ++o;
        const size_t wl = wi - g.bits_off[o];
        const int wpr = g.wpr[o], lr = g.lat_rows[o];
        const int level = (int)(wl / ((size_t)lr * wpr)) + 1;
        const int li = (int)((wl / wpr) % lr), lj0 = (int)(wl % wpr) * 64;
        const size_t P = (size_t)g.rows[o] * g.pitch[o];
        const uint8_t* dog = pyr + f * pframe + g.oct_off[o] + (size_t)VSLAM_NUM_LEVELS * P;
        if (!((w[0] >> lane) & 1ull)) return;
        pos += __popcll(w[0] & ((1ull << lane) - 1ull));
        if (pos >= cap) return;
        const int i = g.pad + li * g.window, j = g.pad + (lj0 + lane) * g.window;
        vslam_point pt;
        pt.row = i;
        pt.col = j;
        pt.value = dog[(size_t)level * P + (size_t)(i - g.pad) * g.pitch[o] + (j - g.pad)];
        pt.padding = g.pad;
        pt.octave = o;
        pt.level = level;
        out[(size_t)f * cap + pos] = pt;
    }
};

struct OrientEntries {  // entry = one keypoint of filterKeypoints: 36-bit mask of histogram peaks
    const unsigned long long* masks;
    const vslam_point* kps;
    size_t n;
    vslam_point* out;
    __device__ size_t count() const { return n; }
    __device__ unsigned int load(int, size_t e, unsigned long long (&w)[4]) const {
        w[0] = masks[e];
        return __popcll(w[0]);
    }
    __device__ void emit_wave(int, size_t e, const unsigned long long (&w)[4], unsigned int pos, unsigned int cap, int lane) const {
        if (!((w[0] >> lane) & 1ull)) return;
        pos += __popcll(w[0] & ((1ull << lane) - 1ull));
        if (pos >= cap) return;
        const vslam_point kp = kps[e];
        vslam_point pt;  // SLAM::point{y, x, angle, 0, octave, level}, Diff_of_Gauss.cpp:365
        pt.row = kp.row;
        pt.col = kp.col;
        pt.value = lane * 10;
        pt.padding = 0;
        pt.octave = kp.octave;
        pt.level = kp.level;
        out[pos] = pt;
    }
};

struct SurvivorEntries {  // entry = 64 consecutive list records of a frame: bits = records that pass the edge test
    const unsigned long long* flags;
    size_t fwords;
    unsigned int* surv;  // [frame][scap] record indices, ascending
    __device__ size_t count() const { return fwords; }
    __device__ unsigned int load(int f, size_t e, unsigned long long (&w)[4]) const {
        w[0] = flags[f * fwords + e];
        return __popcll(w[0]);
    }
    __device__ void emit_wave(int f, size_t e, const unsigned long long (&w)[4], unsigned int pos, unsigned int cap, int lane) const {
        if (!((w[0] >> lane) & 1ull)) return;
        pos += __popcll(w[0] & ((1ull << lane) - 1ull));
        if (pos < cap) surv[(size_t)f * cap + pos] = (unsigned int)(e * 64 + lane);
    }
};

struct OrientBatchEntries {  // entry = one survivor of a frame: 36-bit mask of its histogram peaks
    const unsigned long long* masks;
    const unsigned int* surv;
    const unsigned int* scounts;
    unsigned int scap;
    const vslam_point* pts;
    unsigned int pcap;
    vslam_point* out;
    __device__ size_t count() const { return scap; }
    __device__ unsigned int load(int f, size_t e, unsigned long long (&w)[4]) const {
        w[0] = e < min(scounts[f], scap) ? masks[(size_t)f * scap + e] : 0ull;
        return __popcll(w[0]);
    }
    __device__ void emit_wave(int f, size_t e, const unsigned long long (&w)[4], unsigned int pos, unsigned int cap, int lane) const {
        if (!((w[0] >> lane) & 1ull)) return;
        pos += __popcll(w[0] & ((1ull << lane) - 1ull));
        if (pos >= cap) return;
        const vslam_point kp = pts[(size_t)f * pcap + surv[(size_t)f * scap + e]];
        vslam_point pt;  // SLAM::point{y, x, angle, 0, octave, level}, Diff_of_Gauss.cpp:365
        pt.row = kp.row;
        pt.col = kp.col;
        pt.value = lane * 10;
        pt.padding = 0;
        pt.octave = kp.octave;
        pt.level = kp.level;
        out[(size_t)f * cap + pos] = pt;
    }
};

// localize mode: rewrite the value of the points one octave has just appended the way
// FeaturePointLocalization does at Diff_of_Gauss.cpp:246.  One thread per list record (dense,
// unlike the per-word emit loop); records [begins[f], min(counts[f], cap)) of frame f, begins ==
// nullptr meaning 0.  grid = (ceil(most records an octave can add / 256), frames).
__global__ __launch_bounds__(256) void k_points_localize_value(vslam_point* __restrict__ pts, const unsigned int* __restrict__ begins,
                                                               const unsigned int* __restrict__ counts, unsigned int cap,
                                                               const uint8_t* __restrict__ pyr, size_t pframe, ExtGeom g) {
    const int f = blockIdx.y;
    const unsigned int i = (begins ? begins[f] : 0u) + blockIdx.x * 256 + threadIdx.x;
    if (i >= min(counts[f], cap)) return;
    vslam_point& pt = pts[(size_t)f * cap + i];
    const int o = pt.octave;
    const size_t P = (size_t)g.rows[o] * g.pitch[o];
    const uint8_t* dog = pyr + f * pframe + g.oct_off[o] + (size_t)VSLAM_NUM_LEVELS * P;
    int d_x, d_y, d_s, nv;
    dog_differences(dog, P, g.rows[o], g.cols[o], g.pitch[o], g.pad, pt.level, pt.row, pt.col, d_x, d_y, d_s);
    feature_point_localization(d_x, d_y, d_s, pt.value, nv, g.loc_lut);
    pt.value = nv;
}

// Block-wide exclusive scan for 256 threads; `total` = block sum.
__device__ __forceinline__ unsigned int block_excl_scan_256(unsigned int v, unsigned int* wsum, unsigned int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned int inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned int s = wsum[w];
        if (w < wave) base += s;
        tot += s;
    }
    total = tot;
    return base + inc - v;
}

// grid = (chunks, 1, frames)
template <class E>
__global__ __launch_bounds__(256) void k_flag_count(const E ent, unsigned int* __restrict__ chunk_tot, int nchunks) {
    __shared__ unsigned int wsum[4];
    const int f = blockIdx.z;
    const size_t e = (size_t)blockIdx.x * CMP_CHUNK + threadIdx.x;
    unsigned long long w[4] = {0, 0, 0, 0};
    const unsigned int cnt = e < ent.count() ? ent.load(f, e, w) : 0u;
    unsigned int total;
    block_excl_scan_256(cnt, wsum, total);
    if (threadIdx.x == 0) chunk_tot[(size_t)f * nchunks + blockIdx.x] = total;
}

// grid = (frames); turns chunk totals into exclusive offsets (in place) and writes the frame total.
// append != 0: the list continues after counts[f] records already written by an earlier call.
__global__ __launch_bounds__(256) void k_chunk_scan(unsigned int* __restrict__ chunk_tot, int nchunks,
                                                     unsigned int* __restrict__ counts, int append) {
    __shared__ unsigned int wsum[4];
    const int f = blockIdx.x;
    unsigned int* T = chunk_tot + (size_t)f * nchunks;
    unsigned int running = append ? counts[f] : 0u;
    for (int base = 0; base < nchunks; base += 256) {
        const int i = base + threadIdx.x;
        const unsigned int v = i < nchunks ? T[i] : 0u;
        unsigned int total;
        const unsigned int ex = block_excl_scan_256(v, wsum, total);
        if (i < nchunks) T[i] = running + ex;
        running += total;
        __syncthreads();  // wsum reuse
    }
    if (threadIdx.x == 0) counts[f] = running;
}

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int src_lane) {
    const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)v, src_lane);
    const unsigned int hi = __builtin_amdgcn_readlane((unsigned int)(v >> 32), src_lane);
    return ((unsigned long long)hi << 32) | lo;
}

// grid = (chunks, 1, frames).  Thread t of the block first plays entry t (count, exclusive offset);
// then each wave walks the non-empty entries of its own 64 lanes: the entry's words and offset are
// broadcast from the owning lane (v_readlane, the index is scalar) and all 64 lanes emit together.
template <class E>
__global__ __launch_bounds__(256) void k_flag_scatter(const E ent, const unsigned int* __restrict__ chunk_off, int nchunks,
                                                       unsigned int cap) {
    __shared__ unsigned int wsum[4];
    const int f = blockIdx.z;
    const int lane = threadIdx.x & 63;
    const size_t e = (size_t)blockIdx.x * CMP_CHUNK + threadIdx.x;
    unsigned long long w[4] = {0, 0, 0, 0};
    const unsigned int cnt = e < ent.count() ? ent.load(f, e, w) : 0u;
    unsigned int total;
    const unsigned int ex = block_excl_scan_256(cnt, wsum, total) + chunk_off[(size_t)f * nchunks + blockIdx.x];
    unsigned long long todo = __ballot(cnt != 0);
    const size_t e0 = e - lane;
    while (todo) {  // wave-uniform
        const int i = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        unsigned long long wi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) wi[k] = readlane_u64(w[k], i);
        const unsigned int pos = __builtin_amdgcn_readlane(ex, i);
        if (pos >= cap) break;  // offsets ascend: nothing further fits
        ent.emit_wave(f, e0 + i, wi, pos, cap, lane);
    }
}

}  // namespace vslam
