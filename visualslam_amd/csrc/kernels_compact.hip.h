// Deterministic ordered compaction of ballot flag words into keypoint lists, any batch size:
//   k_flag_count   : one 256-thread block per chunk of 256 flag entries -> chunk totals
//   k_chunk_scan   : one block per frame: exclusive scan of its chunk totals, frame total
//   k_flag_scatter : per chunk again: in-block exclusive scan (wave shuffles + LDS); then ONE THREAD
//                    PER RECORD: thread r of the chunk's records finds its entry by binary search in
//                    the chunk's offsets (LDS) and its bit by rank-select inside the entry's words,
//                    so all 256 threads write records and consecutive threads write consecutive
//                    list slots.  (Round 1 let the entry's thread walk its own bits - up to 60
//                    serial iterations with most lanes idle: 0.6 ms per 256-frame launch.)
// List order = entry order, and inside an entry ascending bit order, which is the reference's
// loop order (row-major pixels for Harris; octave, level, i, j for the DoG lattice).  Three
// short launches instead of one serial workgroup per frame.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels_generic.hip.h"
#include "kernels_harris_strip.hip.h"

namespace vslam {

constexpr int CMP_CHUNK = 256;  // entries per workgroup

// index of the k-th (0-based) set bit of w; k < popcount(w)
__device__ __forceinline__ int select64(unsigned long long w, unsigned int k) {
    unsigned int x = (unsigned int)w;
    int pos = 0;
    unsigned int c = __popc(x);
    if (k >= c) k -= c, pos = 32, x = (unsigned int)(w >> 32);
#pragma unroll
    for (int s = 16; s >= 1; s >>= 1) {
        c = __popc(x & ((1u << s) - 1u));
        if (k >= c) k -= c, pos += s, x >>= s;
    }
    return pos;
}

// Entry kinds: what one flag entry is (load), what its thread precomputes once per entry (Info,
// parked in LDS) and how the k-th set bit of the entry becomes a list record (emit).
struct HarrisStripEntries {  // entry = (row, strip): 4 ballot words (pixel slot k = 0..3), lane-major order
    static constexpr int WORDS = 4;
    struct Info {
        int row, c0;
    };
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
    __device__ Info info(int, size_t e) const {
        const int r = (int)((unsigned int)e / (unsigned int)nstrips), strip = (int)((unsigned int)e % (unsigned int)nstrips);
        return Info{r, strip * HS_STRIP_W - 8};
    }
    // records of an entry in (lane, slot) order = ascending columns
    __device__ void emit(int f, const Info& in, const unsigned long long (&w)[4], unsigned int k, size_t slot) const {
        int l = 0;  // largest lane with fewer than k+1 flags below it
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const unsigned long long m = (1ull << (l + s)) - 1ull;
            const unsigned int c = __popcll(w[0] & m) + __popcll(w[1] & m) + __popcll(w[2] & m) + __popcll(w[3] & m);
            if (c <= k) l += s;
        }
        const unsigned long long m = (1ull << l) - 1ull;
        k -= __popcll(w[0] & m) + __popcll(w[1] & m) + __popcll(w[2] & m) + __popcll(w[3] & m);
        const unsigned int nib = (unsigned int)((w[0] >> l) & 1ull) | ((unsigned int)((w[1] >> l) & 1ull) << 1) |
                                 ((unsigned int)((w[2] >> l) & 1ull) << 2) | ((unsigned int)((w[3] >> l) & 1ull) << 3);
        unsigned int x = nib;
        for (unsigned int i = 0; i < k; ++i) x &= x - 1;  // drop the k lowest set bits (k <= 3)
        const int c = in.c0 + 4 * l + (__ffs((int)x) - 1);
        vslam_kp kp;
        kp.row = in.row;
        kp.col = c;
        kp.response = resp[f * rframe + (size_t)in.row * cols + c];
        out[slot] = kp;
    }
};

struct DogEntries {  // entry = one 64-site word of the (octave, level, lattice row) bitmask layout
    static constexpr int WORDS = 1;
    struct Info {
        int octave, level, i, j0, step;
        unsigned int row_off;  // byte offset of DoG row (level, i - pad) in the frame's pyramid block
    };
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
    __device__ Info info(int, size_t e) const {
        const size_t wi = first() + e;
        int o = o_begin;
        while (o + 1 < o_end && wi >= g.bits_off[o + 1]) ++o;
        const unsigned int wl = (unsigned int)(wi - g.bits_off[o]);
        const unsigned int wpr = g.wpr[o], lr = g.lat_rows[o];
        const int level = (int)(wl / (lr * wpr)) + 1;
        const int li = (int)((wl / wpr) % lr), lj0 = (int)(wl % wpr) * 64;
        const size_t P = (size_t)g.rows[o] * g.pitch[o];
        const int i = g.pad + li * g.window;
        return Info{o, level, i, g.pad + lj0 * g.window, g.window,
                    (unsigned int)(g.oct_off[o] + (size_t)(VSLAM_NUM_LEVELS + level) * P + (size_t)(i - g.pad) * g.pitch[o])};
    }
    __device__ void emit(int f, const Info& in, const unsigned long long (&w)[4], unsigned int k, size_t slot) const {
        const int j = in.j0 + select64(w[0], k) * in.step;
        vslam_point pt;
        pt.row = in.i;
        pt.col = j;
        pt.value = pyr[f * pframe + in.row_off + (j - g.pad)];
        if (g.localize) {
            // the list holds FeaturePointLocalization's survivors: the record carries the value that
            // function stores at Diff_of_Gauss.cpp:246, formed here while the record is written (it was a
            // pass of its own over the list, a millisecond on the side chain of a 256-frame batch)
            const int o = in.octave;
            const size_t P = (size_t)g.rows[o] * g.pitch[o];
            const uint8_t* dog = pyr + f * pframe + g.oct_off[o] + (size_t)VSLAM_NUM_LEVELS * P;
            int d_x, d_y, d_s, nv;
            dog_differences(dog, P, g.rows[o], g.cols[o], g.pitch[o], g.pad, in.level, in.i, j, d_x, d_y, d_s);
            feature_point_localization(d_x, d_y, d_s, pt.value, nv, g.loc_lut);
            pt.value = nv;
        }
        pt.padding = g.pad;
        pt.octave = in.octave;
        pt.level = in.level;
        out[slot] = pt;
    }
};

struct OrientEntries {  // entry = one keypoint of filterKeypoints: 36-bit mask of histogram peaks
    static constexpr int WORDS = 1;
    struct Info {
        int row, col, octave, level;
    };
    const unsigned long long* masks;
    const vslam_point* kps;
    size_t n;
    vslam_point* out;
    __device__ size_t count() const { return n; }
    __device__ unsigned int load(int, size_t e, unsigned long long (&w)[4]) const {
        w[0] = masks[e];
        return __popcll(w[0]);
    }
    __device__ Info info(int, size_t e) const {
        const vslam_point kp = kps[e];
        return Info{kp.row, kp.col, kp.octave, kp.level};
    }
    __device__ void emit(int, const Info& in, const unsigned long long (&w)[4], unsigned int k, size_t slot) const {
        vslam_point pt;  // SLAM::point{y, x, angle, 0, octave, level}, Diff_of_Gauss.cpp:365
        pt.row = in.row;
        pt.col = in.col;
        pt.value = select64(w[0], k) * 10;
        pt.padding = 0;
        pt.octave = in.octave;
        pt.level = in.level;
        out[slot] = pt;
    }
};

struct SurvivorEntries {  // entry = 64 consecutive list records of a frame: bits = records that pass the edge test
    static constexpr int WORDS = 1;
    struct Info {
        unsigned int first;
    };
    const unsigned long long* flags;
    size_t fwords;
    unsigned int* surv;  // [frame][scap] record indices, ascending
    __device__ size_t count() const { return fwords; }
    __device__ unsigned int load(int f, size_t e, unsigned long long (&w)[4]) const {
        w[0] = flags[f * fwords + e];
        return __popcll(w[0]);
    }
    __device__ Info info(int, size_t e) const { return Info{(unsigned int)(e * 64)}; }
    __device__ void emit(int, const Info& in, const unsigned long long (&w)[4], unsigned int k, size_t slot) const {
        surv[slot] = in.first + (unsigned int)select64(w[0], k);
    }
};

struct OrientBatchEntries {  // entry = one survivor of a frame: 36-bit mask of its histogram peaks
    static constexpr int WORDS = 1;
    struct Info {
        int row, col, octave, level;
    };
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
    __device__ Info info(int f, size_t e) const {
        const vslam_point kp = pts[(size_t)f * pcap + surv[(size_t)f * scap + e]];
        return Info{kp.row, kp.col, kp.octave, kp.level};
    }
    __device__ void emit(int, const Info& in, const unsigned long long (&w)[4], unsigned int k, size_t slot) const {
        vslam_point pt;  // SLAM::point{y, x, angle, 0, octave, level}, Diff_of_Gauss.cpp:365
        pt.row = in.row;
        pt.col = in.col;
        pt.value = select64(w[0], k) * 10;
        pt.padding = 0;
        pt.octave = in.octave;
        pt.level = in.level;
        out[slot] = pt;
    }
};

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

// grid = (chunks, 1, frames).  Thread t first plays entry t of the chunk (count, exclusive offset,
// per-entry Info), then record t, t + 256, ... of the chunk.
template <class E>
__global__ __launch_bounds__(256) void k_flag_scatter(const E ent, const unsigned int* __restrict__ chunk_off, int nchunks,
                                                       unsigned int cap) {
    __shared__ unsigned int wsum[4];
    __shared__ unsigned int s_ex[CMP_CHUNK];
    __shared__ unsigned long long s_w[E::WORDS][CMP_CHUNK];
    __shared__ typename E::Info s_info[CMP_CHUNK];
    const int f = blockIdx.z, t = threadIdx.x;
    const size_t e = (size_t)blockIdx.x * CMP_CHUNK + t;
    unsigned long long w[4] = {0, 0, 0, 0};
    const unsigned int cnt = e < ent.count() ? ent.load(f, e, w) : 0u;
    unsigned int total;
    s_ex[t] = block_excl_scan_256(cnt, wsum, total);
#pragma unroll
    for (int k = 0; k < E::WORDS; ++k) s_w[k][t] = w[k];
    if (cnt) s_info[t] = ent.info(f, e);
    __syncthreads();
    const unsigned int base = chunk_off[(size_t)f * nchunks + blockIdx.x];
    if (base >= cap) return;  // block-uniform: nothing of this chunk fits
    for (unsigned int r = t; r < total; r += CMP_CHUNK) {
        if (base + r >= cap) break;
        // the entry holding record r: the last one whose exclusive offset is <= r (it is non-empty)
        int lo = 0;
#pragma unroll
        for (int s = CMP_CHUNK / 2; s >= 1; s >>= 1)
            if (s_ex[lo + s] <= r) lo += s;
        unsigned long long we[4] = {0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < E::WORDS; ++k) we[k] = s_w[k][lo];
        ent.emit(f, s_info[lo], we, r - s_ex[lo], (size_t)f * cap + base + r);
    }
}

// ---- packing the per-frame lists back to back (vslam_pack_lists_dev) ---------------------------------
// one block: offsets[f] = sum over g < f of min(counts[g], cap), f = 0..n (64-bit: n * cap may exceed 2^32)
__global__ __launch_bounds__(256) void k_pack_offsets(const unsigned int* __restrict__ counts, unsigned int cap, int n,
                                                       unsigned long long* __restrict__ offsets) {
    __shared__ unsigned long long sc[256];
    const int t = threadIdx.x;
    unsigned long long running = 0;
    for (int base = 0; base < n; base += 256) {
        const int i = base + t;
        const unsigned long long v = i < n ? (unsigned long long)min(counts[i], cap) : 0ull;
        sc[t] = v;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {  // Hillis-Steele inclusive scan
            const unsigned long long a = t >= off ? sc[t - off] : 0ull;
            __syncthreads();
            sc[t] += a;
            __syncthreads();
        }
        if (i < n) offsets[i] = running + sc[t] - v;
        running += sc[255];
        __syncthreads();
    }
    if (t == 0) offsets[n] = running;
}

// grid = (blocks per frame, 1, frames): dword-wise copy of frame f's first min(counts[f], cap) records to
// packed + offsets[f] records; consecutive threads move consecutive dwords on both sides.
__global__ __launch_bounds__(256) void k_pack_copy(const unsigned int* __restrict__ lists, unsigned int rec_dw, unsigned int cap,
                                                    const unsigned int* __restrict__ counts,
                                                    const unsigned long long* __restrict__ offsets,
                                                    unsigned int* __restrict__ packed, unsigned long long packed_dw) {
    const int f = blockIdx.z;
    const unsigned long long ndw = (unsigned long long)min(counts[f], cap) * rec_dw;
    const unsigned int* src = lists + (size_t)f * cap * rec_dw;
    const unsigned long long d0 = offsets[f] * rec_dw;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < ndw; i += (unsigned long long)gridDim.x * 256)
        if (d0 + i < packed_dw) packed[d0 + i] = src[i];
}

// grid = (blocks per frame, 1, frames): frame f's first min(counts[f], cap) SLAM::point records (six ints) to 16-byte
// records {row, col, value, level | octave << 8 | padding << 16} at packed + offsets[f]; a thread per record: three 8-byte
// loads (records are 8-byte aligned), one 16-byte store.
__global__ __launch_bounds__(256) void k_pack_points16(const int2* __restrict__ lists, unsigned int cap, const unsigned int* __restrict__ counts,
                                                        const unsigned long long* __restrict__ offsets, uint4* __restrict__ packed,
                                                        unsigned long long packed_recs) {
    const int f = blockIdx.z;
    const unsigned int nrec = min(counts[f], cap);
    const int2* src = lists + (size_t)f * cap * 3;
    const unsigned long long d0 = offsets[f];
    for (unsigned int i = blockIdx.x * 256 + threadIdx.x; i < nrec; i += gridDim.x * 256) {
        const int2 a = src[3 * (size_t)i], b = src[3 * (size_t)i + 1], c = src[3 * (size_t)i + 2];  // (row, col) (value, padding) (octave, level)
        if (d0 + i < packed_recs)
            packed[d0 + i] = make_uint4((unsigned int)a.x, (unsigned int)a.y, (unsigned int)b.x,
                                        ((unsigned int)c.y & 0xffu) | (((unsigned int)c.x & 0xffu) << 8) | (((unsigned int)b.y & 0xffffu) << 16));
    }
}

// one block: totals[0] = sum of a[0..n), totals[1] = sum of b[0..n) (either may be null -> 0), 64-bit
__global__ __launch_bounds__(256) void k_count_totals(const unsigned int* __restrict__ a, const unsigned int* __restrict__ b, int n,
                                                       unsigned long long* __restrict__ totals) {
    __shared__ unsigned long long sa[4], sb[4];
    unsigned long long ta = 0, tb = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        if (a) ta += a[i];
        if (b) tb += b[i];
    }
    for (int off = 32; off >= 1; off >>= 1) {
        ta += __shfl_down(ta, off);
        tb += __shfl_down(tb, off);
    }
    if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = ta, sb[threadIdx.x >> 6] = tb;
    __syncthreads();
    if (threadIdx.x == 0) {
        totals[0] = sa[0] + sa[1] + sa[2] + sa[3];
        totals[1] = sb[0] + sb[1] + sb[2] + sb[3];
    }
}

}  // namespace vslam
