// C ABI (include/vslam.h) over the hand-written gfx950 kernels.  Host code here only
// validates arguments, sizes the workspace and enqueues kernels on the context's stream;
// there is no CPU compute path: if HIP is unavailable every compute entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <string>
#include <set>
#include <memory>
#include <vector>

#include "../../include/vslam.h"
#include "kernels_generic.hip.h"
#include "kernels_pyramid.hip.h"
#include "kernels_strip.hip.h"
#include "kernels_aux.hip.h"
#include "kernels_harris_strip.hip.h"
#include "kernels_orient.hip.h"
#include "kernels_orient_batch.hip.h"
#include "kernels_orient_pk.hip.h"
#include "kernels_compact.hip.h"
#include "kernels_sift.hip.h"
#include "kernels_extrema_dense.hip.h"
#include "vslam_internal.h"
#include "vslam_ctx.h"
#include "vslam_mx.h"

using namespace vslam;


struct vslam_pyramid {
    vslam_ctx* ctx = nullptr;
    vslam_params params{};
    vslam_batch_layout layout{};
    vslam_pyramid_info info{};
    uint8_t* d_block = nullptr;  // one pyramid frame block (layout.pyramid_frame_bytes)
    uint8_t* d_bases = nullptr;  // octave bases, octave o at base_off[o]
    size_t block_cap = 0, bases_cap = 0;
    size_t base_off[VSLAM_MAX_OCTAVES] = {};
};


static const char* const kKernelNames =
    "k_resize_linear2x\nk_blur_h_generic\nk_blur_v_generic\n"
    "k_dog5\nk_resize_nearest_half\nk_extrema\nk_pyr_octave\nk_pyr_octave_mx\n"
    "k_gauss_v_strip\nk_gauss_h_strip\nk_gauss_band\nk_resize_linear2x_slide\nk_resize_nearest_half_v4\nk_extrema_w3\nk_extrema_dense\nk_localize_points\nk_orient_keypoints\nk_edge_response_windows\nk_level_gradients\nk_pack_rows\nk_edge_flags\nk_survivor_ranges\nk_orient_survivors\n"
    "k_extrema_pack\nk_harris_strip\nk_flag_count\nk_chunk_scan\nk_flag_scatter\nk_level_gradients\nk_sift_descriptors\nk_pack_offsets\nk_pack_copy\nk_count_totals";


// Launch on the context stream (no dynamic LDS).
#define LAUNCH(ctx, name, kern, grid, block, ...)                                 \
    do {                                                                          \
        {                                                                         \
            TimedScope ts_(ctx, name, (ctx)->launch_tag);                         \
            hipLaunchKernelGGL(kern, grid, block, 0, (ctx)->stream, __VA_ARGS__); \
        }                                                                         \
        HIPCHK(ctx, hipGetLastError());                                           \
    } while (0)


static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }


// Runs `body` with the context's launch stream temporarily replaced (LAUNCH uses ctx->stream).
struct StreamSwap {
    vslam_ctx* c;
    hipStream_t saved;
    StreamSwap(vslam_ctx* ctx, hipStream_t s) : c(ctx), saved(ctx->stream) { c->stream = s; }
    ~StreamSwap() { c->stream = saved; }
};

// Raises a kernel's dynamic shared memory ceiling to the most any launch of it may ask for.
static constexpr int kMaxDynLds = 150 * 1024;
// What one workgroup may really take (160 KB on MI355X), read from the device at the first context
// creation; only the fused band kernel plans against it (plan_octave), the other kernels stay below kMaxDynLds.
static int g_lds_limit = kMaxDynLds;
static int raise_dyn_lds(vslam_ctx* c, const void* fn, int limit = kMaxDynLds) {
    if (c->lds_raised.count(fn)) return VSLAM_OK;
    HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, limit));
    c->lds_raised.insert(fn);
    return VSLAM_OK;
}

static void* block_alloc(vslam_ctx* c, size_t bytes, size_t* cap) {
    int best = -1;
    for (int i = 0; i < (int)c->block_cache.size(); ++i) {
        const size_t sz = c->block_cache[i].first;
        if (sz >= bytes && sz <= 2 * bytes + (1 << 20) && (best < 0 || sz < c->block_cache[best].first)) best = i;
    }
    if (best >= 0) {
        void* p = c->block_cache[best].second;
        *cap = c->block_cache[best].first;
        c->block_cache.erase(c->block_cache.begin() + best);
        return p;
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        for (auto& b : c->block_cache) (void)hipFree(b.second);  // give the cache back and retry once
        c->block_cache.clear();
        if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    }
    *cap = bytes;
    return p;
}
static void block_release(vslam_ctx* c, void* p, size_t cap) {
    if (!p) return;
    if (c && c->block_cache.size() < 8 && cap <= ((size_t)1 << 30))
        c->block_cache.emplace_back(cap, p);
    else
        (void)hipFree(p);
}

static int ws_reserve(vslam_ctx* c, size_t bytes) {
    c->ws_off = 0;
    if (bytes <= c->ws_cap) return VSLAM_OK;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int i = 0; i < vslam_ctx::kAux; ++i)
        if (c->aux[i]) HIPCHK(c, hipStreamSynchronize(c->aux[i]));
    if (c->ws) (void)hipFree(c->ws);
    c->ws = nullptr;
    c->ws_cap = 0;
    const size_t want = align_up(bytes + bytes / 8, 1 << 20);
    HIPCHK(c, hipMalloc((void**)&c->ws, want));
    c->ws_cap = want;
    return VSLAM_OK;
}

template <typename T>
static T* ws_take(vslam_ctx* c, size_t count) {
    char* p = c->ws + c->ws_off;
    c->ws_off += align_up(count * sizeof(T), 256);
    return c->ws_off <= c->ws_cap ? (T*)p : nullptr;  // sized by ws_reserve; nullptr = sizing bug
}
static inline size_t ws_need(size_t bytes) { return align_up(bytes, 256); }

// One-time device tables (tap matrices) are allocated and copied with blocking calls: inside a stream capture that would
// invalidate the capture, so a call that still needs one says so instead (include/vslam.h: the warm-up call must run with
// the same parameters AND the same matrix-path setting as the captured one).
static bool stream_is_capturing(vslam_ctx* c) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cap) != hipSuccess) (void)hipGetLastError();
    return cap != hipStreamCaptureStatusNone;
}
#define NO_TABLE_IN_CAPTURE(ctx, what)                                                                                         \
    if (stream_is_capturing(ctx))                                                                                              \
    return fail(ctx, VSLAM_ERR_UNSUPPORTED, what ": its tap tables are not on the device yet and cannot be put there during a stream capture - " \
                                            "run one warm-up call with the same parameters and the same matrix-path setting first")

// Device copy of the (zero-trimmed) quantised taps of one GaussianBlur; *n_eff = trimmed width.
static int get_taps(vslam_ctx* c, int n, double sigma, const uint16_t** out, int* n_eff) {
    uint64_t sb;
    std::memcpy(&sb, &sigma, 8);
    auto key = std::make_pair(n, sb);
    std::vector<uint16_t> h;
    if (!gauss_taps_q8_trimmed(n, sigma, h)) return fail(c, VSLAM_ERR_INVALID, "invalid Gaussian kernel size");
    *n_eff = (int)h.size();
    auto it = c->taps.find(key);
    if (it == c->taps.end()) {
        NO_TABLE_IN_CAPTURE(c, "Gaussian blur");
        uint16_t* d = nullptr;
        HIPCHK(c, hipMalloc((void**)&d, sizeof(uint16_t) * h.size()));
        HIPCHK(c, hipMemcpy(d, h.data(), sizeof(uint16_t) * h.size(), hipMemcpyHostToDevice));
        it = c->taps.emplace(key, d).first;
    }
    *out = it->second;
    return VSLAM_OK;
}

static inline dim3 grid_rows(int cols, int rows, int frames = 1) { return dim3((cols + 255) / 256, rows, frames); }

// ---------------------------------------------------------------- enqueue helpers (device)

// ---- octave path selection -----------------------------------------------------------------
enum class OctPath { Tile0, Tile1, Band, Strip, Generic };

struct OctPlan {
    OctPath path = OctPath::Generic;
    int ks[6] = {};                 // OpenCV kernel widths (GaussianBlur's ksize)
    double sg[6] = {};
    int ke[6] = {};                 // zero-trimmed widths the fast kernels run with
    std::vector<uint16_t> taps[6];  // trimmed taps
    int sh = 0;                     // rows per horizontal strip workgroup
    int band_sh = 0, band_ri = 0, band_rm = 0, band_colsP = 0, band_pw = 0;  // fused band kernel (OctPath::Band)
    size_t band_lds = 0;
};

static bool taps_fit_u8(const OctPlan& pl) {
    for (int l = 0; l < 6; ++l)
        for (uint16_t v : pl.taps[l])
            if (v > 255) return false;
    return true;
}

template <class CFG>
static bool matches_cfg(const int ks[6]) {
    for (int l = 0; l < 6; ++l)
        if (ks[l] != CFG::n(l)) return false;
    return true;
}

// u16-pair columns of one staged row of the horizontal strip pass (left halo + ceil(cols/8) groups
// + right halo of the widest kernel), a multiple of 4
static int strip_pw(int cols, int nmax) { return ((cols + 7) / 2 + nmax / 2 + 12 + 3) & ~3; }

static OctPlan plan_octave(double sigma0, int o, int rows, int cols) {
    OctPlan pl;
    int nmax = 0;
    for (int l = 0; l < 6; ++l) {
        pl.sg[l] = sigma_at(sigma0, o, l);
        pl.ks[l] = gauss_ksize_u8(pl.sg[l]);
        if (!gauss_taps_q8_trimmed(pl.ks[l], pl.sg[l], pl.taps[l])) return pl;
        pl.ke[l] = (int)pl.taps[l].size();
        nmax = std::max(nmax, pl.ke[l]);
    }
    if (!taps_fit_u8(pl)) return pl;
    if (matches_cfg<PyrCfgOct0>(pl.ke)) {
        pl.path = OctPath::Tile0;
    } else if (matches_cfg<PyrCfgOct1>(pl.ke)) {
        pl.path = OctPath::Tile1;
    } else if (nmax <= STRIP_MAXN) {
        const int RM = (nmax / 2 + 3) & ~3;
        const size_t v_lds = (size_t)((((rows + 3) & ~3) + 2 * RM + 16) / 4) * STRIP_W * 4;
        pl.sh = cols <= 1024 ? 16 : cols <= 2048 ? 8 : cols <= 4096 ? 4 : 0;
        const size_t pw = (size_t)strip_pw(cols, nmax);
        if (pl.sh && v_lds <= 150 * 1024 && pl.sh * pw * 4 <= 150 * 1024) pl.path = OctPath::Strip;
        // The fused band kernel (both passes in one launch, row sums stay in LDS) when a band of 16 or 8 rows
        // with its vertical halo fits one CU's LDS and gives every thread at most two horizontal items.
        // OPT-IN (VSLAM_BAND_KERNEL=1), not the default: measured on MI355X, 256 x 1080p, same box, it takes
        // 2.9 ms per launch against 1.9 ms for the two strip kernels of the same octave (20.1 vs 18.05 ms per
        // step): the band's base rows fill the LDS, so one workgroup = 2 waves per SIMD runs per CU with three
        // barriers per level, and that costs more than the 15.5 MB per frame of scratch traffic it removes.
        static const bool use_band = [] {
            const char* e = VSLAM_DIAG_ENV("VSLAM_BAND_KERNEL");
            return e && e[0] == '1';
        }();
        if (use_band && pl.path == OctPath::Strip) {
            const int colsP = ((cols + 3) & ~3) + 4;  // dword pitch of a row quad (+4: spreads the quads over the banks)
            for (int sh : {16, 8}) {
                const size_t lds = ((size_t)((sh + 2 * RM) / 4 + 1) * colsP + (size_t)sh * pw) * 4;
                const int ri = ((cols + 7) / 8) * (sh / 2) <= 1024 ? 2 : 4;
                if (lds <= (size_t)g_lds_limit && ((cols + 7) / 8) * (sh / ri) <= 1024) {
                    pl.path = OctPath::Band;
                    pl.band_sh = sh, pl.band_ri = ri, pl.band_rm = RM, pl.band_colsP = colsP, pl.band_pw = (int)pw, pl.band_lds = lds;
                    break;
                }
            }
        }
    }
    return pl;
}

static int get_strip_taps(vslam_ctx* c, double sigma0, int o, const OctPlan& pl, const StripTaps** out) {
    uint64_t sb;
    std::memcpy(&sb, &sigma0, 8);
    auto key = std::make_pair(sb, o);
    auto it = c->strip_taps.find(key);
    if (it == c->strip_taps.end()) {
        NO_TABLE_IN_CAPTURE(c, "strip kernels");
        const uint16_t* tp[6];
        for (int l = 0; l < 6; ++l) tp[l] = pl.taps[l].data();
        std::vector<StripTaps> st(1);
        if (!strip_pack_taps(tp, pl.ke, st[0])) return fail(c, VSLAM_ERR_UNSUPPORTED, "strip kernels: taps out of range");
        StripTaps* d = nullptr;
        HIPCHK(c, hipMalloc((void**)&d, sizeof(StripTaps)));
        HIPCHK(c, hipMemcpy(d, st.data(), sizeof(StripTaps), hipMemcpyHostToDevice));
        it = c->strip_taps.emplace(key, d).first;
    }
    *out = it->second;
    return VSLAM_OK;
}

template <int SH, int RI>
static int launch_h_strip(vslam_ctx* c, const uint16_t* h, size_t hframe, uint8_t* oct, size_t pframe, int rows, int cols,
                          int pitch, int pw, int nf, const StripTaps* taps, uint8_t* next_base, size_t nframe, int nrows, int ncols,
                          int npitch) {
    const size_t lds = (size_t)SH * pw * 4;
    TRY(raise_dyn_lds(c, reinterpret_cast<const void*>(&k_gauss_h_strip<SH, RI>)));
    {
        TimedScope ts(c, "k_gauss_h_strip", c->launch_tag);
        hipLaunchKernelGGL((k_gauss_h_strip<SH, RI>), dim3(1, (rows + SH - 1) / SH, nf), dim3(256), lds, c->stream, h, hframe, oct,
                           pframe, rows, cols, pitch, pw, taps, next_base, nframe, nrows, ncols, npitch);
    }
    HIPCHK(c, hipGetLastError());
    return VSLAM_OK;
}

// Coarse octave: vertical strips (dot4) into the u16 scratch, then horizontal strips (dot2).
static int enqueue_strip_octave(vslam_ctx* c, double sigma0, int o, const OctPlan& pl, const uint8_t* base, size_t bframe,
                                uint8_t* oct, size_t pframe, uint16_t* h, int rows, int cols, int pitch, int nf,
                                uint8_t* next_base, size_t nframe, int nrows, int ncols, int npitch) {
    const StripTaps* taps;
    TRY(get_strip_taps(c, sigma0, o, pl, &taps));
    int nmax = 0;
    for (int l = 0; l < 6; ++l) nmax = std::max(nmax, pl.ke[l]);
    const int RM = (nmax / 2 + 3) & ~3;
    const int rhq = (((rows + 3) & ~3) + 2 * RM + 16) / 4;
    const size_t v_lds = (size_t)rhq * STRIP_W * 4;
    const size_t P = (size_t)rows * pitch;
    TRY(raise_dyn_lds(c, reinterpret_cast<const void*>(&k_gauss_v_strip)));
    {
        TimedScope ts(c, "k_gauss_v_strip", o);
        // small batches: split the six levels over workgroups until the launch has >= 256 of them
        const int strips = (cols + STRIP_W - 1) / STRIP_W;
        const int want = (256 + strips * nf - 1) / (strips * nf);
        const int lsplit = want >= 6 ? 6 : want >= 3 ? 3 : want >= 2 ? 2 : 1;
        hipLaunchKernelGGL(k_gauss_v_strip, dim3(strips, lsplit, nf), dim3(256), v_lds, c->stream, base, bframe, h, 6 * P, rows, cols,
                           pitch, RM, rhq, taps);
    }
    HIPCHK(c, hipGetLastError());
    const int pw = strip_pw(cols, nmax);
    c->launch_tag = o;
    // small batches: shorter row strips, more workgroups
    int sh = pl.sh;
    while (sh > 4 && (long)((rows + sh - 1) / sh) * nf < 256) sh >>= 1;
    // ... and, when even that leaves most threads without an item, one row per item
    const bool fine = sh == 4 && (long)((rows + 3) / 4) * nf < 256 && ((cols + 7) / 8) * 4 <= 512;
    if (fine) return launch_h_strip<4, 1>(c, h, 6 * P, oct, pframe, rows, cols, pitch, pw, nf, taps, next_base, nframe, nrows, ncols, npitch);
    if (sh == 16) {
        // rows per item (round 5): the items of a workgroup should fill whole waves.  960 columns x 16 rows are 480 items of 8
        // columns x 4 rows = 7.5 waves (every eighth wave-instruction wasted: the kernel runs AT its VALU issue time), but 960
        // items of 2 rows = 15 waves; 480 columns need 1 row per item.  Fewer rows per item amortise the scalar tap loads over
        // fewer dots, so the smaller item must be at least 2 % fuller to be chosen.  VSLAM_STRIP_RI forces a value (A/B runs).
        const int ncg = (cols + 7) / 8;
        auto waste = [&](int ri) {
            const long items = (long)ncg * (16 / ri);
            if (items > (ri == 4 ? 512 : 1024)) return 1e9;
            return (double)((items + 63) / 64 * 64 - items) / (double)items;
        };
        static const int force_ri = [] {
            const char* e = VSLAM_DIAG_ENV("VSLAM_STRIP_RI");
            return e ? std::atoi(e) : 0;
        }();
        int ri = 4;
        for (int r : {2, 1})
            if (waste(r) + 0.02 < waste(ri)) ri = r;
        if ((force_ri == 1 || force_ri == 2 || force_ri == 4) && waste(force_ri) < 1e8) ri = force_ri;
        if (ri == 2) return launch_h_strip<16, 2>(c, h, 6 * P, oct, pframe, rows, cols, pitch, pw, nf, taps, next_base, nframe, nrows, ncols, npitch);
        if (ri == 1) return launch_h_strip<16, 1>(c, h, 6 * P, oct, pframe, rows, cols, pitch, pw, nf, taps, next_base, nframe, nrows, ncols, npitch);
    }
    switch (sh) {
        case 16: return launch_h_strip<16, 4>(c, h, 6 * P, oct, pframe, rows, cols, pitch, pw, nf, taps, next_base, nframe, nrows, ncols, npitch);
        case 8: return launch_h_strip<8, 4>(c, h, 6 * P, oct, pframe, rows, cols, pitch, pw, nf, taps, next_base, nframe, nrows, ncols, npitch);
        default: return launch_h_strip<4, 4>(c, h, 6 * P, oct, pframe, rows, cols, pitch, pw, nf, taps, next_base, nframe, nrows, ncols, npitch);
    }
}


// Coarse octave, fused: one launch, the row sums stay in LDS (k_gauss_band).
template <int SH, int RI>
static int launch_band(vslam_ctx* c, const OctPlan& pl, const StripTaps* taps, const uint8_t* base, size_t bframe, uint8_t* oct, size_t pframe,
                       int rows, int cols, int pitch, int nf, uint8_t* next_base, size_t nframe, int nrows, int ncols, int npitch) {
    TRY(raise_dyn_lds(c, reinterpret_cast<const void*>(&k_gauss_band<SH, RI>), g_lds_limit));  // no static LDS in this kernel
    {
        TimedScope ts(c, "k_gauss_band");
        hipLaunchKernelGGL((k_gauss_band<SH, RI>), dim3(1, (rows + SH - 1) / SH, nf), dim3(512), pl.band_lds, c->stream, base, bframe, oct, pframe, rows,
                           cols, pitch, pl.band_rm, pl.band_colsP, pl.band_pw, taps, next_base, nframe, nrows, ncols, npitch);
    }
    HIPCHK(c, hipGetLastError());
    return VSLAM_OK;
}
static int enqueue_band_octave(vslam_ctx* c, double sigma0, int o, const OctPlan& pl, const uint8_t* base, size_t bframe, uint8_t* oct,
                               size_t pframe, int rows, int cols, int pitch, int nf, uint8_t* next_base, size_t nframe, int nrows, int ncols,
                               int npitch) {
    const StripTaps* taps;
    TRY(get_strip_taps(c, sigma0, o, pl, &taps));
#define VSLAM_BAND(SH, RI) launch_band<SH, RI>(c, pl, taps, base, bframe, oct, pframe, rows, cols, pitch, nf, next_base, nframe, nrows, ncols, npitch)
    if (pl.band_sh == 16) return pl.band_ri == 2 ? VSLAM_BAND(16, 2) : VSLAM_BAND(16, 4);
    return pl.band_ri == 2 ? VSLAM_BAND(8, 2) : VSLAM_BAND(8, 4);
#undef VSLAM_BAND
}

// GaussianBlur CV_8U on nf dense images; h = u16 scratch of nf*rows*cols elements.
static int enqueue_blur(vslam_ctx* c, const uint8_t* src, size_t sstep, size_t sframe, uint8_t* dst, size_t dstep,
                        size_t dframe, uint16_t* h, int rows, int cols, int nf, int n, double sigma) {
    const uint16_t* taps;
    int rc = get_taps(c, n, sigma, &taps, &n);  // n becomes the trimmed width
    if (rc) return rc;
    const size_t P = (size_t)rows * cols;
    LAUNCH(c, "k_blur_h_generic", k_blur_h_generic, grid_rows(cols, rows, nf), dim3(256), src, sstep, sframe, h, P,
           rows, cols, taps, n);
    LAUNCH(c, "k_blur_v_generic", k_blur_v_generic, grid_rows(cols, rows, nf), dim3(256), h, P, dst, dstep, dframe,
           rows, cols, taps, n);
    return VSLAM_OK;
}

// count -> scan -> scatter over the flag entries of nf frames (kernels_compact.hip.h).
// chunk_ws: scratch of nf * chunks u32.
template <class E>
static int enqueue_compaction(vslam_ctx* c, const E& ent, size_t entries, int nf, unsigned int* chunk_ws, unsigned int cap,
                              unsigned int* counts, int append) {
    const int nchunks = (int)((entries + CMP_CHUNK - 1) / CMP_CHUNK);
    if (nchunks == 0) {
        if (!append) HIPCHK(c, hipMemsetAsync(counts, 0, sizeof(unsigned int) * (size_t)nf, c->stream));
        return VSLAM_OK;
    }
    LAUNCH(c, "k_flag_count", k_flag_count<E>, dim3(nchunks, 1, nf), dim3(256), ent, chunk_ws, nchunks);
    LAUNCH(c, "k_chunk_scan", k_chunk_scan, dim3(nf), dim3(256), chunk_ws, nchunks, counts, append);
    LAUNCH(c, "k_flag_scatter", k_flag_scatter<E>, dim3(nchunks, 1, nf), dim3(256), ent, chunk_ws, nchunks, cap);
    return VSLAM_OK;
}
static inline size_t compaction_ws_elems(size_t entries, int nf) { return (size_t)nf * ((entries + CMP_CHUNK - 1) / CMP_CHUNK) + 64; }

// The table of the localization's quadratic term, filled by the same device function that the
// kernels fall back to (so a lookup cannot differ from the computation).
static int ensure_loc_lut(vslam_ctx* c) {
    if (c->loc_lut) return VSLAM_OK;
    constexpr int n = LOC_LUT_N * LOC_LUT_N * LOC_LUT_N;
    float* lut = nullptr;
    HIPCHK(c, hipMalloc((void**)&lut, sizeof(float) * n));
    hipLaunchKernelGGL(k_build_localization_lut, dim3((n + 255) / 256), dim3(256), 0, c->stream, lut);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {  // never leave a table that was not built
        (void)hipFree(lut);
        return fail(c, VSLAM_ERR_HIP, std::string("k_build_localization_lut: ") + hipGetErrorString(e));
    }
    c->loc_lut = lut;
    return VSLAM_OK;
}

static void fill_geom(const vslam_ctx* c, const vslam_params& p, const vslam_batch_layout& L, ExtGeom& g) {
    std::memset(&g, 0, sizeof(g));
    g.loc_lut = c->loc_lut;
    g.n_oct = L.n_octaves;
    g.window = p.extrema_window;
    g.pad = (p.extrema_window - 1) / 2;
    g.min_contrast = p.min_contrast;
    g.localize = p.localize;
    for (int o = 0; o < L.n_octaves; ++o) {
        g.rows[o] = L.rows[o];
        g.cols[o] = L.cols[o];
        g.pitch[o] = L.pitch[o];
        g.lat_rows[o] = L.lat_rows[o];
        g.lat_cols[o] = L.lat_cols[o];
        g.wpr[o] = L.lat_words[o];
        g.oct_off[o] = L.octave_offset[o];
        g.bits_off[o] = L.bits_offset[o];
    }
}

struct DogScratch {
    uint8_t* bases = nullptr;  // nf * sum_P
    size_t bases_frame = 0;
    size_t base_off[VSLAM_MAX_OCTAVES] = {};
    uint16_t* h = nullptr;  // nf * P0 u16
    unsigned long long* lflags = nullptr;
    unsigned int* cws = nullptr;  // compaction chunk totals / offsets
    unsigned int* pbegin = nullptr;  // localize mode: per-frame list length before the current octave
    // matrix path with the lattice scan fused into the octave kernel: one byte per lattice site of the octaves that kernel covers
    uint8_t* sitemap = nullptr;
    size_t site_frame = 0, site_off[VSLAM_MAX_OCTAVES] = {};
    int site_pitch[VSLAM_MAX_OCTAVES] = {};
    // ... and the two DoG columns on either side of every seam between its 128-column strips (vslam_mx.h: MxScan::colmap)
    uint8_t* colmap = nullptr;
    size_t col_frame = 0, col_off[VSLAM_MAX_OCTAVES] = {};
};

static inline int site_pitch_of(int lat_cols) { return (lat_cols + 63) & ~63; }
static size_t site_frame_bytes(const vslam_batch_layout& L) {
    size_t b = 0;
    for (int o = 0; o < L.n_octaves; ++o) b += (size_t)L.lat_rows[o] * site_pitch_of(L.lat_cols[o]);
    return b;
}
static size_t col_octave_bytes(const vslam_batch_layout& L, int o) { return align_up((size_t)5 * mx_seams(L.cols[o]) * L.rows[o] * 2, 16); }
static size_t col_frame_bytes(const vslam_batch_layout& L) {
    size_t b = 0;
    for (int o = 0; o < L.n_octaves; ++o) b += col_octave_bytes(L, o);
    return b;
}

// u16 scratch elements per frame: 6 row-sum images for a strip octave, 1 for a generic octave,
// none for the LDS-tiled octaves.
static size_t dog_h_elems(const vslam_batch_layout& L, double sigma0) {
    size_t m = 0;
    for (int o = 0; o < L.n_octaves; ++o) {
        const size_t P = (size_t)L.rows[o] * L.pitch[o];
        const OctPath path = plan_octave(sigma0, o, L.rows[o], L.cols[o]).path;
        if (path == OctPath::Strip) m = std::max(m, 6 * P);
        if (path == OctPath::Generic) m = std::max(m, P);
    }
    return m;
}

static size_t dog_scratch_bytes(const vslam_batch_layout& L, double sigma0, int nf, bool sitemap = false) {
    size_t sum_p = 0;
    for (int o = 0; o < L.n_octaves; ++o) sum_p += (size_t)L.rows[o] * L.pitch[o];
    return (sitemap ? ws_need((size_t)nf * site_frame_bytes(L)) + ws_need((size_t)nf * col_frame_bytes(L) + 16) : 0) + ws_need((size_t)nf * sum_p) + ws_need((size_t)nf * dog_h_elems(L, sigma0) * 2 + 256) +
           ws_need((size_t)nf * L.bits_frame_words * 8) + ws_need(4 * compaction_ws_elems(L.bits_frame_words, nf)) + ws_need(4 * (size_t)nf);
}

static int dog_scratch_take(vslam_ctx* c, const vslam_batch_layout& L, double sigma0, int nf, DogScratch& s, bool sitemap = false) {
    if (sitemap) {
        size_t off = 0;
        for (int o = 0; o < L.n_octaves; ++o) {
            s.site_off[o] = off;
            s.site_pitch[o] = site_pitch_of(L.lat_cols[o]);
            off += (size_t)L.lat_rows[o] * s.site_pitch[o];
        }
        s.site_frame = off;
        s.sitemap = ws_take<uint8_t>(c, (size_t)nf * off);
        size_t coff = 0;
        for (int o = 0; o < L.n_octaves; ++o) {
            s.col_off[o] = coff;
            coff += col_octave_bytes(L, o);
        }
        s.col_frame = coff;
        s.colmap = ws_take<uint8_t>(c, (size_t)nf * coff + 16);
        if (!s.sitemap || !s.colmap) return fail(c, VSLAM_ERR_NOMEM, "workspace sizing error (site map)");
    }
    size_t sum_p = 0;
    for (int o = 0; o < L.n_octaves; ++o) {
        s.base_off[o] = sum_p;  // octave bases keep the pitched rows of the pyramid planes
        sum_p += (size_t)L.rows[o] * L.pitch[o];
    }
    s.bases_frame = sum_p;
    s.bases = ws_take<uint8_t>(c, (size_t)nf * sum_p);
    s.h = ws_take<uint16_t>(c, (size_t)nf * dog_h_elems(L, sigma0) + 128);
    s.lflags = ws_take<unsigned long long>(c, (size_t)nf * L.bits_frame_words);
    s.cws = ws_take<unsigned int>(c, compaction_ws_elems(L.bits_frame_words, nf));
    s.pbegin = ws_take<unsigned int>(c, nf);
    if (!s.bases || !s.h || !s.lflags || !s.cws || !s.pbegin) return fail(c, VSLAM_ERR_NOMEM, "workspace sizing error (dog)");
    return VSLAM_OK;
}


// Fused LDS-tiled octave (kernels_pyramid.hip.h); the plan has already matched CFG's widths.
template <class CFG>
static int enqueue_pyr_octave(vslam_ctx* c, double sigma0, int o, const OctPlan& pl, const uint8_t* base, size_t bframe,
                              uint8_t* oct_out, size_t pframe, int rows, int cols, int pitch, int nf, uint8_t* next_base,
                              size_t nframe, int nrows, int ncols, int npitch) {
    uint64_t sb;
    std::memcpy(&sb, &sigma0, 8);
    auto key = std::make_pair(sb, o);
    auto it = c->tile_taps.find(key);
    if (it == c->tile_taps.end()) {
        NO_TABLE_IN_CAPTURE(c, "octave kernel");
        const uint16_t* tp[6];
        for (int l = 0; l < 6; ++l) tp[l] = pl.taps[l].data();
        std::vector<PyrTaps<CFG>> host(1);
        pyr_pack_taps<CFG>(tp, host[0]);
        void* d = nullptr;
        HIPCHK(c, hipMalloc(&d, sizeof(PyrTaps<CFG>)));
        HIPCHK(c, hipMemcpy(d, host.data(), sizeof(PyrTaps<CFG>), hipMemcpyHostToDevice));
        it = c->tile_taps.emplace(key, d).first;
    }
    TRY(raise_dyn_lds(c, reinterpret_cast<const void*>(&k_pyr_octave<CFG>)));  // once per kernel (both tile shapes share the taps)
    const PyrTaps<CFG>* taps = static_cast<const PyrTaps<CFG>*>(it->second);
    const dim3 grid((cols + CFG::TW - 1) / CFG::TW, (rows + CFG::TH - 1) / CFG::TH, nf);
    {
        TimedScope ts(c, "k_pyr_octave", o);
        hipLaunchKernelGGL(k_pyr_octave<CFG>, grid, dim3(CFG::NT), CFG::LDS_BYTES, c->stream, base, bframe, oct_out, pframe, rows,
                           cols, pitch, taps, next_base, nframe, nrows, ncols, npitch);
    }
    HIPCHK(c, hipGetLastError());
    return VSLAM_OK;
}

// The same octave through the matrix-core kernel (kernels_pyramid_mx.hip.h, vslam_mx.hip); `cfg` from mx_config_for.
static int enqueue_pyr_octave_mx(vslam_ctx* c, int cfg, double sigma0, int o, const OctPlan& pl, const uint8_t* base, size_t bframe,
                                 uint8_t* oct_out, size_t pframe, int rows, int cols, int pitch, int nf, uint8_t* next_base, size_t nframe,
                                 int nrows, int ncols, int npitch, const MxScan* scan, int up2_step = 0) {
    uint64_t sb;
    std::memcpy(&sb, &sigma0, 8);
    auto key = std::make_pair(sb, o);
    auto it = c->mx_taps.find(key);
    if (it == c->mx_taps.end()) {
        NO_TABLE_IN_CAPTURE(c, "matrix-core octave kernel");
        const uint16_t* tp[6];
        for (int l = 0; l < 6; ++l) tp[l] = pl.taps[l].data();
        std::vector<char> host(mx_taps_bytes(cfg));
        if (!mx_pack(cfg, tp, host.data())) return fail(c, VSLAM_ERR_UNSUPPORTED, "matrix-core octave kernel: a tap exceeds 127");
        HIPCHK(c, mx_prepare(cfg));
        void* d = nullptr;
        HIPCHK(c, hipMalloc(&d, host.size()));
        HIPCHK(c, hipMemcpy(d, host.data(), host.size(), hipMemcpyHostToDevice));
        it = c->mx_taps.emplace(key, d).first;
    }
    hipError_t e;
    {
        TimedScope ts(c, "k_pyr_octave_mx", o);
        e = mx_launch(cfg, c->stream, it->second, base, bframe, oct_out, pframe, rows, cols, pitch, nf, next_base, nframe, nrows, ncols, npitch, scan, up2_step);
    }
    HIPCHK(c, e);
    return VSLAM_OK;
}

// The octave whose kernels the held-back side work of a batch waits for (enqueue_dog): the last
// LDS-tiled one for batches of 32 frames or more, -1 (no gate) otherwise.
// vslam_ctx_follow: the point of a batch call behind which a second context's batch may start (its heavy octave-0
// kernels then run beside this call's remaining, shorter kernels instead of beside its own octave 0).
static int follow_octave() {
    static const int o = [] {
        const char* e = VSLAM_DIAG_ENV("VSLAM_FOLLOW_OCTAVE");
        return e ? atoi(e) : 0;
    }();
    return o;
}
static int mark_phase(vslam_ctx* c) {
    if (!c->ev_phase) HIPCHK(c, hipEventCreateWithFlags(&c->ev_phase, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_phase, c->stream));
    c->phase_marked = true;
    return VSLAM_OK;
}

// Diagnostics build only: VSLAM_DIAG_SKIP_SCAN="<octave mask>,<n>" leaves the plain lattice scan (k_extrema_w3) of the octaves in
// the mask out of every batch call of a context after its n-th - a TIMING knock-out (profiles/r06_scan_knockout.txt: the upper
// bound of what folding that scan into the octave kernel could win).  The list kernels then compact the flag words the earlier
// calls left, so their work is unchanged when the frames are.  Never in the shipped library.
static bool diag_skip_scan(const vslam_ctx* c, int octave) {
    static const std::pair<unsigned, int> cfg = [] {
        const char* e = VSLAM_DIAG_ENV("VSLAM_DIAG_SKIP_SCAN");
        unsigned mask = 0;
        int after = 0;
        if (e) {
            mask = (unsigned)std::strtoul(e, nullptr, 0);
            if (const char* comma = std::strchr(e, ',')) after = std::atoi(comma + 1);
        }
        return std::make_pair(mask, after);
    }();
    return cfg.first && ((cfg.first >> octave) & 1u) && c->batch_calls > cfg.second;
}

static int dog_side_gate(const vslam_params& p, const vslam_batch_layout& L, int nf) {
    int gate = -1;
    if (nf >= 32)
        for (int o = 0; o < L.n_octaves; ++o) {
            const OctPath path = plan_octave(p.sigma0, o, L.rows[o], L.cols[o]).path;
            if (path == OctPath::Tile0 || path == OctPath::Tile1) gate = o;
        }
    return gate;
}

// Geometry of the dense 3x3x3 scan (kernels_extrema_dense.hip.h) for octave o of nf frames.
static DenseGeom dense_geom(const vslam_batch_layout& L, int o, int min_contrast, int nf) {
    DenseGeom g;
    g.rows = L.rows[o], g.cols = L.cols[o], g.pitch = L.pitch[o];
    g.wpr = (g.cols + 63) / 64;
    g.min_contrast = min_contrast;
    g.P = (unsigned int)((size_t)g.rows * g.pitch);
    g.dog_off = (unsigned int)(L.octave_offset[o] + (size_t)VSLAM_NUM_LEVELS * g.P);
    // a lane walks g.seg rows: long segments amortise the two halo rows, short ones give a small launch
    // enough waves to hide the loads (about 8 per SIMD)
    const long waves_per_rowseg = 4L * (((g.cols + 3) / 4 + 255) / 256) * nf;
    const long segs_wanted = std::max<long>(1, 8192 / waves_per_rowseg);
    g.seg = (int)std::min<long>(XD_SEG_MAX, std::max<long>(4, (g.rows + segs_wanted - 1) / segs_wanted));
    return g;
}

// createPyramid (GaussPyramid.cpp:106-131) + initialKeypointDetection (Diff_of_Gauss.cpp:254)
// for nf frames; pyr/bits/points are per-frame blocks with the given strides.
static int enqueue_dog(vslam_ctx* c, const vslam_params& p, const vslam_batch_layout& L, const uint8_t* frames,
                       size_t fstep, size_t fframe, int nf, uint8_t* pyr, size_t pframe, DogScratch& s,
                       unsigned long long* bits, bool do_extrema, vslam_point* points, unsigned int* counts,
                       hipStream_t side = nullptr, const std::function<int(int)>& after_list = nullptr,
                       const std::function<int(int)>& after_octave = nullptr, hipStream_t up = nullptr, bool later_chunk = false,
                       bool bases_are_scratch = false) {
    // `side`: stream for the extrema scans and the list compaction (they only read what the
    // octave kernels wrote); ordered after the octave kernels by events.  nullptr = same stream.
    //
    // Where the side work runs decides how much of the VALU-issue-bound octave kernels it costs
    // (gate = dog_side_gate(): the last LDS-tiled octave, -1 = none; 256 x 1080p, same box):
    //  * the Harris chain (VALU-heavy) always waits for that octave's kernel - the caller enqueues it
    //    from after_octave(gate), behind ev_oct[gate] - and runs beside the coarse-octave strip
    //    kernels, which are short of waves: k_pyr_octave 7.46 -> 6.06 ms per launch, +1.4 % frames/s;
    //  * the plain extrema scan (HBM-heavy, ~90 VALU instructions per thread) starts as soon as its
    //    octave is written, i.e. octave 0's scan runs beside octave 1's kernel: +2..3 % frames/s
    //    over holding it back as well (the tail after the tiled octaves is as VALU-bound as they are,
    //    so the scan's memory time is what gets hidden);
    //  * the scan with FeaturePointLocalization inside (params.localize, ~8x the instructions) is
    //    held back like the Harris chain: +0.5 % in the localize / orient / describe modes.
    // Small batches keep the eager order: there the chain's latency matters, not the issue slots.
    const int gate = (side && p.localize) ? dog_side_gate(p, L, nf) : -1;
    ExtGeom g;
    if (p.localize) TRY(ensure_loc_lut(c));
    fill_geom(c, p, L, g);
    // Large batches go through the upsample and octave 0 in two halves: the second half is upsampled
    // on the side stream (idle until octave 0 is done) while the first half's octave-0 kernel runs, so
    // only half of the bandwidth-bound upsample is exposed in front of the VALU-bound octave kernels.
    // The second half goes to a stream of its own (`up`): on `side` it would queue behind the previous
    // chunk's whole list chain.  It overwrites bases the previous chunk's octave kernels (main stream)
    // read, and nothing else orders it behind them - with pyramid-only outputs there is not even a scan
    // waiting on ev_oct - so it waits for ev_chunk, recorded at the end of every chunk's main-stream work.
    // (quarters and eighths were measured again in round 3 with the side upsample at normal priority: 21.41 /
    // 21.44 ms against 21.29 for halves in the same configuration - no gain)
    // Matrix path, batched entry (the octave bases are scratch nobody reads afterwards): octave 0's kernel forms its base from
    // the frame while it stages a tile (kernels_pyramid_mx.hip.h: mx_stage_tile_up2) - no upsample kernel, no base in HBM.
    const bool up2_fused = c->mx && bases_are_scratch && L.n_octaves > 0 && L.rows[0] == 2 * p.rows && L.cols[0] == 2 * p.cols && fstep <= 0x7fffffff &&
                           [&] {
                               const OctPlan pl0 = plan_octave(p.sigma0, 0, L.rows[0], L.cols[0]);
                               return pl0.path != OctPath::Generic && mx_up2_supported(mx_config_for(pl0.ke));
                           }();
    const int nf_a = (!up2_fused && side && up && nf >= 64) ? nf / 2 : nf;
    if (!up2_fused)
        LAUNCH(c, "k_resize_linear2x_slide", k_resize_linear2x_slide, dim3(((p.cols + 3) / 4 + 255) / 256, (p.rows + 15) / 16, nf_a), dim3(256),
               frames, fstep, fframe, s.bases + s.base_off[0], s.bases_frame, L.pitch[0], p.rows, p.cols, 16);
    if (nf_a < nf) {
        if (later_chunk) HIPCHK(c, hipStreamWaitEvent(up, c->ev_chunk, 0));
        StreamSwap sw(c, up);
        LAUNCH(c, "k_resize_linear2x_slide", k_resize_linear2x_slide, dim3(((p.cols + 3) / 4 + 255) / 256, (p.rows + 15) / 16, nf - nf_a), dim3(256),
               frames + (size_t)nf_a * fframe, fstep, fframe, s.bases + s.base_off[0] + (size_t)nf_a * s.bases_frame, s.bases_frame, L.pitch[0],
               p.rows, p.cols, 16);
        HIPCHK(c, hipEventRecord(c->ev_up2, c->stream));
    }
    bool fused[VSLAM_MAX_OCTAVES] = {};
    int fused_rows[VSLAM_MAX_OCTAVES] = {};  // rows of a wave's strip in the octave kernel that ran the fused scan (32, or 16 for the 16 x 16 x 64 form)
    // a later chunk reuses the site / seam maps: its octave kernels (main stream) must not overwrite them while the previous
    // chunk's k_extrema_pack launches (side stream, low priority, possibly on a slow hardware queue) are still reading
    if (later_chunk && c->pack_pending) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_pack, 0));
        c->pack_pending = false;
    }
    struct TagReset {
        vslam_ctx* c;
        ~TagReset() { c->launch_tag = -1; }
    } tag_reset{c};
    for (int o = 0; o < L.n_octaves; ++o) {
        c->launch_tag = o;  // the timing hook's "name@o" (TimedScope) for every launch of this octave
        const int rows = L.rows[o], cols = L.cols[o], pitch = L.pitch[o];
        const size_t P = (size_t)rows * pitch;
        const uint8_t* base = s.bases + s.base_off[o];
        uint8_t* oct = pyr + L.octave_offset[o];
        const OctPlan pl = plan_octave(p.sigma0, o, rows, cols);
        // the fast octave kernels also emit the next octave's base (Gaussian[3] decimated 2:1)
        const bool has_next = o + 1 < L.n_octaves;
        const bool fuse_next = has_next && pl.path != OctPath::Generic;
        uint8_t* nb = fuse_next ? s.bases + s.base_off[o + 1] : nullptr;
        const int nr = has_next ? L.rows[o + 1] : 0, nc = has_next ? L.cols[o + 1] : 0, np = has_next ? L.pitch[o + 1] : 0;
        // tile shape: the wide tile (256 x 32) when it needs no more tile area than the tall one (128 x 64).
        // A 384 x 32 tile (1920 = 5 x 384) on 384-thread workgroups was measured in round 3: six waves per
        // workgroup sit 2-2-1-1 on the four SIMDs and meet at every barrier: 21.3 vs 18.3 ms per step.
        static const int force_shape = [] {
            const char* e = VSLAM_DIAG_ENV("VSLAM_TILE_SHAPE");  // A/B runs: 0 = 128 x 64 everywhere, 1 = 256 x 32 everywhere
            return e ? std::atoi(e) : -1;
        }();
        const int shape = force_shape >= 0 ? (force_shape ? 1 : 0)
                                           : ((long)((cols + 255) / 256) * ((rows + 31) / 32) <= (long)((cols + 127) / 128) * ((rows + 63) / 64) ? 1 : 0);
        // matrix path: the plain lattice scan (window 3, candidates + contrast list) runs inside the octave kernel
        // while the DoG rows are in LDS (kernels_pyramid_mx.hip.h); k_extrema_pack then replaces k_extrema_w3
        MxScan scan{};
        const bool fused_scan = c->mx && s.sitemap && do_extrema && !p.localize && !p.extrema_dense && p.extrema_window == 3 && L.lat_rows[o] > 0 &&
                                L.lat_cols[o] > 0 && pl.path != OctPath::Generic && mx_scan_supported(mx_config_for(pl.ke));
        if (fused_scan)
            scan = MxScan{s.sitemap + s.site_off[o], s.site_frame, L.lat_rows[o], L.lat_cols[o], s.site_pitch[o], p.min_contrast,
                          s.colmap + s.col_off[o], s.col_frame, mx_seams(cols)};
        fused[o] = fused_scan;
        fused_rows[o] = fused_scan ? mx_strip_rows(mx_config_for(pl.ke)) : 0;
        // frames [f_lo, f_lo + n) of this octave through the LDS-tiled kernel
        auto tiled = [&](int f_lo, int n) -> int {
            const uint8_t* b = base + (size_t)f_lo * s.bases_frame;
            uint8_t* oc = oct + (size_t)f_lo * pframe;
            uint8_t* nbh = nb ? nb + (size_t)f_lo * s.bases_frame : nullptr;
            if (c->mx)
                if (const int cfg = mx_config_for(pl.ke))
                {
                    MxScan sc = scan;
                    sc.sitemap += (size_t)f_lo * s.site_frame;
                    sc.colmap += (size_t)f_lo * s.col_frame;
                    if (o == 0 && up2_fused)
                        return enqueue_pyr_octave_mx(c, cfg, p.sigma0, o, pl, frames + (size_t)f_lo * fframe, fframe, oc, pframe, rows, cols, pitch, n, nbh,
                                                     s.bases_frame, nr, nc, np, fused_scan ? &sc : nullptr, (int)fstep);
                    return enqueue_pyr_octave_mx(c, cfg, p.sigma0, o, pl, b, s.bases_frame, oc, pframe, rows, cols, pitch, n, nbh, s.bases_frame, nr, nc, np,
                                                 fused_scan ? &sc : nullptr);
                }
#define VSLAM_TILED(CFG) enqueue_pyr_octave<CFG>(c, p.sigma0, o, pl, b, s.bases_frame, oc, pframe, rows, cols, pitch, n, nbh, s.bases_frame, nr, nc, np)
            if (pl.path == OctPath::Tile0) return shape == 1 ? VSLAM_TILED(PyrCfgOct0W) : VSLAM_TILED(PyrCfgOct0);
            return shape == 1 ? VSLAM_TILED(PyrCfgOct1W) : VSLAM_TILED(PyrCfgOct1);
#undef VSLAM_TILED
        };
        // opt-in matrix path: also the octaves the default path runs through the strip kernels (no u16 scratch round trip)
        const bool is_tiled = pl.path == OctPath::Tile0 || pl.path == OctPath::Tile1 || (c->mx && pl.path != OctPath::Generic && mx_config_for(pl.ke) != 0);
        if (o == 0 && nf_a < nf && !is_tiled) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_up2, 0));  // octave 0 needs every base
        if (is_tiled && o == 0 && nf_a < nf) {
            TRY(tiled(0, nf_a));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_up2, 0));
            TRY(tiled(nf_a, nf - nf_a));
        } else if (is_tiled)
            TRY(tiled(0, nf));
        else if (pl.path == OctPath::Band)
            TRY(enqueue_band_octave(c, p.sigma0, o, pl, base, s.bases_frame, oct, pframe, rows, cols, pitch, nf, nb, s.bases_frame, nr, nc, np));
        else if (pl.path == OctPath::Strip)
            TRY(enqueue_strip_octave(c, p.sigma0, o, pl, base, s.bases_frame, oct, pframe, s.h, rows, cols, pitch, nf, nb, s.bases_frame, nr, nc, np));
        else {
            for (int l = 0; l < VSLAM_NUM_LEVELS; ++l)
                TRY(enqueue_blur(c, base, (size_t)pitch, s.bases_frame, oct + (size_t)l * P, (size_t)pitch, pframe, s.h, rows,
                                 cols, nf, pl.ks[l], pl.sg[l]));
            LAUNCH(c, "k_dog5", k_dog5, dim3((unsigned)((P + 255) / 256), 1, nf), dim3(256), oct,
                   oct + (size_t)VSLAM_NUM_LEVELS * P, P, pframe);
        }
        if (has_next && !fuse_next)
            LAUNCH(c, "k_resize_nearest_half_v4", k_resize_nearest_half_v4, dim3((L.cols[o + 1] / 4 + 256) / 256, L.rows[o + 1], nf),
                   dim3(256), oct + (size_t)3 * P, pframe, pitch, s.bases + s.base_off[o + 1], s.bases_frame, L.pitch[o + 1], rows,
                   L.rows[o + 1], L.cols[o + 1]);
        if (side) HIPCHK(c, hipEventRecord(c->ev_oct[o], c->stream));
        if (o == follow_octave() || (o == L.n_octaves - 1 && o < follow_octave())) TRY(mark_phase(c));
        if (after_octave) TRY(after_octave(o));  // octave o's kernels are enqueued and ev_oct[o] marks their end
        if (o < gate) continue;  // scan + compaction of this octave are enqueued behind octave `gate`
        const int o_done = o;    // the octave whose kernels were enqueued last
        for (int oo = (o_done == gate ? 0 : o_done); oo <= o_done; ++oo) {
        const int o = oo;
        if (do_extrema && L.lat_rows[o] > 0 && L.lat_cols[o] > 0) {
            hipStream_t es = c->stream;
            if (side) {
                HIPCHK(c, hipStreamWaitEvent(side, c->ev_oct[o_done], 0));
                es = side;
            }
            if (p.extrema_dense) {
                // extension: the dense 3x3x3 test on every pixel (kernels_extrema_dense.hip.h); the layout's
                // lattice of this mode is the image itself
                const DenseGeom dg = dense_geom(L, o, p.min_contrast, nf);
                hipLaunchKernelGGL(k_extrema_dense, dim3(((dg.cols + 3) / 4 + 255) / 256, (dg.rows + dg.seg - 1) / dg.seg, nf), dim3(256), 0, es,
                                   pyr, pframe, dg, bits ? bits + L.bits_offset[o] : nullptr, s.lflags + L.bits_offset[o], L.bits_frame_words);
            } else if (fused[o]) {
                const MxScan sc{s.sitemap + s.site_off[o], s.site_frame, L.lat_rows[o], L.lat_cols[o], s.site_pitch[o], p.min_contrast,
                                s.colmap + s.col_off[o], s.col_frame, mx_seams(L.cols[o])};
                {
                    TimedScope ts(c, "k_extrema_pack", o, es);
                    HIPCHK(c, mx_launch_pack(es, sc, L.rows[o], L.lat_words[o], nf, bits ? bits + L.bits_offset[o] : nullptr, s.lflags + L.bits_offset[o],
                                             L.bits_frame_words, fused_rows[o]));
                }
                // the lattice rows whose windows straddle a strip's first image row (3a a multiple of the strip's rows, a power of two:
                // a = 32, 64, ... or 16, 32, ...) are not in the site map: the plain scan kernel runs on exactly those rows
                const int sr = fused_rows[o];
                const int n_straddle = (L.lat_rows[o] - 1) / sr;
                TimedScope ts(c, "k_extrema_w3", o, es);
                if (n_straddle > 0)
                    hipLaunchKernelGGL(k_extrema_w3<false>, dim3((L.lat_words[o] + 3) / 4, n_straddle, nf), dim3(256), 0, es, pyr, pframe, g, o, bits, s.lflags,
                                       L.bits_frame_words, sr, sr);
                if (es != c->stream) {  // the maps' reader is on another stream than their writer: mark its end for the next chunk
                    HIPCHK(c, hipEventRecord(c->ev_pack, es));
                    c->pack_pending = true;
                }
            } else if (p.extrema_window == 3 && diag_skip_scan(c, o)) {
                // (diagnostics build, VSLAM_DIAG_SKIP_SCAN: timing knock-out - the flag words keep what an earlier call wrote)
            } else if (p.extrema_window == 3) {
                const dim3 eg((L.lat_words[o] + 3) / 4, L.lat_rows[o], nf);
                TimedScope ts(c, "k_extrema_w3", o, es);
                if (p.localize)
                    hipLaunchKernelGGL(k_extrema_w3<true>, eg, dim3(256), 0, es, pyr, pframe, g, o, bits, s.lflags, L.bits_frame_words, 0, 1);
                else
                    hipLaunchKernelGGL(k_extrema_w3<false>, eg, dim3(256), 0, es, pyr, pframe, g, o, bits, s.lflags, L.bits_frame_words, 0, 1);
            } else
                hipLaunchKernelGGL(k_extrema, dim3((L.lat_cols[o] + 255) / 256, L.lat_rows[o], nf * 3), dim3(256), 0, es, pyr,
                                   pframe, g, o, bits, s.lflags, L.bits_frame_words);
            HIPCHK(c, hipGetLastError());
        }
        // compact this octave's points right away (appending to the frame's list): on the side
        // stream it overlaps the next octave's kernels instead of forming a serial tail
        if (do_extrema && points && counts) {
            StreamSwap sw(c, side ? side : c->stream);
            const size_t entries = (size_t)3 * L.lat_rows[o] * L.lat_words[o];
            if (p.extrema_dense) {
                DenseDogEntries ent{s.lflags + L.bits_offset[o], L.bits_frame_words, pyr, pframe, dense_geom(L, o, p.min_contrast, nf), o, points};
                TRY(enqueue_compaction(c, ent, entries, nf, s.cws, p.dog_cap, counts, o > 0 ? 1 : 0));
            } else {
                DogEntries ent{s.lflags, L.bits_frame_words, pyr, pframe, g, o, o + 1, points};
                TRY(enqueue_compaction(c, ent, entries, nf, s.cws, p.dog_cap, counts, o > 0 ? 1 : 0));
            }
            if (after_list) TRY(after_list(o));  // on the stream the list is written on, behind octave o's records
        }
        }  // oo
    }
    if (side && up) HIPCHK(c, hipEventRecord(c->ev_chunk, c->stream));
    return VSLAM_OK;
}

// Words of keypoint-flag scratch per frame for the Harris chain: 4 ballot words per strip row.
static size_t harris_flag_words(int rows, int cols) { return (size_t)rows * ((cols + HS_STRIP_W - 1) / HS_STRIP_W) * 4; }

// Harris chain on nf device frames with dense rows: response (required buffer), optional mask /
// nms2 / keypoint list, all from the single-pass wave-strip kernel - its aligned form when every
// row and frame starts on a dword, the any-width form otherwise.
static int enqueue_harris(vslam_ctx* c, const uint8_t* frames, size_t fframe, int rows, int cols, int nf, float k,
                          float* resp, uint8_t* mask, float* nms2, unsigned long long* hflags, vslam_kp* kps,
                          unsigned int cap, unsigned int* counts, unsigned int* chunk_ws) {
    const size_t N = (size_t)rows * cols;
    if (N >= ((size_t)1 << 31)) return fail(c, VSLAM_ERR_UNSUPPORTED, "Harris: images of 2^31 pixels or more are not supported (32-bit row offsets)");
    const bool aligned = cols % 4 == 0 && fframe % 4 == 0;
    HarrisStripArgs a;
    a.img = frames;
    a.frame = fframe;
    a.rows = rows;
    a.cols = cols;
    a.k = k;
    a.resp = resp;
    a.mask = mask;
    a.nms2 = nms2;
    a.flags = hflags;
    a.nstrips = (cols + HS_STRIP_W - 1) / HS_STRIP_W;
    a.fframe = (size_t)rows * a.nstrips * 4;
    if (!c->dump) HIPCHK(c, hipMalloc((void**)&c->dump, 256));
    a.dump = c->dump;
    // enough waves to fill the chip several times over, long enough strips to amortise the
    // 9-row pipeline fill
    const long want_seg = std::max<long>(1, 12288 / ((long)a.nstrips * nf));
    a.seg = (int)std::min<long>(rows, std::max<long>(16, (rows + want_seg - 1) / want_seg));
    const int nseg = (rows + a.seg - 1) / a.seg;
    if (aligned)
        LAUNCH(c, "k_harris_strip", k_harris_strip<false>, dim3((a.nstrips * nseg + 3) / 4, 1, nf), dim3(256), a);
    else
        LAUNCH(c, "k_harris_strip", k_harris_strip<true>, dim3((a.nstrips * nseg + 3) / 4, 1, nf), dim3(256), a);
    if (hflags && kps && counts) {
        HarrisStripEntries ent{hflags, a.fframe, rows, cols, a.nstrips, resp, N, kps};
        TRY(enqueue_compaction(c, ent, (size_t)rows * a.nstrips, nf, chunk_ws, cap, counts, 0));
    }
    return VSLAM_OK;
}

static int h2d(vslam_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes,
               size_t rows) {
    // dense rows are one linear copy: hipMemcpy2D degrades to a copy per row for widths that are
    // not a multiple of 4 bytes (9 ms instead of 0.1 ms for a 1754x1240 image)
    if (dpitch == width_bytes && spitch == width_bytes)
        HIPCHK(c, hipMemcpyAsync(dst, src, width_bytes * rows, hipMemcpyHostToDevice, c->stream));
    else
        HIPCHK(c, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyHostToDevice, c->stream));
    return VSLAM_OK;
}
static int d2h(vslam_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes,
               size_t rows) {
    if (dpitch == width_bytes && spitch == width_bytes)
        HIPCHK(c, hipMemcpyAsync(dst, src, width_bytes * rows, hipMemcpyDeviceToHost, c->stream));
    else
        HIPCHK(c, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyDeviceToHost, c->stream));
    return VSLAM_OK;
}


extern "C" {

// ------------------------------------------------------------------------------ lifecycle

int vslam_ctx_create(int device, void* stream, vslam_ctx** out) {
    if (!out || device < 0) return VSLAM_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device >= n) return VSLAM_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return VSLAM_ERR_HIP;
    vslam_ctx* c = new (std::nothrow) vslam_ctx();
    if (!c) return VSLAM_ERR_NOMEM;
    c->device = device;
    {
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > kMaxDynLds) g_lds_limit = lds;
        (void)hipGetLastError();
    }
    if (stream == VSLAM_STREAM_LEGACY) {
        c->stream = nullptr;  // the NULL stream itself: every HIP call below takes it as "stream 0"
    } else if (stream) {
        c->stream = (hipStream_t)stream;
    } else {
        if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
            delete c;
            return VSLAM_ERR_HIP;
        }
        c->own_stream = true;
    }
    {
        const char* e = std::getenv("VSLAM_MX");
        c->mx = e && e[0] == '1';
        if (const char* mf = VSLAM_DIAG_ENV("VSLAM_MX_FORM")) mx_set_form(std::atoi(mf));  // diagnostics build: A/B of the two MFMA shapes
        const char* ff = std::getenv("VSLAM_F32_FUSED");
        c->f32_fused = ff && ff[0] == '1';
        const char* es = VSLAM_DIAG_ENV("VSLAM_ORIENT_SCALAR");
        c->orient_scalar_form = es && es[0] == '1';
        sched_init_from_env(c);
    }
    *out = c;
    return VSLAM_OK;
}

int vslam_ctx_set_matrix_path(vslam_ctx* c, int on) {
    if (!c) return VSLAM_ERR_INVALID;
    c->mx = on != 0;
    return VSLAM_OK;
}

int vslam_ctx_get_matrix_path(const vslam_ctx* c) { return c && c->mx ? 1 : 0; }

int vslam_ctx_set_f32_fused(vslam_ctx* c, int on) {
    if (!c) return VSLAM_ERR_INVALID;
    c->f32_fused = on != 0;
    return VSLAM_OK;
}

int vslam_ctx_get_f32_fused(const vslam_ctx* c) { return c && c->f32_fused ? 1 : 0; }

int vslam_ctx_destroy(vslam_ctx* c) {
    if (!c) return VSLAM_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (int i = 0; i < vslam_ctx::kAux; ++i)  // a failed batch call may have left side-stream work un-joined
        if (c->aux[i]) (void)hipStreamSynchronize(c->aux[i]);
    for (auto& kv : c->taps) (void)hipFree(kv.second);
    for (auto& kv : c->strip_taps) (void)hipFree(kv.second);
    for (auto& kv : c->tile_taps) (void)hipFree(kv.second);
    for (auto& kv : c->mx_taps) (void)hipFree(kv.second);
    if (c->ws) (void)hipFree(c->ws);
    for (auto& b : c->block_cache) (void)hipFree(b.second);
    if (c->loc_lut) (void)hipFree(c->loc_lut);
    if (c->dump) (void)hipFree(c->dump);
    for (auto& kv : c->orient_taps) (void)hipFree(kv.second);
    sched_destroy(c);
    for (hipEvent_t e : {c->ev_phase, c->ev_up2, c->ev_chunk, c->ev_pack, c->ev_list0, c->ev_edge, c->ev_or_fork, c->ev_or_join[0], c->ev_or_join[1]})
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->ev_oct)
        if (e) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return VSLAM_OK;
}

int vslam_ctx_sync(vslam_ctx* c) {
    TRY(bind_device(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VSLAM_OK;
}

const char* vslam_last_error(const vslam_ctx* c) { return c ? c->err.c_str() : "null context"; }
const char* vslam_kernel_names(void) { return kKernelNames; }


// ------------------------------------------------------------------- host-buffer primitives

int vslam_gaussian_blur_u8(vslam_ctx* c, const uint8_t* src, int rows, int cols, size_t step, int ksize,
                           double sigma, uint8_t* dst, size_t dst_step) {
    TRY(bind_device(c));
    ARGCHK(c, src && dst && rows > 0 && cols > 0 && step >= (size_t)cols && dst_step >= (size_t)cols, "blur: bad image");
    const int n = ksize > 0 ? ksize : (sigma > 0 ? gauss_ksize_u8(sigma) : -1);
    ARGCHK(c, n > 0 && (n & 1) && n <= VSLAM_MAX_KSIZE, "blur: kernel size must be odd (or 0 with sigma > 0)");
    const size_t P = (size_t)rows * cols;
    TRY(ws_reserve(c, 2 * ws_need(P) + ws_need(2 * P)));
    uint8_t* d_src = ws_take<uint8_t>(c, P);
    uint8_t* d_dst = ws_take<uint8_t>(c, P);
    uint16_t* d_h = ws_take<uint16_t>(c, P);
    TRY(h2d(c, d_src, cols, src, step, cols, rows));
    TRY(enqueue_blur(c, d_src, cols, P, d_dst, cols, P, d_h, rows, cols, 1, n, sigma));
    TRY(d2h(c, dst, dst_step, d_dst, cols, cols, rows));
    return vslam_ctx_sync(c);
}

int vslam_sobel_k1_u8_f32(vslam_ctx* c, const uint8_t* src, int rows, int cols, size_t step, int dx, int dy,
                          float* dst, size_t dst_step) {
    TRY(bind_device(c));
    ARGCHK(c, src && dst && rows > 0 && cols > 0 && step >= (size_t)cols && dst_step >= 4 * (size_t)cols, "sobel: bad image");
    ARGCHK(c, (dx == 1 && dy == 0) || (dx == 0 && dy == 1), "sobel: (dx,dy) must be (1,0) or (0,1)");
    const size_t P = (size_t)rows * cols;
    TRY(ws_reserve(c, ws_need(P) + ws_need(4 * P)));
    uint8_t* d_src = ws_take<uint8_t>(c, P);
    float* d_dst = ws_take<float>(c, P);
    TRY(h2d(c, d_src, cols, src, step, cols, rows));
    LAUNCH(c, "k_sobel_k1", k_sobel_k1, grid_rows(cols, rows), dim3(256), d_src, (size_t)cols, d_dst, (size_t)cols, rows,
           cols, dx);
    TRY(d2h(c, dst, dst_step, d_dst, 4 * (size_t)cols, 4 * (size_t)cols, rows));
    return vslam_ctx_sync(c);
}

int vslam_resize_linear2x_u8(vslam_ctx* c, const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst,
                             size_t dst_step) {
    TRY(bind_device(c));
    ARGCHK(c, src && dst && rows > 0 && cols > 0 && step >= (size_t)cols && dst_step >= 2 * (size_t)cols, "resize2x: bad image");
    const size_t P = (size_t)rows * cols;
    TRY(ws_reserve(c, ws_need(P) + ws_need(4 * P)));
    uint8_t* d_src = ws_take<uint8_t>(c, P);
    uint8_t* d_dst = ws_take<uint8_t>(c, 4 * P);
    TRY(h2d(c, d_src, cols, src, step, cols, rows));
    if (cols % 4 == 0)
        LAUNCH(c, "k_resize_linear2x_slide", k_resize_linear2x_slide, dim3((cols / 4 + 255) / 256, (rows + 15) / 16, 1), dim3(256),
               d_src, (size_t)cols, P, d_dst, 4 * P, 2 * cols, rows, cols, 16);
    else
        LAUNCH(c, "k_resize_linear2x", k_resize_linear2x, grid_rows(2 * cols, 2 * rows), dim3(256), d_src, (size_t)cols, P,
               d_dst, (size_t)2 * cols, 4 * P, rows, cols);
    TRY(d2h(c, dst, dst_step, d_dst, 2 * (size_t)cols, 2 * (size_t)cols, 2 * (size_t)rows));
    return vslam_ctx_sync(c);
}

int vslam_resize_nearest_half_u8(vslam_ctx* c, const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst,
                                 size_t dst_step) {
    TRY(bind_device(c));
    int dr, dc;
    half_size(rows, cols, &dr, &dc);
    ARGCHK(c, src && dst && rows > 0 && cols > 0 && dr > 0 && dc > 0 && step >= (size_t)cols && dst_step >= (size_t)dc,
           "resize half: bad image");
    const size_t P = (size_t)rows * cols, Q = (size_t)dr * dc;
    TRY(ws_reserve(c, ws_need(P) + ws_need(Q)));
    uint8_t* d_src = ws_take<uint8_t>(c, P);
    uint8_t* d_dst = ws_take<uint8_t>(c, Q);
    TRY(h2d(c, d_src, cols, src, step, cols, rows));
    LAUNCH(c, "k_resize_nearest_half", k_resize_nearest_half, grid_rows(dc, dr), dim3(256), d_src, (size_t)cols, P, d_dst,
           (size_t)dc, Q, rows, cols, dr, dc);
    TRY(d2h(c, dst, dst_step, d_dst, dc, dc, dr));
    return vslam_ctx_sync(c);
}

int vslam_convert_scale_abs_f32(vslam_ctx* c, const float* src, int rows, int cols, size_t step, uint8_t* dst,
                                size_t dst_step) {
    TRY(bind_device(c));
    ARGCHK(c, src && dst && rows > 0 && cols > 0 && step >= 4 * (size_t)cols && dst_step >= (size_t)cols, "convertScaleAbs: bad image");
    const size_t P = (size_t)rows * cols;
    TRY(ws_reserve(c, ws_need(4 * P) + ws_need(P)));
    float* d_src = ws_take<float>(c, P);
    uint8_t* d_dst = ws_take<uint8_t>(c, P);
    TRY(h2d(c, d_src, 4 * (size_t)cols, src, step, 4 * (size_t)cols, rows));
    LAUNCH(c, "k_convert_scale_abs", k_convert_scale_abs, grid_rows(cols, rows), dim3(256), d_src, (size_t)cols, d_dst,
           (size_t)cols, rows, cols);
    TRY(d2h(c, dst, dst_step, d_dst, cols, cols, rows));
    return vslam_ctx_sync(c);
}

// -------------------------------------------------------------------------------- Harris

int vslam_harris_from_grad_f32(vslam_ctx* c, const float* ix, const float* iy, int rows, int cols, size_t step,
                               float k, int window, float* resp, size_t resp_step) {
    TRY(bind_device(c));
    ARGCHK(c, ix && iy && resp && rows > 0 && cols > 0 && step >= 4 * (size_t)cols && resp_step >= 4 * (size_t)cols,
           "HarrisCorner: bad image");
    ARGCHK(c, window >= 1 && (window & 1), "HarrisCorner: window must be odd");
    const size_t P = (size_t)rows * cols, rb = 4 * (size_t)cols;
    TRY(ws_reserve(c, 3 * ws_need(4 * P)));
    float* d_ix = ws_take<float>(c, P);
    float* d_iy = ws_take<float>(c, P);
    float* d_r = ws_take<float>(c, P);
    TRY(h2d(c, d_ix, rb, ix, step, rb, rows));
    TRY(h2d(c, d_iy, rb, iy, step, rb, rows));
    LAUNCH(c, "k_harris_from_grad", k_harris_from_grad, grid_rows(cols, rows), dim3(256), d_ix, d_iy, (size_t)cols, rows,
           cols, k, (window - 1) / 2, d_r, (size_t)cols);
    TRY(d2h(c, resp, resp_step, d_r, rb, rb, rows));
    return vslam_ctx_sync(c);
}

int vslam_harris_response_u8(vslam_ctx* c, const uint8_t* img, int rows, int cols, size_t step, float k, int window,
                             float* resp, size_t resp_step) {
    TRY(bind_device(c));
    ARGCHK(c, img && resp && rows > 0 && cols > 0 && step >= (size_t)cols && resp_step >= 4 * (size_t)cols,
           "harris_response: bad image");
    if (window != 3) return fail(c, VSLAM_ERR_UNSUPPORTED, "harris_response: fused kernel implements windowSize 3 (Harris_corners.cpp:34); use the per-stage entry points for other windows");
    const size_t P = (size_t)rows * cols;
    TRY(ws_reserve(c, ws_need(P) + ws_need(4 * P)));
    uint8_t* d_img = ws_take<uint8_t>(c, P);
    float* d_r = ws_take<float>(c, P);
    TRY(h2d(c, d_img, cols, img, step, cols, rows));
    TRY(enqueue_harris(c, d_img, P, rows, cols, 1, k, d_r, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr));
    TRY(d2h(c, resp, resp_step, d_r, 4 * (size_t)cols, 4 * (size_t)cols, rows));
    return vslam_ctx_sync(c);
}

static int nms_strict_common(vslam_ctx* c, const void* src, int elem, int rows, int cols, size_t step, int window,
                             uint8_t* mask, size_t mask_step) {
    TRY(bind_device(c));
    ARGCHK(c, src && mask && rows > 0 && cols > 0 && step >= (size_t)elem * cols && mask_step >= (size_t)cols, "NonMaximumSuppression: bad image");
    ARGCHK(c, window >= 1 && (window & 1), "NonMaximumSuppression: windowSize must be odd");
    const size_t P = (size_t)rows * cols, rb = (size_t)elem * cols;
    TRY(ws_reserve(c, ws_need(elem * P) + ws_need(P)));
    char* d_src = ws_take<char>(c, elem * P);
    uint8_t* d_m = ws_take<uint8_t>(c, P);
    TRY(h2d(c, d_src, rb, src, step, rb, rows));
    const int p = (window - 1) / 2;
    if (elem == 1)
        LAUNCH(c, "k_nms_strict_generic", k_nms_strict_generic<uint8_t>, grid_rows(cols, rows), dim3(256),
               (const uint8_t*)d_src, (size_t)cols, rows, cols, p, d_m, (size_t)cols);
    else
        LAUNCH(c, "k_nms_strict_generic", k_nms_strict_generic<float>, grid_rows(cols, rows), dim3(256),
               (const float*)d_src, (size_t)cols, rows, cols, p, d_m, (size_t)cols);
    TRY(d2h(c, mask, mask_step, d_m, cols, cols, rows));
    return vslam_ctx_sync(c);
}

int vslam_nms_strict_u8(vslam_ctx* c, const uint8_t* src, int rows, int cols, size_t step, int window, uint8_t* mask,
                        size_t mask_step) {
    return nms_strict_common(c, src, 1, rows, cols, step, window, mask, mask_step);
}
int vslam_nms_strict_f32(vslam_ctx* c, const float* src, int rows, int cols, size_t step, int window, uint8_t* mask,
                         size_t mask_step) {
    return nms_strict_common(c, src, 4, rows, cols, step, window, mask, mask_step);
}

int vslam_nms2_f32(vslam_ctx* c, const float* resp, int rows, int cols, size_t step, int window, float* out,
                   size_t out_step, float* true_max) {
    TRY(bind_device(c));
    ARGCHK(c, resp && out && rows > 0 && cols > 0 && step >= 4 * (size_t)cols && out_step >= 4 * (size_t)cols && window >= 1,
           "NMS2: bad arguments");
    const size_t P = (size_t)rows * cols, rb = 4 * (size_t)cols;
    TRY(ws_reserve(c, 2 * ws_need(4 * P) + 256));
    float* d_r = ws_take<float>(c, P);
    float* d_o = ws_take<float>(c, P);
    unsigned int* d_max = ws_take<unsigned int>(c, 1);
    TRY(h2d(c, d_r, rb, resp, step, rb, rows));
    HIPCHK(c, hipMemsetAsync(d_max, 0, 4, c->stream));
    LAUNCH(c, "k_nms2_generic", k_nms2_generic, grid_rows(cols, rows), dim3(256), d_r, (size_t)cols, rows, cols,
           (window - 1) / 2, d_o, (size_t)cols, d_max);
    TRY(d2h(c, out, out_step, d_o, rb, rb, rows));
    unsigned int bits = 0;
    HIPCHK(c, hipMemcpyAsync(&bits, d_max, 4, hipMemcpyDeviceToHost, c->stream));
    TRY(vslam_ctx_sync(c));
    if (true_max) std::memcpy(true_max, &bits, 4);
    return VSLAM_OK;
}

int vslam_harris_keypoints_u8(vslam_ctx* c, const uint8_t* img, int rows, int cols, size_t step, float k,
                              vslam_kp* out, size_t cap, size_t* count) {
    TRY(bind_device(c));
    ARGCHK(c, img && count && rows > 0 && cols > 0 && step >= (size_t)cols && (out || cap == 0), "harris_keypoints: bad arguments");
    const size_t P = (size_t)rows * cols;
    const unsigned int dcap = (unsigned int)std::min<size_t>(cap, 0x7fffffff);
    TRY(ws_reserve(c, ws_need(P) + ws_need(4 * P) + ws_need(harris_flag_words(rows, cols) * 8) + ws_need(sizeof(vslam_kp) * (size_t)dcap) + 256 +
                          ws_need(4 * compaction_ws_elems(harris_flag_words(rows, cols), 1))));
    uint8_t* d_img = ws_take<uint8_t>(c, P);
    float* d_r = ws_take<float>(c, P);
    unsigned long long* d_f = ws_take<unsigned long long>(c, harris_flag_words(rows, cols));
    vslam_kp* d_k = ws_take<vslam_kp>(c, dcap);
    unsigned int* d_n = ws_take<unsigned int>(c, 1);
    unsigned int* d_cws = ws_take<unsigned int>(c, compaction_ws_elems(harris_flag_words(rows, cols), 1));
    TRY(h2d(c, d_img, cols, img, step, cols, rows));
    TRY(enqueue_harris(c, d_img, P, rows, cols, 1, k, d_r, nullptr, nullptr, d_f, d_k, dcap, d_n, d_cws));
    unsigned int n = 0;
    HIPCHK(c, hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, c->stream));
    TRY(vslam_ctx_sync(c));
    *count = n;
    const size_t m = std::min<size_t>(n, dcap);
    if (m) HIPCHK(c, hipMemcpy(out, d_k, m * sizeof(vslam_kp), hipMemcpyDeviceToHost));
    return VSLAM_OK;
}

// ------------------------------------------------------------------------------ DoG pyramid

int vslam_pyramid_build_u8(vslam_ctx* c, const uint8_t* img, int rows, int cols, size_t step, int n_octaves,
                           double sigma0, vslam_pyramid** out) {
    TRY(bind_device(c));
    ARGCHK(c, img && out && rows > 0 && cols > 0 && step >= (size_t)cols && sigma0 > 0, "GaussPyramid: bad arguments");
    *out = nullptr;
    if (n_octaves <= 0) n_octaves = auto_num_octaves(rows, cols);  // GaussPyramid.hpp:18-21
    ARGCHK(c, n_octaves >= 1 && n_octaves <= VSLAM_MAX_OCTAVES, "GaussPyramid: octave count out of range");
    vslam_params p;
    vslam_params_default(&p, rows, cols);
    p.n_octaves = n_octaves;
    p.sigma0 = sigma0;
    vslam_batch_layout L;
    if (make_layout(&p, &L) != VSLAM_OK) return fail(c, VSLAM_ERR_INVALID, "GaussPyramid: image too small for the octave count");
    vslam_pyramid* py = new (std::nothrow) vslam_pyramid();
    if (!py) return fail(c, VSLAM_ERR_NOMEM, "host allocation failed");
    py->ctx = c;
    py->params = p;
    py->layout = L;
    py->info.n_octaves = n_octaves;
    py->info.n_levels = VSLAM_NUM_LEVELS;
    py->info.n_dogs = VSLAM_NUM_DOGS;
    py->info.sigma0 = sigma0;
    size_t sum_p = 0;
    for (int o = 0; o < n_octaves; ++o) {
        py->info.rows[o] = L.rows[o];
        py->info.cols[o] = L.cols[o];
        py->base_off[o] = sum_p;
        sum_p += (size_t)L.rows[o] * L.pitch[o];
        for (int l = 0; l < VSLAM_NUM_LEVELS; ++l) {
            py->info.sigma[o][l] = sigma_at(sigma0, o, l);
            py->info.ksize[o][l] = gauss_ksize_u8(py->info.sigma[o][l]);
        }
    }
    auto cleanup = [&](int rc) {
        vslam_pyramid_destroy(py);
        return rc;
    };
    py->d_block = (uint8_t*)block_alloc(c, L.pyramid_frame_bytes, &py->block_cap);
    py->d_bases = (uint8_t*)block_alloc(c, sum_p, &py->bases_cap);
    if (!py->d_block || !py->d_bases) return cleanup(fail(c, VSLAM_ERR_NOMEM, "device allocation failed (pyramid)"));
    const size_t N = (size_t)rows * cols;
    int rc = ws_reserve(c, ws_need(N) + dog_scratch_bytes(L, sigma0, 1));
    if (rc) return cleanup(rc);
    uint8_t* d_img = ws_take<uint8_t>(c, N);
    DogScratch s;
    if ((rc = dog_scratch_take(c, L, sigma0, 1, s))) return cleanup(rc);
    if ((rc = h2d(c, d_img, cols, img, step, cols, rows))) return cleanup(rc);
    if ((rc = enqueue_dog(c, p, L, d_img, cols, N, 1, py->d_block, L.pyramid_frame_bytes, s, nullptr, false, nullptr,
                          nullptr)))
        return cleanup(rc);
    if (hipMemcpyAsync(py->d_bases, s.bases, sum_p, hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
        return cleanup(fail(c, VSLAM_ERR_HIP, "device copy failed (bases)"));
    if ((rc = vslam_ctx_sync(c))) return cleanup(rc);
    *out = py;
    return VSLAM_OK;
}

int vslam_pyramid_destroy(vslam_pyramid* py) {
    if (!py) return VSLAM_ERR_INVALID;
    if (py->ctx) (void)hipSetDevice(py->ctx->device);
    block_release(py->ctx, py->d_block, py->block_cap);
    block_release(py->ctx, py->d_bases, py->bases_cap);
    delete py;
    return VSLAM_OK;
}

int vslam_pyramid_get_info(const vslam_pyramid* py, vslam_pyramid_info* out) {
    if (!py || !out) return VSLAM_ERR_INVALID;
    *out = py->info;
    return VSLAM_OK;
}

static int pyramid_fetch(const vslam_pyramid* py, int octave, const uint8_t* d_src, uint8_t* dst, size_t dst_step) {
    vslam_ctx* c = py->ctx;
    TRY(bind_device(c));
    const int rows = py->layout.rows[octave], cols = py->layout.cols[octave], pitch = py->layout.pitch[octave];
    ARGCHK(c, dst && dst_step >= (size_t)cols, "pyramid getter: bad destination");
    if (pitch != cols) {  // pitched plane: pack the rows on the device, then one linear copy
        const size_t P = (size_t)rows * cols;
        TRY(ws_reserve(c, ws_need(P)));
        uint8_t* d_dense = ws_take<uint8_t>(c, P);
        LAUNCH(c, "k_pack_rows", k_pack_rows, grid_rows(cols, rows), dim3(256), d_src, pitch, d_dense, rows, cols);
        d_src = d_dense;
    }
    TRY(d2h(c, dst, dst_step, d_src, cols, cols, rows));
    return vslam_ctx_sync(c);
}

int vslam_pyramid_get_base(const vslam_pyramid* py, int octave, uint8_t* dst, size_t dst_step) {
    if (!py) return VSLAM_ERR_INVALID;
    if (octave < 0 || octave >= py->layout.n_octaves) return fail(py->ctx, VSLAM_ERR_RANGE, "octave out of range");
    return pyramid_fetch(py, octave, py->d_bases + py->base_off[octave], dst, dst_step);
}

int vslam_pyramid_get_gauss(const vslam_pyramid* py, int octave, int level, uint8_t* dst, size_t dst_step) {
    if (!py) return VSLAM_ERR_INVALID;
    if (octave < 0 || octave >= py->layout.n_octaves || level < 0 || level >= VSLAM_NUM_LEVELS)
        return fail(py->ctx, VSLAM_ERR_RANGE, "octave/level out of range");
    const size_t P = (size_t)py->layout.rows[octave] * py->layout.pitch[octave];
    return pyramid_fetch(py, octave, py->d_block + py->layout.octave_offset[octave] + (size_t)level * P, dst, dst_step);
}

int vslam_pyramid_get_dog(const vslam_pyramid* py, int octave, int level, uint8_t* dst, size_t dst_step) {
    if (!py) return VSLAM_ERR_INVALID;
    if (octave < 0 || octave >= py->layout.n_octaves || level < 0 || level >= VSLAM_NUM_DOGS)
        return fail(py->ctx, VSLAM_ERR_RANGE, "octave/level out of range");
    const size_t P = (size_t)py->layout.rows[octave] * py->layout.pitch[octave];
    return pyramid_fetch(py, octave,
                         py->d_block + py->layout.octave_offset[octave] + (size_t)(VSLAM_NUM_LEVELS + level) * P, dst,
                         dst_step);
}

int vslam_pyramid_get_gradients(const vslam_pyramid* py, int octave, int level, float* grad_x, float* grad_y, float* mag,
                                float* orient, size_t dst_step) {
    if (!py) return VSLAM_ERR_INVALID;
    vslam_ctx* c = py->ctx;
    TRY(bind_device(c));
    if (octave < 0 || octave >= py->layout.n_octaves || level < 0 || level >= VSLAM_NUM_LEVELS)
        return fail(c, VSLAM_ERR_RANGE, "octave/level out of range");
    const int rows = py->layout.rows[octave], cols = py->layout.cols[octave];
    ARGCHK(c, dst_step >= 4 * (size_t)cols, "pyramid gradients: bad destination step");
    float* host[4] = {grad_x, grad_y, mag, orient};
    const size_t P = (size_t)rows * cols;
    TRY(ws_reserve(c, 4 * ws_need(4 * P)));
    float* dev[4];
    for (int i = 0; i < 4; ++i) {
        float* d = ws_take<float>(c, P);
        dev[i] = host[i] ? d : nullptr;
    }
    const int pitch = py->layout.pitch[octave];
    const uint8_t* g = py->d_block + py->layout.octave_offset[octave] + (size_t)level * rows * pitch;
    if (c->f32_fused)
        LAUNCH(c, "k_level_gradients", k_level_gradients<true>, grid_rows(cols, rows), dim3(256), g, pitch, rows, cols, dev[0], dev[1], dev[2], dev[3]);
    else
        LAUNCH(c, "k_level_gradients", k_level_gradients<false>, grid_rows(cols, rows), dim3(256), g, pitch, rows, cols, dev[0], dev[1], dev[2], dev[3]);
    for (int i = 0; i < 4; ++i)
        if (host[i]) TRY(d2h(c, host[i], dst_step, dev[i], 4 * (size_t)cols, 4 * (size_t)cols, rows));
    return vslam_ctx_sync(c);
}

static int dog_points_host(vslam_ctx* c, const vslam_pyramid* py, int octave, int window, int min_contrast, int localize,
                           uint64_t* bits, vslam_point* out, size_t cap, size_t* count) {
    TRY(bind_device(c));
    ARGCHK(c, py && py->ctx == c && count && (out || cap == 0), "initialKeypointDetection: bad arguments");
    if (octave < 0 || octave >= py->layout.n_octaves) return fail(c, VSLAM_ERR_RANGE, "octave out of range");
    ARGCHK(c, window >= 3 && (window & 1), "initialKeypointDetection: windowSize must be odd and >= 3");
    vslam_params p = py->params;
    p.extrema_window = window;
    p.min_contrast = min_contrast;
    p.localize = localize;
    p.dog_cap = (uint32_t)std::min<size_t>(cap, 0x7fffffff);
    vslam_batch_layout L;
    if (make_layout(&p, &L) != VSLAM_OK) return fail(c, VSLAM_ERR_INVALID, "bad extrema parameters");
    ExtGeom g;
    if (p.localize) TRY(ensure_loc_lut(c));
    fill_geom(c, p, L, g);
    const size_t words = L.bits_frame_words;
    TRY(ws_reserve(c, 2 * ws_need(words * 8) + ws_need(sizeof(vslam_point) * (size_t)p.dog_cap) + 256 + ws_need(4 * compaction_ws_elems(words, 1))));
    unsigned long long* d_bits = ws_take<unsigned long long>(c, words);
    unsigned long long* d_lf = ws_take<unsigned long long>(c, words);
    vslam_point* d_pts = ws_take<vslam_point>(c, p.dog_cap);
    unsigned int* d_n = ws_take<unsigned int>(c, 1);
    unsigned int* d_cws = ws_take<unsigned int>(c, compaction_ws_elems(words, 1));
    HIPCHK(c, hipMemsetAsync(d_n, 0, 4, c->stream));
    const size_t ow = (size_t)3 * L.lat_rows[octave] * L.lat_words[octave];
    if (ow) {
        if (window == 3) {
            const dim3 eg((L.lat_words[octave] + 3) / 4, L.lat_rows[octave], 1);
            if (localize)
                LAUNCH(c, "k_extrema_w3", k_extrema_w3<true>, eg, dim3(256), py->d_block, L.pyramid_frame_bytes, g, octave, d_bits, d_lf, words, 0, 1);
            else
                LAUNCH(c, "k_extrema_w3", k_extrema_w3<false>, eg, dim3(256), py->d_block, L.pyramid_frame_bytes, g, octave, d_bits, d_lf, words, 0, 1);
        } else {
            LAUNCH(c, "k_extrema", k_extrema, dim3((L.lat_cols[octave] + 255) / 256, L.lat_rows[octave], 3), dim3(256),
                   py->d_block, L.pyramid_frame_bytes, g, octave, d_bits, d_lf, words);
        }
        DogEntries ent{d_lf, words, py->d_block, L.pyramid_frame_bytes, g, octave, octave + 1, d_pts};
        TRY(enqueue_compaction(c, ent, ow, 1, d_cws, p.dog_cap, d_n, 0));
    }
    unsigned int n = 0;
    HIPCHK(c, hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, c->stream));
    if (bits && ow)
        HIPCHK(c, hipMemcpyAsync(bits, d_bits + L.bits_offset[octave], ow * 8, hipMemcpyDeviceToHost, c->stream));
    TRY(vslam_ctx_sync(c));
    *count = n;
    const size_t m = std::min<size_t>(n, p.dog_cap);
    if (m) HIPCHK(c, hipMemcpy(out, d_pts, m * sizeof(vslam_point), hipMemcpyDeviceToHost));
    return VSLAM_OK;
}

int vslam_dog_extrema(vslam_ctx* c, const vslam_pyramid* py, int octave, int window, int min_contrast, uint64_t* bits,
                      vslam_point* out, size_t cap, size_t* count) {
    return dog_points_host(c, py, octave, window, min_contrast, 0, bits, out, cap, count);
}

int vslam_dog_extrema_dense(vslam_ctx* c, const vslam_pyramid* py, int octave, int min_contrast, uint64_t* bits,
                            vslam_point* out, size_t cap, size_t* count) {
    TRY(bind_device(c));
    ARGCHK(c, py && py->ctx == c && count && (out || cap == 0), "dense extrema: bad arguments");
    if (octave < 0 || octave >= py->layout.n_octaves) return fail(c, VSLAM_ERR_RANGE, "octave out of range");
    ARGCHK(c, min_contrast >= 0 && min_contrast <= 65535, "dense extrema: min_contrast out of range");
    const vslam_batch_layout& L = py->layout;
    DenseGeom g = dense_geom(L, octave, min_contrast, 1);
    const size_t words = (size_t)3 * g.rows * g.wpr;
    const unsigned int ocap = (unsigned int)std::min<size_t>(cap, 0x7fffffff);
    TRY(ws_reserve(c, 2 * ws_need(words * 8) + ws_need(sizeof(vslam_point) * (size_t)ocap) + 256 + ws_need(4 * compaction_ws_elems(words, 1))));
    unsigned long long* d_bits = ws_take<unsigned long long>(c, words);
    unsigned long long* d_lf = ws_take<unsigned long long>(c, words);
    vslam_point* d_pts = ws_take<vslam_point>(c, ocap);
    unsigned int* d_n = ws_take<unsigned int>(c, 1);
    unsigned int* d_cws = ws_take<unsigned int>(c, compaction_ws_elems(words, 1));
    HIPCHK(c, hipMemsetAsync(d_n, 0, 4, c->stream));
    LAUNCH(c, "k_extrema_dense", k_extrema_dense, dim3(((g.cols + 3) / 4 + 255) / 256, (g.rows + g.seg - 1) / g.seg, 1), dim3(256),
           py->d_block, L.pyramid_frame_bytes, g, bits ? d_bits : nullptr, d_lf, words);
    DenseDogEntries ent{d_lf, words, py->d_block, L.pyramid_frame_bytes, g, octave, d_pts};
    TRY(enqueue_compaction(c, ent, words, 1, d_cws, ocap, d_n, 0));
    unsigned int n = 0;
    HIPCHK(c, hipMemcpyAsync(&n, d_n, 4, hipMemcpyDeviceToHost, c->stream));
    if (bits) HIPCHK(c, hipMemcpyAsync(bits, d_bits, words * 8, hipMemcpyDeviceToHost, c->stream));
    TRY(vslam_ctx_sync(c));
    *count = n;
    const size_t m = std::min<size_t>(n, ocap);
    if (m) HIPCHK(c, hipMemcpy(out, d_pts, m * sizeof(vslam_point), hipMemcpyDeviceToHost));
    return VSLAM_OK;
}

int vslam_dog_keypoints(vslam_ctx* c, const vslam_pyramid* py, int octave, int window, vslam_point* out, size_t cap,
                        size_t* count) {
    return dog_points_host(c, py, octave, window, 0, 1, nullptr, out, cap, count);
}

int vslam_localize_points(vslam_ctx* c, const int* diffs, size_t n, int* keep, int* value) {
    TRY(bind_device(c));
    ARGCHK(c, (diffs && keep && value) || n == 0, "FeaturePointLocalization: bad arguments");
    ARGCHK(c, n <= 0x7fffffff, "FeaturePointLocalization: too many points");
    if (n == 0) return VSLAM_OK;
    TRY(ws_reserve(c, ws_need(16 * n) + ws_need(8 * n)));
    int4* d_in = ws_take<int4>(c, n);
    int2* d_out = ws_take<int2>(c, n);
    HIPCHK(c, hipMemcpyAsync(d_in, diffs, 16 * n, hipMemcpyHostToDevice, c->stream));
    TRY(ensure_loc_lut(c));  // same path as the fused kernels: table for small differences, closed form otherwise
    LAUNCH(c, "k_localize_points", k_localize_points, dim3((unsigned)((n + 255) / 256)), dim3(256), d_in, (int)n, d_out, c->loc_lut);
    std::vector<int2> h(n);
    HIPCHK(c, hipMemcpyAsync(h.data(), d_out, 8 * n, hipMemcpyDeviceToHost, c->stream));
    TRY(vslam_ctx_sync(c));
    for (size_t i = 0; i < n; ++i) keep[i] = h[i].x, value[i] = h[i].y;
    return VSLAM_OK;
}

int vslam_filter_keypoints(vslam_ctx* c, const vslam_pyramid* py, int octave, const vslam_point* kps, size_t n,
                           vslam_point* out, size_t cap, size_t* count) {
    TRY(bind_device(c));
    ARGCHK(c, py && py->ctx == c && count && (kps || n == 0) && (out || cap == 0), "filterKeypoints: bad arguments");
    if (octave < 0 || octave >= py->layout.n_octaves) return fail(c, VSLAM_ERR_RANGE, "octave out of range");
    ARGCHK(c, n <= 0x7fffffff, "filterKeypoints: too many keypoints");
    *count = 0;
    if (n == 0) return VSLAM_OK;
    const int rows = py->layout.rows[octave], cols = py->layout.cols[octave], pitch = py->layout.pitch[octave];
    const size_t P = (size_t)rows * cols;  // dense f32 scratch images
    // what the reference would throw on (vector::at, Rect outside the padded Mat) is an error here
    bool used[VSLAM_NUM_LEVELS] = {};
    for (size_t i = 0; i < n; ++i) {
        const vslam_point& k = kps[i];
        if (k.level < 0 || k.level >= VSLAM_NUM_LEVELS || k.col < 0 || k.row < 0 || k.col > cols || k.row > rows || k.octave != octave)
            return fail(c, VSLAM_ERR_RANGE, "filterKeypoints: keypoint outside the octave's data");
        used[k.level] = true;
    }
    std::vector<float> taps[VSLAM_NUM_LEVELS];
    int n_used = 0, max_r = 0;
    size_t tap_elems = 0;
    for (int l = 0; l < VSLAM_NUM_LEVELS; ++l) {
        if (!used[l]) continue;
        const double sigma = 1.5 * py->info.sigma[octave][l];  // Diff_of_Gauss.cpp:346
        if (!gauss_kernel_f32(gauss_ksize_f32(sigma), sigma, taps[l])) return fail(c, VSLAM_ERR_INVALID, "filterKeypoints: bad blur kernel");
        ++n_used;
        max_r = std::max(max_r, (int)taps[l].size() / 2);
        tap_elems += align_up(taps[l].size(), 64);
    }
    const size_t lds = orient_lds_bytes(max_r);
    if (lds > 150 * 1024) return fail(c, VSLAM_ERR_UNSUPPORTED, "filterKeypoints: blur kernel too wide for the LDS strip");
    const unsigned int ocap = (unsigned int)std::min<size_t>(cap, 0x7fffffff);
    TRY(ws_reserve(c, ws_need(sizeof(vslam_point) * n) + ws_need(8 * n) + ws_need(sizeof(vslam_point) * (size_t)ocap) + 256 +
                          ws_need(4 * compaction_ws_elems(n, 1)) + (size_t)n_used * 2 * ws_need(4 * P) + ws_need(4 * tap_elems)));
    vslam_point* d_kps = ws_take<vslam_point>(c, n);
    unsigned long long* d_masks = ws_take<unsigned long long>(c, n);
    vslam_point* d_out = ws_take<vslam_point>(c, ocap);
    unsigned int* d_n = ws_take<unsigned int>(c, 1);
    unsigned int* d_cws = ws_take<unsigned int>(c, compaction_ws_elems(n, 1));
    float* d_taps = ws_take<float>(c, tap_elems);
    OrientLevels lv{};
    HIPCHK(c, hipMemcpyAsync(d_kps, kps, sizeof(vslam_point) * n, hipMemcpyHostToDevice, c->stream));
    size_t toff = 0;
    for (int l = 0; l < VSLAM_NUM_LEVELS; ++l) {
        if (!used[l]) continue;
        float* d_mag = ws_take<float>(c, P);
        float* d_ori = ws_take<float>(c, P);
        const uint8_t* g = py->d_block + py->layout.octave_offset[octave] + (size_t)l * rows * pitch;
        // processGradients for the level (GaussPyramid.cpp:65-104): magnitude and orientation only
        // (the orientation image is only BINNED here: no bin depends on the arctangent's variant, kernels_aux.hip.h)
        LAUNCH(c, "k_level_gradients", k_level_gradients<false>, grid_rows(cols, rows), dim3(256), g, pitch, rows, cols, (float*)nullptr,
               (float*)nullptr, d_mag, d_ori);
        HIPCHK(c, hipMemcpyAsync(d_taps + toff, taps[l].data(), 4 * taps[l].size(), hipMemcpyHostToDevice, c->stream));
        lv.gauss[l] = g;
        lv.mag[l] = d_mag;
        lv.orient[l] = d_ori;
        lv.kern[l] = d_taps + toff;
        lv.kn[l] = (int)taps[l].size();
        toff += align_up(taps[l].size(), 64);
    }
    const auto k_orient = c->f32_fused ? k_orient_keypoints<true> : k_orient_keypoints<false>;
    TRY(raise_dyn_lds(c, reinterpret_cast<const void*>(k_orient)));
    {
        TimedScope ts(c, "k_orient_keypoints");
        hipLaunchKernelGGL(k_orient, dim3((unsigned)n), dim3(256), lds, c->stream, d_kps, (int)n, lv, pitch, rows, cols, d_masks);
    }
    HIPCHK(c, hipGetLastError());
    OrientEntries ent{d_masks, d_kps, n, d_out};
    TRY(enqueue_compaction(c, ent, n, 1, d_cws, ocap, d_n, 0));
    unsigned int total = 0;
    HIPCHK(c, hipMemcpyAsync(&total, d_n, 4, hipMemcpyDeviceToHost, c->stream));
    TRY(vslam_ctx_sync(c));
    *count = total;
    const size_t m = std::min<size_t>(total, ocap);
    if (m) HIPCHK(c, hipMemcpy(out, d_out, m * sizeof(vslam_point), hipMemcpyDeviceToHost));
    return VSLAM_OK;
}

int vslam_sift_descriptors(vslam_ctx* c, const vslam_pyramid* py, int octave, const vslam_point* kps, size_t n, float* desc,
                           uint8_t* defined) {
    TRY(bind_device(c));
    ARGCHK(c, py && py->ctx == c && (kps || n == 0) && (desc || n == 0), "SIFT: bad arguments");
    if (octave < 0 || octave >= py->layout.n_octaves) return fail(c, VSLAM_ERR_RANGE, "octave out of range");
    ARGCHK(c, n <= 0x7fffffff / 128, "SIFT: too many keypoints");
    if (n == 0) return VSLAM_OK;
    const int rows = py->layout.rows[octave], cols = py->layout.cols[octave], pitch = py->layout.pitch[octave];
    bool used[VSLAM_NUM_LEVELS] = {};
    std::vector<float2> cs(n);
    for (size_t i = 0; i < n; ++i) {
        const vslam_point& k = kps[i];
        // what the reference would throw on (vector::at(level)) or could not have produced is an error here
        if (k.level < 0 || k.level >= VSLAM_NUM_LEVELS || k.octave != octave || k.col < -SIFT_PAD || k.row < -SIFT_PAD ||
            k.col > cols + SIFT_PAD || k.row > rows + SIFT_PAD)
            return fail(c, VSLAM_ERR_RANGE, "SIFT: keypoint outside the octave's data");
        used[k.level] = true;
        vslam_cos_sin_deg((float)k.value, &cs[i].x, &cs[i].y);  // keypoint.value holds the angle in degrees (:594)
    }
    std::vector<float> taps[VSLAM_NUM_LEVELS];
    size_t tap_elems = 0;
    for (int l = 0; l < VSLAM_NUM_LEVELS; ++l) {
        if (!used[l]) continue;
        const double sigma = 1.5 * py->info.sigma[octave][l];  // Diff_of_Gauss.cpp:616
        if (!gauss_kernel_f32(gauss_ksize_f32(sigma), sigma, taps[l])) return fail(c, VSLAM_ERR_INVALID, "SIFT: bad blur kernel");
        tap_elems += align_up(taps[l].size(), 64);
    }
    TRY(ws_reserve(c, ws_need(sizeof(vslam_point) * n) + ws_need(sizeof(float2) * n) + ws_need(sizeof(float) * 128 * n) + ws_need(n) +
                          ws_need(4 * tap_elems)));
    vslam_point* d_kps = ws_take<vslam_point>(c, n);
    float2* d_cs = ws_take<float2>(c, n);
    float* d_desc = ws_take<float>(c, 128 * n);
    uint8_t* d_def = ws_take<uint8_t>(c, n);
    float* d_taps = ws_take<float>(c, tap_elems);
    if (!d_kps || !d_cs || !d_desc || !d_def || !d_taps) return fail(c, VSLAM_ERR_NOMEM, "workspace sizing error (sift)");
    HIPCHK(c, hipMemcpyAsync(d_kps, kps, sizeof(vslam_point) * n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_cs, cs.data(), sizeof(float2) * n, hipMemcpyHostToDevice, c->stream));
    SiftLevels lv{};
    size_t toff = 0;
    for (int l = 0; l < VSLAM_NUM_LEVELS; ++l) {
        if (!used[l]) continue;
        HIPCHK(c, hipMemcpyAsync(d_taps + toff, taps[l].data(), 4 * taps[l].size(), hipMemcpyHostToDevice, c->stream));
        lv.gauss[l] = py->d_block + py->layout.octave_offset[octave] + (size_t)l * rows * pitch;
        lv.kern[l] = d_taps + toff;
        lv.kn[l] = (int)taps[l].size();
        toff += align_up(taps[l].size(), 64);
    }
    if (c->f32_fused)
        LAUNCH(c, "k_sift_descriptors", k_sift_descriptors<true>, dim3((unsigned)n), dim3(256), d_kps, d_cs, (int)n, lv, pitch, rows, cols, d_desc, d_def);
    else
        LAUNCH(c, "k_sift_descriptors", k_sift_descriptors<false>, dim3((unsigned)n), dim3(256), d_kps, d_cs, (int)n, lv, pitch, rows, cols, d_desc, d_def);
    std::vector<uint8_t> h_def(n);
    HIPCHK(c, hipMemcpyAsync(desc, d_desc, sizeof(float) * 128 * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_def.data(), d_def, n, hipMemcpyDeviceToHost, c->stream));
    TRY(vslam_ctx_sync(c));
    size_t undefined = 0;
    for (size_t i = 0; i < n; ++i) undefined += h_def[i] == 0;
    if (defined) std::memcpy(defined, h_def.data(), n);
    if (undefined && !defined)
        return fail(c, VSLAM_ERR_RANGE, "SIFT: " + std::to_string(undefined) + " keypoint window(s) leave the padded level (undefined in the reference)");
    return VSLAM_OK;
}

static int gradient_windows(vslam_ctx* c, const float* gx_windows, const float* gy_windows, int window_elems, size_t n, float* sums,
                            float* response, const char* what) {
    TRY(bind_device(c));
    ARGCHK(c, window_elems >= 0 && ((gx_windows && gy_windows) || window_elems == 0 || n == 0) && (sums || response || n == 0), what);
    ARGCHK(c, n <= 0x7fffffff / 3, what);
    if (n == 0) return VSLAM_OK;
    const size_t we = (size_t)window_elems * n;
    TRY(ws_reserve(c, 2 * ws_need(4 * we + 4) + ws_need(12 * n) + ws_need(4 * n)));
    float* d_gx = ws_take<float>(c, we + 1);
    float* d_gy = ws_take<float>(c, we + 1);
    float* d_s = ws_take<float>(c, 3 * n);
    float* d_r = ws_take<float>(c, n);
    if (we) {
        HIPCHK(c, hipMemcpyAsync(d_gx, gx_windows, 4 * we, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(d_gy, gy_windows, 4 * we, hipMemcpyHostToDevice, c->stream));
    }
    LAUNCH(c, "k_edge_response_windows", k_edge_response_windows, dim3((unsigned)((n + 255) / 256)), dim3(256), d_gx, d_gy, window_elems, (int)n,
           sums ? d_s : (float*)nullptr, response ? d_r : (float*)nullptr);
    if (sums) HIPCHK(c, hipMemcpyAsync(sums, d_s, 12 * n, hipMemcpyDeviceToHost, c->stream));
    if (response) HIPCHK(c, hipMemcpyAsync(response, d_r, 4 * n, hipMemcpyDeviceToHost, c->stream));
    return vslam_ctx_sync(c);
}

int vslam_edge_response_windows(vslam_ctx* c, const float* gx_windows, const float* gy_windows, int window_elems, size_t n,
                                float* response) {
    return gradient_windows(c, gx_windows, gy_windows, window_elems, n, nullptr, response, "computeEdgeResponse: bad arguments");
}

int vslam_structure_matrix_windows(vslam_ctx* c, const float* gx_windows, const float* gy_windows, int window_elems, size_t n,
                                   float* sums) {
    return gradient_windows(c, gx_windows, gy_windows, window_elems, n, sums, nullptr, "StructureMatrix: bad arguments");
}

// ------------------------------------------------------------- device-resident batched path

// Device copy of getGaussianKernel(n, sigma, CV_32F) for the orientation blur, cached per context.
static int get_orient_taps(vslam_ctx* c, double sigma, const float** out, int* n_out) {
    const int n = gauss_ksize_f32(sigma);
    uint64_t sb;
    std::memcpy(&sb, &sigma, 8);
    auto key = std::make_pair(sb, n);
    auto it = c->orient_taps.find(key);
    if (it == c->orient_taps.end()) {
        NO_TABLE_IN_CAPTURE(c, "filterKeypoints");
        std::vector<float> t;
        if (!gauss_kernel_f32(n, sigma, t)) return fail(c, VSLAM_ERR_INVALID, "filterKeypoints: bad blur kernel");
        // behind the taps: the zero-padded row / column forms k_orient_survivors_pk reads through scalar loads
        t.resize((size_t)orient_taps_pk_floats(n), 0.0f);
        for (int i = 0; i < n; ++i) t[(size_t)orient_taps_row_off(n) + 3 + i] = t[(size_t)i];
        for (int i = 1; i <= n / 2; ++i) t[(size_t)orient_taps_col_off(n) + i - 1] = t[(size_t)(n / 2 + i)];
        float* d = nullptr;
        HIPCHK(c, hipMalloc((void**)&d, 4 * t.size()));
        HIPCHK(c, hipMemcpy(d, t.data(), 4 * t.size(), hipMemcpyHostToDevice));
        it = c->orient_taps.emplace(key, d).first;
    }
    *out = it->second;
    *n_out = n;
    return VSLAM_OK;
}

struct OrientScratch {
    unsigned long long* flags = nullptr;  // [nf][fwords] edge-test ballots
    unsigned int* surv = nullptr;         // [nf][scap] surviving record indices
    unsigned int* scounts = nullptr;      // [nf]
    unsigned long long* masks = nullptr;  // [nf][scap] histogram-peak masks
    unsigned int* cws = nullptr;          // compaction scratch
    unsigned int* obegin = nullptr;       // [nf] list length after octave 0 (the early edge-test launch covers [0, obegin))
    unsigned int* ranges = nullptr;       // [nf][OR_RANGE_STRIDE] first survivor of each octave, of each (octave, level) (k_survivor_ranges)
    bool early_done = false;              // the early launch has been enqueued for this chunk
    bool early_forked = false;            // ... on another stream: ev_edge marks its end
    size_t fwords = 0;
};
static size_t orient_scratch_bytes(const vslam_params& p, int nf) {
    const size_t fwords = ((size_t)p.dog_cap + 63) / 64, scap = p.oriented_cap;
    return ws_need((size_t)nf * fwords * 8) + ws_need((size_t)nf * scap * 4) + 2 * ws_need((size_t)nf * 4) + ws_need((size_t)nf * scap * 8) +
           ws_need((size_t)nf * OR_RANGE_STRIDE * 4) +
           ws_need(4 * compaction_ws_elems(std::max(fwords, scap), nf));
}
static int orient_scratch_take(vslam_ctx* c, const vslam_params& p, int nf, OrientScratch& s) {
    s.fwords = ((size_t)p.dog_cap + 63) / 64;
    const size_t scap = p.oriented_cap;
    s.flags = ws_take<unsigned long long>(c, (size_t)nf * s.fwords);
    s.surv = ws_take<unsigned int>(c, (size_t)nf * scap);
    s.scounts = ws_take<unsigned int>(c, nf);
    s.obegin = ws_take<unsigned int>(c, nf);
    s.ranges = ws_take<unsigned int>(c, (size_t)nf * OR_RANGE_STRIDE);
    s.early_done = false;
    s.masks = ws_take<unsigned long long>(c, (size_t)nf * scap);
    s.cws = ws_take<unsigned int>(c, compaction_ws_elems(std::max(s.fwords, scap), nf));
    if (!s.flags || !s.surv || !s.scounts || !s.obegin || !s.ranges || !s.masks || !s.cws) return fail(c, VSLAM_ERR_NOMEM, "workspace sizing error (orient)");
    return VSLAM_OK;
}

// Geometry / blur taps of the batched filterKeypoints and the LDS budget of each octave's launch.
struct OrientPlan {
    OrientBatchGeom g;
    int need[VSLAM_MAX_OCTAVES] = {};  // LDS floats of octave o's launch of k_orient_survivors
    int smax[VSLAM_MAX_OCTAVES] = {};  // largest span (16 + 2 R) of the octave's levels: <= OR_PK_MAX_SPAN -> k_orient_survivors_pk
};
static int make_orient_plan(vslam_ctx* c, const vslam_params& p, const vslam_batch_layout& L, OrientPlan& pl) {
    OrientBatchGeom& g = pl.g;
    std::memset(&g, 0, sizeof(g));
    g.n_oct = L.n_octaves;
    constexpr int kBigLds = 36 * 1024;  // floats: 144 KB
    for (int o = 0; o < L.n_octaves; ++o) {
        g.rows[o] = L.rows[o];
        g.cols[o] = L.cols[o];
        g.pitch[o] = L.pitch[o];
        g.oct_off[o] = L.octave_offset[o];
        int worst = 0, strip = 0;
        for (int l = 1; l <= 3; ++l) {  // the levels initialKeypointDetection produces (Diff_of_Gauss.cpp:264)
            TRY(get_orient_taps(c, 1.5 * sigma_at(p.sigma0, o, l), &g.kern[o][l], &g.kn[o][l]));  // :346
            const int span = OR_WIN + 2 * (g.kn[o][l] / 2);
            // maps + taps + strip + region + the u8 patch of interior survivors (k_orient_survivors)
            worst = std::max(worst, 3 * span + span * OR_WIN + span * span + (span + 2) * ((span + 8) >> 2));
            strip = std::max(strip, span * (OR_WIN + 3));
            pl.smax[o] = std::max(pl.smax[o], span);
        }
        // regions that exceed even the big budget are read tap by tap; the strip, the maps and the taps still need room
        pl.need[o] = worst <= kBigLds ? worst : std::max(strip, std::min(worst, kBigLds));
    }
    return VSLAM_OK;
}

// The edge test (Diff_of_Gauss.cpp:331-335) for the records octave 0 has appended.  Called on the stream
// the list is written on, right behind that octave's compaction (`counts` is then the list length after
// octave 0); the kernel itself goes to `other` (the Harris chain's stream, idle by then) so that it runs
// beside the next octaves' scans instead of between them.
static int enqueue_edge_flags_early(vslam_ctx* c, const vslam_params& p, const OrientPlan& pl, int nf, const uint8_t* pyr, size_t pframe,
                                    const vslam_point* points, const unsigned int* counts, OrientScratch& s, hipStream_t other) {
    HIPCHK(c, hipMemcpyAsync(s.obegin, counts, sizeof(unsigned int) * (size_t)nf, hipMemcpyDeviceToDevice, c->stream));
    const bool fork = other && other != c->stream;
    if (fork) {
        HIPCHK(c, hipEventRecord(c->ev_list0, c->stream));
        HIPCHK(c, hipStreamWaitEvent(other, c->ev_list0, 0));
    }
    {
        StreamSwap sw(c, fork ? other : c->stream);
        LAUNCH(c, "k_edge_flags", k_edge_flags, dim3((unsigned)((s.fwords * 64 + 255) / 256), nf), dim3(256), points, (const unsigned int*)nullptr,
               (const unsigned int*)s.obegin, p.dog_cap, pyr, pframe, pl.g, s.flags, s.fwords);
        if (fork) HIPCHK(c, hipEventRecord(c->ev_edge, c->stream));
    }
    s.early_done = true;
    s.early_forked = fork;
    return VSLAM_OK;
}

// filterKeypoints for the keypoint lists of nf frames (kernels_orient_batch.hip.h), on the
// context's current stream; the lists must be complete on that stream.
static int enqueue_orient_batch(vslam_ctx* c, const vslam_params& p, const vslam_batch_layout& L, const OrientPlan& pl, int nf, const uint8_t* pyr,
                                size_t pframe, const vslam_point* points, const unsigned int* counts, OrientScratch& s,
                                vslam_point* oriented, unsigned int* oriented_counts, hipStream_t side_a = nullptr, hipStream_t side_b = nullptr) {
    const OrientBatchGeom& g = pl.g;
    const size_t scap = p.oriented_cap;
    const size_t fw = s.fwords;
    if (s.early_done && s.early_forked) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_edge, 0));  // the two launches share a flag word
    LAUNCH(c, "k_edge_flags", k_edge_flags, dim3((unsigned)((fw * 64 + 255) / 256), nf), dim3(256), points,
           (const unsigned int*)(s.early_done ? s.obegin : nullptr), counts, p.dog_cap, pyr, pframe, g, s.flags, fw);
    SurvivorEntries se{s.flags, fw, s.surv};
    TRY(enqueue_compaction(c, se, fw, nf, s.cws, (unsigned int)scap, s.scounts, 0));
    LAUNCH(c, "k_survivor_ranges", k_survivor_ranges, dim3(nf), dim3(128), points, p.dog_cap, s.surv, s.scounts, (unsigned int)scap, L.n_octaves, s.ranges);
    const auto k_surv = c->f32_fused ? k_orient_survivors<true> : k_orient_survivors<false>;
    TRY(raise_dyn_lds(c, reinterpret_cast<const void*>(k_surv)));
    const int gwg = (int)std::min<long>(1024, std::max<long>(16, 8192 / nf));
    // The launches below are independent (each takes its own survivors, each writes its own mask words) and every one ends
    // on a tail of half-empty CUs: with two idle side streams at hand (the Harris chain's and the upsample's: both are
    // done long before the list is) they go out round-robin over three streams and share the chip.
    const bool spread = side_a && side_b && side_a != c->stream && side_b != c->stream && side_a != side_b;
    hipStream_t lanes[3] = {c->stream, spread ? side_a : c->stream, spread ? side_b : c->stream};
    if (spread) {
        HIPCHK(c, hipEventRecord(c->ev_or_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(side_a, c->ev_or_fork, 0));
        HIPCHK(c, hipStreamWaitEvent(side_b, c->ev_or_fork, 0));
    }
    int next = 0;
    {
        TimedScope ts(c, "k_orient_survivors");  // with the launches spread, this brackets the main stream's share only
        for (int o = 0; o < L.n_octaves; ++o) {  // one launch per octave: its own survivors, its own LDS footprint
            if (pl.smax[o] <= OR_PK_MAX_SPAN && !c->orient_scalar_form) {  // the fine octaves: packed-f32 form, two waves per survivor, one launch per level
                // 8 gwg workgroups per frame (256 at 256 frames): ~20 survivors each on a dense frame.  With 2 gwg (80 each) the
                // launch ended on a long tail of half-empty CUs: 9.0 ms of these launches per 256-frame step against 8.4
                constexpr int pk_mult = 8;
                for (int l = 1; l <= 3; ++l) {
                    auto kfn = k_orient_survivors_pk<0, false>;
                    if (c->f32_fused) {  // (vslam_ctx_set_f32_fused: the multiply-adds of the blur as v_pk_fma_f32; the generic tap count serves every level)
                        kfn = k_orient_survivors_pk<0, true>;
                    } else {
                        switch (g.kn[o][l]) {  // the default pyramid's octave 0 (sigma0 = 1.6) with its tap counts compiled in: 6.85 -> 6.5 ms per step
                            case 25: kfn = k_orient_survivors_pk<25, false>; break;
                            case 31: kfn = k_orient_survivors_pk<31, false>; break;
                            case 39: kfn = k_orient_survivors_pk<39, false>; break;
                            default: break;
                        }
                    }
                    hipLaunchKernelGGL(kfn, dim3(pk_mult * gwg, nf), dim3(128), (size_t)orient_pk_lds_floats(OR_WIN + 2 * (g.kn[o][l] / 2)) * 4,
                                       lanes[next++ % 3], points, p.dog_cap, s.surv, s.ranges, (unsigned int)scap, pyr, pframe, g, o, l, s.masks);
                }
                continue;
            }
            hipLaunchKernelGGL(k_surv, dim3(gwg, nf), dim3(256), (size_t)pl.need[o] * 4, lanes[next++ % 3], points, p.dog_cap, s.surv, s.ranges,
                               (unsigned int)scap, pyr, pframe, g, pl.need[o], o, s.masks);
        }
    }
    HIPCHK(c, hipGetLastError());
    if (spread) {
        HIPCHK(c, hipEventRecord(c->ev_or_join[0], side_a));
        HIPCHK(c, hipEventRecord(c->ev_or_join[1], side_b));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_or_join[0], 0));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_or_join[1], 0));
    }
    OrientBatchEntries oe{s.masks, s.surv, s.scounts, (unsigned int)scap, points, p.dog_cap, oriented};
    TRY(enqueue_compaction(c, oe, scap, nf, s.cws, (unsigned int)scap, oriented_counts, 0));
    return VSLAM_OK;
}

// SIFT() for the oriented lists of nf frames (kernels_sift.hip.h), on the context's current stream,
// behind the kernels that produced the lists.
static int enqueue_sift_batch(vslam_ctx* c, const vslam_params& p, const vslam_batch_layout& L, int nf, const uint8_t* pyr, size_t pframe,
                              const vslam_point* oriented, const unsigned int* ocounts, float* desc, uint8_t* defined) {
    SiftBatchGeom g;
    std::memset(&g, 0, sizeof(g));
    for (int o = 0; o < L.n_octaves; ++o) {
        g.rows[o] = L.rows[o];
        g.cols[o] = L.cols[o];
        g.pitch[o] = L.pitch[o];
        g.oct_off[o] = L.octave_offset[o];
        for (int l = 1; l <= 3; ++l)  // the levels initialKeypointDetection produces; same taps as the orientation blur
            TRY(get_orient_taps(c, 1.5 * sigma_at(p.sigma0, o, l), &g.kern[o][l], &g.kn[o][l]));  // Diff_of_Gauss.cpp:616
    }
    for (int b = 0; b < 36; ++b) vslam_cos_sin_deg((float)(10 * b), &g.cs36[b].x, &g.cs36[b].y);  // host libm, like the reference
    // 8 x 32 workgroups per frame at 256 frames (~120 points each on a dense frame): with 32 the launch ended on a tail of
    // half-empty CUs (11.3 ms per 256-frame step against 10.1)
    const int gwg = 8 * (int)std::min<long>(1024, std::max<long>(16, 8192 / nf));
    if (c->f32_fused)
        LAUNCH(c, "k_sift_descriptors", k_sift_descriptors_batch<true>, dim3(gwg, nf), dim3(256), oriented, ocounts, p.oriented_cap, pyr, pframe, g, desc, defined);
    else
        LAUNCH(c, "k_sift_descriptors", k_sift_descriptors_batch<false>, dim3(gwg, nf), dim3(256), oriented, ocounts, p.oriented_cap, pyr, pframe, g, desc, defined);
    return VSLAM_OK;
}

int vslam_detect_batch_dev(vslam_ctx* c, const vslam_params* pp, const uint8_t* d_frames, size_t frame_stride,
                           int n_frames, const vslam_batch_out* out) {
    TRY(bind_device(c));
    ARGCHK(c, pp && d_frames && out && n_frames > 0, "detect_batch: bad arguments");
    const vslam_params p = *pp;
    ARGCHK(c, p.rows > 0 && p.cols > 0 && frame_stride >= (size_t)p.rows * p.cols, "detect_batch: bad frame geometry");
    vslam_batch_layout L;
    if (make_layout(&p, &L) != VSLAM_OK) return fail(c, VSLAM_ERR_INVALID, "detect_batch: bad parameters");
    const bool dog = p.n_octaves > 0;
    const bool want_kps = out->harris_kps && out->harris_counts;
    const bool harris = p.do_harris && (out->response || out->nms_mask || out->nms2 || want_kps);
    ARGCHK(c, !dog || out->pyramid, "detect_batch: the DoG path needs out->pyramid");
    {   // every buffer against the size the caller states for it: nothing is launched on an undersized buffer
        ARGCHK(c, out->struct_size == sizeof(vslam_batch_out),
               "detect_batch: out->struct_size is not sizeof(vslam_batch_out) (set it and every x_bytes; vslam_batch_out_required gives the numbers)");
        vslam_batch_out need{};
        if (vslam_batch_out_required(&p, n_frames, &need) != VSLAM_OK) return fail(c, VSLAM_ERR_INVALID, "detect_batch: bad parameters");
#define VSLAM_SIZECHK(field)                                                                                               \
    if (out->field && out->field##_bytes < need.field##_bytes)                                                             \
        return fail(c, VSLAM_ERR_INVALID, std::string("detect_batch: out->" #field " holds ") + std::to_string(out->field##_bytes) + \
                                              " bytes, " + std::to_string(n_frames) + " frames need " + std::to_string(need.field##_bytes))
        VSLAM_SIZECHK(response);
        VSLAM_SIZECHK(nms_mask);
        VSLAM_SIZECHK(nms2);
        VSLAM_SIZECHK(harris_kps);
        VSLAM_SIZECHK(harris_counts);
        VSLAM_SIZECHK(pyramid);
        VSLAM_SIZECHK(extrema_bits);
        VSLAM_SIZECHK(dog_points);
        VSLAM_SIZECHK(dog_counts);
        VSLAM_SIZECHK(oriented_points);
        VSLAM_SIZECHK(oriented_counts);
        VSLAM_SIZECHK(oriented_survivors);
        VSLAM_SIZECHK(descriptors);
        VSLAM_SIZECHK(descriptor_defined);
#undef VSLAM_SIZECHK
    }
    const bool orient = dog && p.orient;
    ARGCHK(c, !orient || (p.localize && out->dog_points && out->dog_counts && out->oriented_points && out->oriented_counts &&
                          p.oriented_cap > 0 && p.dog_cap > 0 && p.extrema_window == 3),
           "detect_batch: orient needs localize = 1, windowSize 3, the DoG point list and the oriented outputs");
    ARGCHK(c, !out->descriptors || orient, "detect_batch: descriptors need orient = 1");
    ARGCHK(c, !out->descriptor_defined || out->descriptors, "detect_batch: descriptor_defined without descriptors");
    const size_t N = (size_t)p.rows * p.cols;
    // Whole-batch launches: every kernel sees all frames (grid.z = frames), so even the coarse
    // octaves fill the chip.  Scratch: octave bases (+ u16 row sums of the non-tiled octaves).
    const int chunk = std::min(n_frames, 256);
    size_t need = 0;
    const bool want_sitemap = dog && c->mx && !p.localize && !p.extrema_dense && p.extrema_window == 3;
    if (dog) need += dog_scratch_bytes(L, p.sigma0, chunk, want_sitemap);
    if (harris)
        need += (out->response ? 0 : ws_need((size_t)chunk * N * 4)) + ws_need((size_t)chunk * harris_flag_words(p.rows, p.cols) * 8) +
                ws_need(4 * compaction_ws_elems(harris_flag_words(p.rows, p.cols), chunk));
    if (orient) need += orient_scratch_bytes(p, chunk);
    c->phase_marked = false;
    ++c->batch_calls;
    c->pack_pending = false;  // the previous call joined its side streams back
    TRY(ws_reserve(c, need));
    DogScratch s;
    if (dog) TRY(dog_scratch_take(c, L, p.sigma0, chunk, s, want_sitemap));
    OrientScratch os;
    if (orient) TRY(orient_scratch_take(c, p, chunk, os));
    OrientPlan opl;
    if (orient) TRY(make_orient_plan(c, p, L, opl));
    float* resp_ws = (harris && !out->response) ? ws_take<float>(c, (size_t)chunk * N) : nullptr;
    unsigned long long* hflags = harris ? ws_take<unsigned long long>(c, (size_t)chunk * harris_flag_words(p.rows, p.cols)) : nullptr;
    unsigned int* hcws = harris ? ws_take<unsigned int>(c, compaction_ws_elems(harris_flag_words(p.rows, p.cols), chunk)) : nullptr;
    // Fork (VSLAM_AUX_STREAMS=0 disables): the Harris chain and the extrema/compaction chain run on
    // the context's auxiliary streams beside the octave kernels.  Measured on MI355X: +2.8 %
    // (11.2k vs 10.9k frames/s) -- small, because every kernel of the batch is VALU-issue-bound
    // rather than HBM-bound; per-kernel durations grow accordingly when kernels share the chip.
    static const bool use_aux = [] {
        const char* e = VSLAM_DIAG_ENV("VSLAM_AUX_STREAMS");
        return !(e && e[0] == '0');
    }();
    static const bool orient_spread = [] {
        const char* e = VSLAM_DIAG_ENV("VSLAM_ORIENT_SPREAD");
        return !(e && e[0] == '0');
    }();
    // Under stream capture (hipGraph) no two SIDE streams of the call may wait on each other's events.  The topology is legal
    // (fork from the origin stream, cross edges between the forked streams, all joined back) and every event is recorded on
    // a stream that is already part of the capture; but this HIP runtime (ROCm 7.2's libamdhip64.so.7 and the copy torch 2.10
    // bundles alike) never returns from hipStreamEndCapture then: a function that calls itself for every entry of a
    // per-stream vector (the shape of hip::Stream::EndCapture over parallelCaptureStreams_) recurses 174,000 frames deep and
    // the process dies of stack exhaustion - no HIP status is ever seen.  tools/graph_try.py reproduces it through this
    // library (each case in a child process; both ends of the stack in profiles/r05_graph_try.json), and
    // tools/capture_cycle_repro.hip WITHOUT it: two side streams that wait on each other's events are harmless by themselves
    // (modes 2-4) and fatal as soon as each also waits on an origin-stream event again in between (mode 5) - which this
    // library's side streams do on every octave's event.  The orientation stage has two such pairs - the early edge test
    // (list stream -> Harris stream -> back) and the spread launches (list stream -> two idle side streams -> back); each
    // alone reproduces the crash.  While a capture is on, both stay on the list stream; everything else forks from and joins
    // to the main (origin) stream.
    bool capturing = false, cap_early = false, cap_spread = false;  // the nested forks stay out of a capture
    {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(c->stream, &cap) != hipSuccess) (void)hipGetLastError();
        capturing = cap != hipStreamCaptureStatusNone;
        // diagnostic switch for tools/graph_try.py (the reproducer of that fault): leave the nested forks in while capturing
        // (1: both nested forks stay in, 2: only the early edge test's, 3: only the spread orientation launches')
        static const int nested_in_capture = [] {
            const char* e = VSLAM_DIAG_ENV("VSLAM_CAPTURE_NESTED_FORKS");
            return e ? std::atoi(e) : 0;
        }();
        cap_early = capturing && !(nested_in_capture == 1 || nested_in_capture == 2);
        cap_spread = capturing && !(nested_in_capture == 1 || nested_in_capture == 3);
    }
    hipStream_t sh = c->stream, sx = nullptr;  // Harris stream, extrema stream (nullptr = main)
    // Any early return between the fork and the join must not leave the side streams running into
    // buffers the caller (or the next ws_reserve) is about to reuse: drain them on the error path.
    struct ForkGuard {
        vslam_ctx* c;
        bool armed = false;
        ~ForkGuard() {
            if (!armed) return;
            for (int i = 0; i < vslam_ctx::kAux; ++i)
                if (c->aux[i]) (void)hipStreamSynchronize(c->aux[i]);
            (void)hipStreamSynchronize(c->stream);
        }
    } guard{c};
    // the shape of this call (the stream tuner and the join watchdog compare calls of one shape only)
    const unsigned long long call_key = ((unsigned long long)(unsigned)n_frames << 40) ^ ((unsigned long long)(unsigned)p.rows << 20) ^ (unsigned)p.cols ^
                                        ((unsigned long long)(p.localize + 2 * p.orient + 4 * p.extrema_dense + 8 * (out->descriptors != nullptr)) << 60) ^
                                        ((unsigned long long)(unsigned)p.n_octaves << 56) ^ ((unsigned long long)(c->mx ? 1 : 0) << 39);
    bool side_streams = use_aux;
    if (use_aux) {
        TRY(sched_ensure_aux(c));
        // (calls of a few megapixels are dominated by launch latencies: their lag says nothing about starvation)
        TRY(sched_watch_before_call(c, call_key, dog && harris && n_frames >= 32 && (size_t)n_frames * N >= ((size_t)16 << 20), capturing));
        if (c->watch.level == 2) side_streams = false;  // the watchdog's last step: everything on the caller's stream
    }
    if (side_streams) {
        // the side-stream pair of this call (StreamTuner): only full-size batches with both paths are compared
        TRY(sched_tuner_before_call(c, call_key, dog && harris && n_frames >= 32));
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        for (int i = 0; i < vslam_ctx::kAux; ++i) HIPCHK(c, hipStreamWaitEvent(c->aux[i], c->ev_fork, 0));
        sh = c->aux[0];
        sx = c->aux[1];
        guard.armed = true;
    }
    for (int f0 = 0; f0 < n_frames; f0 += chunk) {
        const int nf = std::min(chunk, n_frames - f0);
        const uint8_t* fr = d_frames + (size_t)f0 * frame_stride;
        auto do_harris = [&]() -> int {
            StreamSwap sw(c, sh);
            float* resp = out->response ? out->response + (size_t)f0 * N : resp_ws;
            TRY(enqueue_harris(c, fr, frame_stride, p.rows, p.cols, nf, p.harris_k, resp,
                               out->nms_mask ? out->nms_mask + (size_t)f0 * N : nullptr,
                               out->nms2 ? out->nms2 + (size_t)f0 * N : nullptr, want_kps ? hflags : nullptr,
                               want_kps ? out->harris_kps + (size_t)f0 * p.harris_cap : nullptr, p.harris_cap,
                               want_kps ? out->harris_counts + f0 : nullptr, hcws));
            return VSLAM_OK;
        };
        // The Harris chain: without the DoG path or the side streams it is simply enqueued here.  With
        // them it goes to its own stream - at once for small batches, behind the last LDS-tiled octave
        // kernel for large ones (enqueue_dog explains the gate) - and is enqueued from inside enqueue_dog,
        // right after that octave's event, so that nothing enqueued on its stream later can get in front of it.
        // (matrix path: starting the Harris chain at once, or behind octave 0 / 2 / 3 instead of 1, moved the step by less than
        // +-1.5 % - 14.54 .. 14.90 ms on one box - so the gate stays where the default path has it)
        const int harris_gate = (dog && side_streams) ? dog_side_gate(p, L, nf) : -1;
        if (harris && harris_gate < 0) TRY(do_harris());
        if (dog) {
            const bool ext = out->extrema_bits || (out->dog_points && out->dog_counts);
            os.early_done = os.early_forked = false;
            // filterKeypoints' edge test for octave 0's records as soon as that octave's part of the list exists
            const std::function<int(int)> after_list = [&](int o) -> int {
                if (orient && o == 0 && L.n_octaves > 1)
                    return enqueue_edge_flags_early(c, p, opl, nf, out->pyramid + (size_t)f0 * L.pyramid_frame_bytes, L.pyramid_frame_bytes,
                                                    out->dog_points + (size_t)f0 * p.dog_cap, out->dog_counts + f0, os, (side_streams && !cap_early) ? sh : nullptr);
                return VSLAM_OK;
            };
            const std::function<int(int)> after_octave = [&](int o) -> int {
                if (harris && o == harris_gate) {
                    HIPCHK(c, hipStreamWaitEvent(sh, c->ev_oct[o], 0));
                    return do_harris();
                }
                return VSLAM_OK;
            };
            TRY(enqueue_dog(c, p, L, fr, p.cols, frame_stride, nf, out->pyramid + (size_t)f0 * L.pyramid_frame_bytes,
                            L.pyramid_frame_bytes, s,
                            out->extrema_bits ? (unsigned long long*)out->extrema_bits + (size_t)f0 * L.bits_frame_words : nullptr,
                            ext, out->dog_points ? out->dog_points + (size_t)f0 * p.dog_cap : nullptr,
                            out->dog_counts ? out->dog_counts + f0 : nullptr, sx, after_list, after_octave,
                            side_streams ? c->aux[2] : nullptr, f0 > 0, /*bases_are_scratch=*/true));
            if (orient) {  // filterKeypoints behind the list, on the stream that produced it
                StreamSwap sw(c, sx ? sx : c->stream);
                TRY(enqueue_orient_batch(c, p, L, opl, nf, out->pyramid + (size_t)f0 * L.pyramid_frame_bytes, L.pyramid_frame_bytes,
                                         out->dog_points + (size_t)f0 * p.dog_cap, out->dog_counts + f0, os,
                                         out->oriented_points + (size_t)f0 * p.oriented_cap, out->oriented_counts + f0,
                                         (side_streams && orient_spread && !cap_spread) ? sh : nullptr,
                                         (side_streams && orient_spread && !cap_spread) ? c->aux[2] : nullptr));
                if (out->oriented_survivors)
                    HIPCHK(c, hipMemcpyAsync(out->oriented_survivors + f0, os.scounts, sizeof(unsigned int) * (size_t)nf,
                                             hipMemcpyDeviceToDevice, c->stream));
                if (out->descriptors)
                    TRY(enqueue_sift_batch(c, p, L, nf, out->pyramid + (size_t)f0 * L.pyramid_frame_bytes, L.pyramid_frame_bytes,
                                           out->oriented_points + (size_t)f0 * p.oriented_cap, out->oriented_counts + f0,
                                           out->descriptors + (size_t)f0 * p.oriented_cap * 128,
                                           out->descriptor_defined ? out->descriptor_defined + (size_t)f0 * p.oriented_cap : nullptr));
            }
        }
    }
    if (c->watch.recording >= 0) HIPCHK(c, hipEventRecord(c->watch.tm[c->watch.recording], c->stream));  // the main stream's own work ends here
    if (side_streams)  // join
        for (int i = 0; i < vslam_ctx::kAux; ++i) {
            HIPCHK(c, hipEventRecord(c->ev_join[i], c->aux[i]));
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join[i], 0));
        }
    if (c->watch.recording >= 0) {
        HIPCHK(c, hipEventRecord(c->watch.t1[c->watch.recording], c->stream));
        c->watch.live[c->watch.recording] = true;
        c->watch.recording = -1;
    }
    if (!c->phase_marked) TRY(mark_phase(c));  // no DoG path in this call: its end is the mark
    if (side_streams) TRY(sched_tuner_after_call(c));
    guard.armed = false;
    return VSLAM_OK;
}


int vslam_ctx_follow(vslam_ctx* c, const vslam_ctx* leader) {
    TRY(bind_device(c));
    ARGCHK(c, leader && leader != c && leader->device == c->device, "ctx_follow: the leader must be another context on the same device");
    if (leader->ev_phase) HIPCHK(c, hipStreamWaitEvent(c->stream, leader->ev_phase, 0));  // no batch call yet: nothing to wait for
    return VSLAM_OK;
}

int vslam_detect_batch_host(vslam_ctx* c, const vslam_params* pp, const uint8_t* frames, size_t frame_stride, int n_frames,
                            const vslam_host_lists* out) {
    TRY(bind_device(c));
    ARGCHK(c, pp && frames && out && n_frames > 0 && n_frames <= 65535, "detect_batch_host: bad arguments");
    ARGCHK(c, out->struct_size == sizeof(vslam_host_lists), "detect_batch_host: out->struct_size is not sizeof(vslam_host_lists)");
    vslam_params p = *pp;
    ARGCHK(c, p.rows > 0 && p.cols > 0 && frame_stride >= (size_t)p.rows * p.cols, "detect_batch_host: bad frame geometry");
    ARGCHK(c, !p.orient && !p.extrema_dense, "detect_batch_host: the Harris and DoG lists only (orient / extrema_dense: vslam_detect_batch_dev)");
    const bool want_h = out->harris || out->harris_offsets || out->harris_counts, want_d = out->dog || out->dog_offsets || out->dog_counts;
    ARGCHK(c, !want_h || (out->harris_offsets && out->harris_counts && (out->harris || out->harris_bytes == 0)), "detect_batch_host: incomplete Harris list");
    ARGCHK(c, !want_d || (out->dog_offsets && out->dog_counts && (out->dog || out->dog_bytes == 0)), "detect_batch_host: incomplete DoG list");
    ARGCHK(c, want_h || want_d, "detect_batch_host: no list requested");
    p.do_harris = want_h ? 1 : 0;
    if (!want_d) p.n_octaves = 0;
    ARGCHK(c, !want_d || p.n_octaves > 0, "detect_batch_host: the DoG list needs n_octaves > 0");
    vslam_batch_out need{};
    if (vslam_batch_out_required(&p, n_frames, &need) != VSLAM_OK) return fail(c, VSLAM_ERR_INVALID, "detect_batch_host: bad parameters");
    const size_t n = (size_t)n_frames, N = (size_t)p.rows * p.cols;
    // one device block for everything this call needs (recycled through the context's block cache when small)
    const size_t hpk = want_h ? out->harris_bytes / sizeof(vslam_kp) * sizeof(vslam_kp) : 0, dpk = want_d ? out->dog_bytes / sizeof(vslam_point) * sizeof(vslam_point) : 0;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t o = off;
        off += align_up(bytes ? bytes : 1, 256);
        return o;
    };
    const size_t o_frames = carve(n * N), o_pyr = carve(want_d ? need.pyramid_bytes : 0), o_hk = carve(want_h ? need.harris_kps_bytes : 0),
                 o_hc = carve(n * 4), o_dp = carve(want_d ? need.dog_points_bytes : 0), o_dc = carve(n * 4), o_hp = carve(hpk), o_dpk = carve(dpk),
                 o_off = carve(2 * (n + 1) * 8);
    size_t cap = 0;
    char* blk = (char*)block_alloc(c, off, &cap);
    if (!blk) return fail(c, VSLAM_ERR_NOMEM, "detect_batch_host: device allocation failed");
    struct Release {
        vslam_ctx* c;
        void* p;
        size_t cap;
        ~Release() {
            (void)hipStreamSynchronize(c->stream);
            block_release(c, p, cap);
        }
    } rel{c, blk, cap};
    uint8_t* d_frames = (uint8_t*)(blk + o_frames);
    HIPCHK(c, hipMemsetAsync(blk + o_hc, 0, n * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(blk + o_dc, 0, n * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(blk + o_off, 0, 2 * (n + 1) * 8, c->stream));
    TRY(h2d(c, d_frames, N, frames, frame_stride, N, n));
    vslam_batch_out bo{};
    bo.struct_size = sizeof(bo);
    if (want_h) {
        bo.harris_kps = (vslam_kp*)(blk + o_hk), bo.harris_kps_bytes = need.harris_kps_bytes;
        bo.harris_counts = (uint32_t*)(blk + o_hc), bo.harris_counts_bytes = need.harris_counts_bytes;
    }
    if (want_d) {
        bo.pyramid = (uint8_t*)(blk + o_pyr), bo.pyramid_bytes = need.pyramid_bytes;
        bo.dog_points = (vslam_point*)(blk + o_dp), bo.dog_points_bytes = need.dog_points_bytes;
        bo.dog_counts = (uint32_t*)(blk + o_dc), bo.dog_counts_bytes = need.dog_counts_bytes;
    }
    TRY(vslam_detect_batch_dev(c, &p, d_frames, N, n_frames, &bo));
    uint64_t* d_off = (uint64_t*)(blk + o_off);
    if (want_h) TRY(vslam_pack_lists_dev(c, bo.harris_kps, sizeof(vslam_kp), p.harris_cap, bo.harris_counts, n_frames, blk + o_hp, hpk, d_off));
    if (want_d) TRY(vslam_pack_lists_dev(c, bo.dog_points, sizeof(vslam_point), p.dog_cap, bo.dog_counts, n_frames, blk + o_dpk, dpk, d_off + (n + 1)));
    if (want_h) {
        HIPCHK(c, hipMemcpyAsync(out->harris_offsets, d_off, (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(out->harris_counts, bo.harris_counts, n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    if (want_d) {
        HIPCHK(c, hipMemcpyAsync(out->dog_offsets, d_off + (n + 1), (n + 1) * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(out->dog_counts, bo.dog_counts, n * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));  // the offsets say how many records exist
    if (want_h) {
        const size_t bytes = std::min<size_t>(out->harris_offsets[n] * sizeof(vslam_kp), hpk);
        if (bytes) HIPCHK(c, hipMemcpyAsync(out->harris, blk + o_hp, bytes, hipMemcpyDeviceToHost, c->stream));
    }
    if (want_d) {
        const size_t bytes = std::min<size_t>(out->dog_offsets[n] * sizeof(vslam_point), dpk);
        if (bytes) HIPCHK(c, hipMemcpyAsync(out->dog, blk + o_dpk, bytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return VSLAM_OK;
}

int vslam_pack_lists_dev(vslam_ctx* c, const void* lists, size_t record_bytes, uint32_t cap, const uint32_t* counts, int n_frames,
                         void* packed, size_t packed_bytes, uint64_t* offsets) {
    TRY(bind_device(c));
    ARGCHK(c, lists && counts && offsets && n_frames > 0 && n_frames <= 65535 && cap > 0 && (packed || packed_bytes == 0),
           "pack_lists: bad arguments (1 .. 65535 frames per call)");
    ARGCHK(c, record_bytes >= 4 && record_bytes % 4 == 0 && record_bytes <= 4096, "pack_lists: record_bytes must be a multiple of 4");
    const unsigned int rec_dw = (unsigned int)(record_bytes / 4);
    LAUNCH(c, "k_pack_offsets", k_pack_offsets, dim3(1), dim3(256), counts, cap, n_frames, (unsigned long long*)offsets);
    const unsigned long long frame_dw = (unsigned long long)cap * rec_dw;
    const unsigned int gx = (unsigned int)std::min<unsigned long long>((frame_dw + 2047) / 2048, 96);
    if (packed_bytes >= 4)
        LAUNCH(c, "k_pack_copy", k_pack_copy, dim3(gx, 1, n_frames), dim3(256), (const unsigned int*)lists, rec_dw, cap, counts,
               (const unsigned long long*)offsets, (unsigned int*)packed, (unsigned long long)(packed_bytes / 4));
    return VSLAM_OK;
}

int vslam_pack_points16_dev(vslam_ctx* c, const vslam_point* lists, uint32_t cap, const uint32_t* counts, int n_frames, vslam_point16* packed,
                            size_t packed_bytes, uint64_t* offsets) {
    TRY(bind_device(c));
    ARGCHK(c, lists && counts && offsets && n_frames > 0 && n_frames <= 65535 && cap > 0 && (packed || packed_bytes == 0),
           "pack_points16: bad arguments (1 .. 65535 frames per call)");
    ARGCHK(c, (reinterpret_cast<uintptr_t>(lists) & 7) == 0 && (reinterpret_cast<uintptr_t>(packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(offsets) & 7) == 0,
           "pack_points16: lists must be 8-byte aligned, packed 16-byte aligned (the records move as 8- and 16-byte words)");
    static_assert(sizeof(vslam_point) == 24 && sizeof(vslam_point16) == 16, "record layouts");
    LAUNCH(c, "k_pack_offsets", k_pack_offsets, dim3(1), dim3(256), counts, cap, n_frames, (unsigned long long*)offsets);
    const unsigned int gx = (unsigned int)std::min<unsigned long long>(((unsigned long long)cap + 1023) / 1024, 96);
    if (packed_bytes >= sizeof(vslam_point16))
        LAUNCH(c, "k_pack_copy", k_pack_points16, dim3(gx, 1, n_frames), dim3(256), reinterpret_cast<const int2*>(lists), cap, counts,
               (const unsigned long long*)offsets, reinterpret_cast<uint4*>(packed), (unsigned long long)(packed_bytes / sizeof(vslam_point16)));
    return VSLAM_OK;
}

int vslam_count_totals_dev(vslam_ctx* c, const uint32_t* harris_counts, const uint32_t* dog_counts, int n_frames, uint64_t* totals) {
    TRY(bind_device(c));
    ARGCHK(c, totals && n_frames > 0 && (harris_counts || dog_counts), "count_totals: bad arguments");
    LAUNCH(c, "k_count_totals", k_count_totals, dim3(1), dim3(256), harris_counts, dog_counts, n_frames, (unsigned long long*)totals);
    return VSLAM_OK;
}

}  // extern "C"
