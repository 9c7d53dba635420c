// The context of the C ABI and the pieces every translation unit of the library shares: error plumbing, the bench
// timing hook's scope object, and the interface of the side-stream scheduling unit (vslam_sched.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vslam.h"

namespace vslam {
struct StripTaps;
}

// Diagnostic environment switches (A/B leftovers, the capture-fault reproducer's knobs) exist only in a build with
// -DVSLAM_DIAGNOSTICS (lib/libvslam_diag.so, `make diag`): the shipped library never reads them.
#ifdef VSLAM_DIAGNOSTICS
#define VSLAM_DIAG_ENV(name) std::getenv(name)
#else
#define VSLAM_DIAG_ENV(name) (static_cast<const char*>(nullptr))
#endif

// Which pair of low-priority side streams the batched path runs on: see "side-stream placement" below.
struct StreamTuner {
    static constexpr int K = 3;                      // candidate pairs
    static constexpr int M = K + 1;                  // measured calls: pair 0, 1, 2, 0
    hipStream_t cand[K][2] = {};                     // cand[0] = the pair ensure_aux created
    hipEvent_t t0[M] = {}, t1[M] = {};
    int measured = 0;                                // calls measured so far
    int measuring = -1;                              // slot being measured by the current call
    bool enabled = false;                            // vslam_ctx_tune_side_streams (or VSLAM_STREAM_TUNER=1 when the context was created)
    bool done = false;
    int chosen = 0;
    unsigned long long key = 0;                      // shape of the calls being compared (0: none yet)
    int calls = 0, resets = 0;
};

// Side-stream priority and the join watchdog (round 5).
// The batched path's two side streams (Harris chain; scans and lists) may run at the LOWEST stream priority, so that they
// yield to the octave kernels, or at the main stream's.  Which is faster is decided by the hardware queue each stream
// happens to land on (HIP multiplexes streams onto GPU_MAX_HW_QUEUES queues per priority level, default 4; DESIGN section
// 5.4).  Same box, C++ host, device-resident, frames/s with 2 / 3 / 4 / 6 / 12 queues: yielding 14.1 k / 11.5 k / 13.4 k /
// 14.0 k / 14.0 k, same priority 13.7 k / 13.7 k / 14.0 k / 13.7 k / 14.2 k - yielding wins 2-3 % on a lucky layout and
// loses 18 % on an unlucky one (a low-priority queue behind the main queue's barrier packet crawls), same priority never
// moves more than 3.5 %.  The default is therefore the SAME priority (level 1): a caller that embeds the library in a
// process with streams of its own gets a sane schedule with HIP's default queue count, without setting an environment
// variable or opting in to anything.  A host that owns its queue layout asks for yielding streams (level 0) with
// vslam_ctx_set_side_stream_priority / VSLAM_SIDE_PRIORITY=low (Stream's host-fed mode, which also asks for 12 queues).
//
// The watchdog keeps either choice honest.  The first full-size batch calls of a context are bracketed by three events on
// the main stream - start, "my own kernels are enqueued up to here" (just before the waits on the side streams' join
// events) and end.  t(end) - t(own) is how long the main stream sat waiting for side work: 0.4 % of an 18 ms batch when
// the side streams run freely, 5 % with yielding streams on four queues, 20 % when one of them is being starved.  A later
// call reads the events once they are complete (hipEventQuery: nothing ever waits on the host).  Three measured calls with
// a median lag above the level's limit (3 % at level 0, 10 % at level 1) start a TRIAL of the next level - same priority,
// then no side streams at all (level 2) - and the trial is kept only if its fastest call beats the previous level's fastest
// by 1 %; otherwise the context goes back.  Either way the watch ends after at most ten measured calls.  Off while a capture
// is on, while the opt-in tuner is comparing pairs, and under VSLAM_JOIN_WATCH=0; VSLAM_JOIN_WATCH_LEVEL pins a level.
// Results never depend on the level.
struct JoinWatch {
    static constexpr int RING = 4, NEED = 3;
    hipEvent_t t0[RING] = {}, tm[RING] = {}, t1[RING] = {};
    bool live[RING] = {};       // events of slot i are recorded and not yet read
    int head = 0;               // next slot to record
    int recording = -1;         // slot of the call being enqueued
    int calls = 0;              // eligible calls at the current level (the first is not measured)
    int n_meas = 0;             // measurements at the current level
    float lag[NEED] = {}, best_total = 0.0f;
    float level_best[3] = {0.0f, 0.0f, 0.0f};  // fastest measured call at each level tried
    int level = 1;              // 0: low-priority (yielding) side streams, 1: the main stream's priority, 2: no side streams
    int trial_from = -1;        // the level a running trial came from (-1: the current level is not a trial)
    unsigned long long key = 0; // shape of the calls being measured (only calls of one shape are compared)
    int restarts = 0;
    bool done = false, disabled = false, pinned = false;
    float last_lag_frac = -1.0f;
    hipStream_t pair[2][2] = {};  // the side-stream pairs of levels 0 and 1 (both live until the context goes)
};

struct vslam_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // bump workspace in HBM, grown between calls only (never inside a launch sequence)
    char* ws = nullptr;
    size_t ws_cap = 0, ws_off = 0;
    std::map<std::pair<int, uint64_t>, uint16_t*> taps;  // device copies of quantised taps
    std::map<std::pair<uint64_t, int>, vslam::StripTaps*> strip_taps;  // (sigma0 bits, octave) -> device tables
    std::map<std::pair<uint64_t, int>, void*> tile_taps;        // (sigma0 bits, octave) -> PyrTaps<CFG>
    // OPT-IN matrix-core form of the LDS-tiled octave kernels (VSLAM_MX=1 / vslam_ctx_set_matrix_path): never the default
    bool mx = false;
    // vslam_ctx_set_f32_fused / VSLAM_F32_FUSED=1: the f32 stages (separable f32 filter of filterKeypoints / SIFT, the arctangent of
    // processGradients) with fused multiply-adds, as an OpenCV that dispatches its AVX2 + FMA3 code computes them (default: every
    // product and sum rounded, OpenCV's SSE2 baseline)
    bool f32_fused = false;
    bool orient_scalar_form = false;  // VSLAM_ORIENT_SCALAR=1: k_orient_survivors for every octave (the round-3 form, kept for comparison)
    std::map<std::pair<uint64_t, int>, void*> mx_taps;          // (sigma0 bits, octave) -> MxTaps<CFG>
    // auxiliary streams of the batched path: the HBM-bound chains (Harris; extrema + compaction)
    // run beside the VALU-bound pyramid kernels; forked from / joined to `stream` by events
    // aux[2] carries only the second-half upsample of a large batch (enqueue_dog): it must not queue behind
    // the previous chunk's list chain on aux[1]
    static constexpr int kAux = 3;
    hipStream_t aux[kAux] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kAux] = {nullptr, nullptr, nullptr}, ev_oct[VSLAM_MAX_OCTAVES] = {};
    int prio_lo = 0;      // priority of the two side streams in use (0: the main stream's)
    int prio_dev_lo = 0;  // the device's lowest stream priority (0: it has no priority levels)
    StreamTuner tuner;    // which pair of side streams the batched path runs on (see StreamTuner)
    JoinWatch watch;      // steps the side streams down when their join lags (see JoinWatch)
    hipEvent_t ev_phase = nullptr;  // recorded by every vslam_detect_batch_dev call once its octave-0 kernels are enqueued (vslam_ctx_follow)
    bool phase_marked = false;
    int batch_calls = 0;  // vslam_detect_batch_dev calls so far
    hipEvent_t ev_up2 = nullptr;  // the second half of a batch has been upsampled (enqueue_dog)
    hipEvent_t ev_chunk = nullptr;  // the main-stream kernels of a chunk (the readers of the octave bases) are enqueued up to here
    // matrix path, fused lattice scan: the side stream's k_extrema_pack launches of a chunk have read the site / seam maps
    // (the one scratch of the DoG path written on the main stream and read on a side stream: the next chunk's octave
    // kernels wait for this before they overwrite it)
    hipEvent_t ev_pack = nullptr;
    bool pack_pending = false;
    hipEvent_t ev_list0 = nullptr, ev_edge = nullptr;  // octave 0's part of the DoG list is written / its edge test is done
    hipEvent_t ev_or_fork = nullptr, ev_or_join[2] = {nullptr, nullptr};  // the orientation launches spread over the idle side streams (enqueue_orient_batch)
    // recycled pyramid blocks: a GaussPyramid per image would otherwise pay hipMalloc + hipFree of
    // >100 MB each time (milliseconds, more than the kernels)
    std::vector<std::pair<size_t, void*>> block_cache;
    // kernels whose dynamic-LDS ceiling has been raised on this device (once, not per launch)
    std::set<const void*> lds_raised;
    float* loc_lut = nullptr;  // FeaturePointLocalization table (kernels_localize.hip.h), built on first use
    uint8_t* dump = nullptr;   // 256 bytes nobody reads: where the Harris kernel's margin lanes store in its steady rows
    std::map<std::pair<uint64_t, int>, float*> orient_taps;  // (sigma bits, kernel width) -> f32 Gaussian taps on the device
    // bench timing hook
    std::string timing_name;
    int launch_tag = -1;  // octave of the launch being enqueued, for helpers that do not get it as an argument
    int timing_tag = -1;  // "name@N": only launches tagged N (the octave)
    std::deque<std::pair<hipEvent_t, hipEvent_t>> timing_ev;  // (a deque: a TimedScope keeps a pointer to its slot while later scopes append)
    size_t timing_used = 0;
};

static inline int fail(vslam_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(ctx, expr)                                                                        \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(ctx, e_ == hipErrorOutOfMemory ? VSLAM_ERR_NOMEM : VSLAM_ERR_HIP,        \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                     \
    } while (0)

#define ARGCHK(ctx, cond, msg) \
    if (!(cond)) return fail(ctx, VSLAM_ERR_INVALID, msg)
#define TRY(expr)              \
    do {                       \
        int rc_ = (expr);      \
        if (rc_) return rc_;   \
    } while (0)

static inline int bind_device(vslam_ctx* c) {
    if (!c) return VSLAM_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    return VSLAM_OK;
}

// ---- vslam_sched.cpp: side-stream placement, the join watchdog, the bench timing hook -----------------------------
namespace vslam {
void sched_init_from_env(vslam_ctx* c);   // VSLAM_JOIN_WATCH, VSLAM_SIDE_PRIORITY, VSLAM_STREAM_TUNER (vslam_ctx_create)
void sched_destroy(vslam_ctx* c);         // side streams, their events, the hook's events (vslam_ctx_destroy)
int sched_ensure_aux(vslam_ctx* c);       // creates the side streams and the fork / join events on the first batch call
// Before the fork of a batch call: read finished measurements, move between the levels, start this call's measurement.
int sched_watch_before_call(vslam_ctx* c, unsigned long long key, bool eligible, bool capturing);
// The opt-in tuner: picks the pair of (yielding) side streams this call uses / closes the call's measurement.
int sched_tuner_before_call(vslam_ctx* c, unsigned long long key, bool eligible);
int sched_tuner_after_call(vslam_ctx* c);
std::pair<hipEvent_t, hipEvent_t>* timing_slot(vslam_ctx* c);
}  // namespace vslam

// Brackets the launches made inside its scope with HIP events on the context stream when the
// bench hook (vslam_kernel_timing_enable) names this kernel.
struct TimedScope {
    vslam_ctx* c;
    std::pair<hipEvent_t, hipEvent_t>* ev;
    hipStream_t st;
    // `tag`: the octave of the launch (-1: none) - "name@2" times only the launches tagged 2; `stream`: where the launch
    // goes when that is not the context's current stream (the scans go straight to the side stream)
    TimedScope(vslam_ctx* ctx, const char* name, int tag = -1, hipStream_t stream = nullptr)
        : c(ctx), ev(!ctx->timing_name.empty() && ctx->timing_name == name && (ctx->timing_tag < 0 || ctx->timing_tag == tag) ? vslam::timing_slot(ctx) : nullptr),
          st(stream ? stream : ctx->stream) {
        if (ev) (void)hipEventRecord(ev->first, st);
    }
    ~TimedScope() {
        if (ev) (void)hipEventRecord(ev->second, st);
    }
};
