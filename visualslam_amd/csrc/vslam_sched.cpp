// Side-stream placement (the opt-in stream tuner), the join watchdog and the bench timing hook of the batched path:
// host-side scheduling code only, no kernels.  Kept apart from the C ABI / launch plan (vslam_hip.hip).
#include "vslam_ctx.h"

#include <algorithm>

namespace vslam {

std::pair<hipEvent_t, hipEvent_t>* timing_slot(vslam_ctx* c) {
    if (c->timing_used == c->timing_ev.size()) {
        if (c->timing_ev.size() >= 65536) return nullptr;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess) return nullptr;
        if (hipEventCreate(&b) != hipSuccess) {
            (void)hipEventDestroy(a);
            return nullptr;
        }
        c->timing_ev.emplace_back(a, b);
    }
    return &c->timing_ev[c->timing_used++];
}

// ---- side-stream placement ---------------------------------------------------------------------------------------
// HIP binds every stream to one of GPU_MAX_HW_QUEUES hardware queues per priority level, and the placement is not ours to
// choose.  Measured: depending on the queue a LOW-priority side stream lands on, the batch runs up
// to 20 % slower (the same binary: 11.4 k frames/s with 3 queues per level, 14.2 k with 12) - on one bad queue the side
// kernels crawl while the main stream's queue sits on the barrier that waits for them.  A host that wants the library to
// look for a better pair OPTS IN (vslam_ctx_tune_side_streams; `Stream --tuner`): the 2nd to 5th full-size batch
// call of the context then run on three candidate pairs of side streams (the first pair twice), each call bracketed by two
// events on the main stream, and the first later call that finds all of them complete (hipEventQuery: the entry point stays
// asynchronous, nothing waits on the host) adopts the fastest pair - the first one unless another is at least 3 % faster.
// Only calls of one shape are compared (calls of another shape, small or odd calls run on the pair in use and do not
// disturb the comparison; a caller whose full-size shape keeps changing ends it on the first pair after three restarts);
// nothing is timed while the stream is being captured.  Results never depend on the streams a call runs on.
static int tuner_pair_of(int slot) { return slot == StreamTuner::K ? 0 : slot; }

static int create_side_stream(vslam_ctx* c, int prio_lo, hipStream_t* out) {
    if (prio_lo == 0 || hipStreamCreateWithPriority(out, hipStreamNonBlocking, prio_lo) != hipSuccess) {
        (void)hipGetLastError();  // priorities are a speed matter only
        HIPCHK(c, hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    }
    return VSLAM_OK;
}

static void tuner_finish(vslam_ctx* c, int chosen) {
    StreamTuner& t = c->tuner;
    t.chosen = chosen;
    t.done = true;
    t.measuring = -1;
    c->aux[0] = t.cand[chosen][0], c->aux[1] = t.cand[chosen][1];
    // the watchdog's cached pair of this level must be the pair that survives (the other candidates are destroyed below)
    if (c->watch.level <= 1 && t.cand[chosen][0]) c->watch.pair[c->watch.level][0] = t.cand[chosen][0], c->watch.pair[c->watch.level][1] = t.cand[chosen][1];
    for (int k = 0; k < StreamTuner::K; ++k) {
        if (k == chosen) continue;
        for (hipStream_t& st : t.cand[k])
            if (st) (void)hipStreamSynchronize(st), (void)hipStreamDestroy(st), st = nullptr;
    }
    for (int m = 0; m < StreamTuner::M; ++m) {
        if (t.t0[m]) (void)hipEventDestroy(t.t0[m]), t.t0[m] = nullptr;
        if (t.t1[m]) (void)hipEventDestroy(t.t1[m]), t.t1[m] = nullptr;
    }
}

// Before the fork of a batch call (ensure_aux has run): picks the pair of side streams this call uses.  Never blocks.
int sched_tuner_before_call(vslam_ctx* c, unsigned long long key, bool eligible) {
    StreamTuner& t = c->tuner;
    if (t.done || !t.enabled) return VSLAM_OK;
    if (c->watch.level > 0) {  // the pairs it compares are yielding ones: at the main stream's priority (or without side
        t.done = true;         // streams) there is nothing to compare, and the join watchdog must not wait for a verdict
        return VSLAM_OK;
    }
    if (c->prio_lo == 0) {  // no priority levels: one pair is as good as another
        t.done = true;
        return VSLAM_OK;
    }
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(c->stream, &cap) != hipSuccess) (void)hipGetLastError();
    if (cap != hipStreamCaptureStatusNone) return VSLAM_OK;  // a captured call records no timing events and runs on the pair in use
    ++t.calls;
    if (t.calls == 1 || !eligible) return VSLAM_OK;  // the first call pays one-time costs; small / odd calls are not what is being tuned
    if (t.key == 0) t.key = key;
    if (key != t.key) {  // another full-size shape: start over with it (the pairs created so far stay), but not for ever
        if (++t.resets > 3) {
            tuner_finish(c, 0);
            return VSLAM_OK;
        }
        t.key = key;
        t.measured = 0;
        c->aux[0] = t.cand[0][0], c->aux[1] = t.cand[0][1];
        return VSLAM_OK;
    }
    if (t.measured < StreamTuner::M) {
        const int m = t.measured, k = tuner_pair_of(m);
        for (hipStream_t& st : t.cand[k])
            if (!st) TRY(create_side_stream(c, c->prio_lo, &st));  // created while the other pairs exist: binds to another queue
        if (!t.t0[m]) HIPCHK(c, hipEventCreate(&t.t0[m]));
        if (!t.t1[m]) HIPCHK(c, hipEventCreate(&t.t1[m]));
        c->aux[0] = t.cand[k][0], c->aux[1] = t.cand[k][1];
        HIPCHK(c, hipEventRecord(t.t0[m], c->stream));
        t.measuring = m;
        return VSLAM_OK;
    }
    // every candidate has been timed: decide once the last measured call has finished - until then on the first pair
    c->aux[0] = t.cand[0][0], c->aux[1] = t.cand[0][1];
    const hipError_t q = hipEventQuery(t.t1[StreamTuner::M - 1]);
    if (q == hipErrorNotReady) {
        (void)hipGetLastError();
        return VSLAM_OK;
    }
    float ms[StreamTuner::M] = {};
    bool ok = q == hipSuccess;
    for (int m = 0; m < StreamTuner::M && ok; ++m) ok = hipEventElapsedTime(&ms[m], t.t0[m], t.t1[m]) == hipSuccess;
    int best = 0;
    if (ok) {
        const float first = std::min(ms[0], ms[StreamTuner::K]);  // pair 0 was timed twice (the early calls run on cold clocks)
        float best_ms = first;
        for (int k = 1; k < StreamTuner::K; ++k)
            if (ms[k] < 0.97f * first && ms[k] < best_ms) best = k, best_ms = ms[k];
    } else {
        (void)hipGetLastError();
    }
    tuner_finish(c, best);  // the discarded streams are idle (every measured call has joined them back): nothing to wait for
    return VSLAM_OK;
}

int sched_tuner_after_call(vslam_ctx* c) {
    StreamTuner& t = c->tuner;
    if (t.measuring < 0) return VSLAM_OK;
    HIPCHK(c, hipEventRecord(t.t1[t.measuring], c->stream));
    t.measuring = -1;
    ++t.measured;
    return VSLAM_OK;
}

// Before the fork of a batch call: reads finished measurements, moves between the levels, starts this call's measurement.
static int watch_set_level(vslam_ctx* c, int level) {
    JoinWatch& w = c->watch;
    for (bool& l : w.live) l = false;  // measurements in flight belong to the form being left
    w.calls = w.n_meas = 0;
    w.best_total = 0.0f;
    w.level = level;
    if (level <= 1) {
        for (int i = 0; i < 2; ++i) {
            if (!w.pair[level][i]) TRY(create_side_stream(c, level == 0 ? c->prio_dev_lo : 0, &w.pair[level][i]));
            c->aux[i] = w.pair[level][i];  // the pair being left is idle: every call joins its side streams back
        }
        c->prio_lo = level == 0 ? c->prio_dev_lo : 0;
    }
    return VSLAM_OK;
}

int sched_watch_before_call(vslam_ctx* c, unsigned long long key, bool eligible, bool capturing) {
    JoinWatch& w = c->watch;
    w.recording = -1;
    if (w.done || w.disabled || capturing || (c->tuner.enabled && !c->tuner.done)) return VSLAM_OK;
    if (eligible && key != w.key) {  // calls of another shape: their times say nothing about the ones measured so far
        for (bool& l : w.live) l = false;
        w.calls = w.n_meas = 0;
        w.best_total = 0.0f;
        if (w.key != 0 && ++w.restarts > 3) {  // a caller whose shape keeps changing: stop watching (a running trial ends where it started)
            if (w.trial_from >= 0) TRY(watch_set_level(c, w.trial_from));
            w.trial_from = -1;
            w.done = true;
            return VSLAM_OK;
        }
        w.key = key;
    }
    if (w.level <= 1 && !w.pair[w.level][0]) w.pair[w.level][0] = c->aux[0], w.pair[w.level][1] = c->aux[1];
    for (int i = 0; i < JoinWatch::RING; ++i) {
        if (!w.live[i]) continue;
        const hipError_t q = hipEventQuery(w.t1[i]);
        if (q == hipErrorNotReady) {
            (void)hipGetLastError();
            continue;
        }
        w.live[i] = false;
        float total = 0.0f, lag = 0.0f;
        if (q != hipSuccess || hipEventElapsedTime(&total, w.t0[i], w.t1[i]) != hipSuccess || hipEventElapsedTime(&lag, w.tm[i], w.t1[i]) != hipSuccess || !(total > 0.0f)) {
            (void)hipGetLastError();
            continue;
        }
        w.last_lag_frac = lag / total;
        if (w.n_meas < JoinWatch::NEED) {
            w.lag[w.n_meas++] = lag / total;
            w.best_total = (w.best_total == 0.0f || total < w.best_total) ? total : w.best_total;
        }
    }
    if (w.n_meas >= JoinWatch::NEED) {
        const float a = w.lag[0], b = w.lag[1], m = w.lag[2];
        const float med = std::max(std::min(a, b), std::min(std::max(a, b), m));
        w.level_best[w.level] = w.best_total;
        bool keep = true;
        if (w.trial_from >= 0 && !(w.best_total < 0.99f * w.level_best[w.trial_from])) {  // the trial did not pay: go back, stop
            TRY(watch_set_level(c, w.trial_from));
            w.trial_from = -1;
            w.done = true;
            keep = false;
        }
        if (keep) {
            w.trial_from = -1;
            const float limit = w.level == 0 ? 0.03f : 0.10f;
            if (w.level < 2 && med > limit && (w.level == 1 || c->prio_dev_lo != 0)) {  // (level 0 without priority levels IS level 1)
                const int from = w.level;
                TRY(watch_set_level(c, from + 1));
                w.trial_from = from;
            } else
                w.done = true;
        }
        if (w.done) return VSLAM_OK;
    }
    if (!eligible) return VSLAM_OK;
    if (++w.calls == 1) return VSLAM_OK;  // the first call of a form pays one-time costs
    const int slot = w.head;
    if (w.live[slot]) return VSLAM_OK;  // the host is more than RING calls ahead: skip this one
    if (!w.t0[slot]) {
        HIPCHK(c, hipEventCreate(&w.t0[slot]));
        HIPCHK(c, hipEventCreate(&w.tm[slot]));
        HIPCHK(c, hipEventCreate(&w.t1[slot]));
    }
    HIPCHK(c, hipEventRecord(w.t0[slot], c->stream));
    w.recording = slot;
    w.head = (slot + 1) % JoinWatch::RING;
    return VSLAM_OK;
}

int sched_ensure_aux(vslam_ctx* c) {
    if (c->ev_fork) return VSLAM_OK;
    int prio_lo = 0, prio_hi = 0;
    if (hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess) {  // numerically: lowest priority, highest priority
        (void)hipGetLastError();  // priorities are a speed matter only: do not leave the error for the next launch check
        prio_lo = 0;
    }
    c->prio_dev_lo = prio_lo;
    if (prio_lo == 0 && c->watch.level == 0) c->watch.level = 1;  // no priority levels on this device
    c->prio_lo = c->watch.level >= 1 ? 0 : prio_lo;
    for (int i = 0; i < vslam_ctx::kAux; ++i) {
        // aux[0], aux[1] (Harris chain, scans and lists) yield to the octave kernels; aux[2] carries only the
        // second-half upsample, which the main stream WAITS for: at low priority it was starved for the whole
        // first-half octave kernel whenever its start slipped behind that kernel's (C++ host, 0.35 ms per step)
        if (i == 2)
            HIPCHK(c, hipStreamCreateWithFlags(&c->aux[i], hipStreamNonBlocking));
        else
            TRY(create_side_stream(c, c->prio_lo, &c->aux[i]));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    c->tuner.cand[0][0] = c->aux[0], c->tuner.cand[0][1] = c->aux[1];
    for (auto& e : c->ev_oct) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_up2, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_chunk, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_pack, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_list0, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_edge, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_or_fork, hipEventDisableTiming));
    for (auto& e : c->ev_or_join) HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    return VSLAM_OK;
}

void sched_init_from_env(vslam_ctx* c) {
    const char* jw = std::getenv("VSLAM_JOIN_WATCH");
    c->watch.disabled = jw && jw[0] == '0';
    if (const char* sp = std::getenv("VSLAM_SIDE_PRIORITY")) c->watch.level = (sp[0] == 'l' || sp[0] == 'L') ? 0 : 1;  // low | main
    if (const char* lv = VSLAM_DIAG_ENV("VSLAM_JOIN_WATCH_LEVEL")) {  // A/B runs: start (and stay) at a level; the API is vslam_ctx_pin_side_streams
        c->watch.level = std::min(2, std::max(0, std::atoi(lv)));
        c->watch.done = c->watch.pinned = true;
    }
    const char* t = std::getenv("VSLAM_STREAM_TUNER");
    if (t && t[0] == '1') (void)vslam_ctx_tune_side_streams(c, 1);
}

void sched_destroy(vslam_ctx* c) {
    for (auto& ev : c->timing_ev) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    if (c->ev_fork && !c->tuner.done) tuner_finish(c, 0);  // candidate pairs of an unfinished comparison go first (aux = pair 0 again)
    for (int i = 0; i < vslam_ctx::kAux; ++i) {
        if (c->aux[i]) (void)hipStreamDestroy(c->aux[i]);
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
    }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int i = 0; i < JoinWatch::RING; ++i)
        for (hipEvent_t e : {c->watch.t0[i], c->watch.tm[i], c->watch.t1[i]})
            if (e) (void)hipEventDestroy(e);
    for (auto& pr : c->watch.pair)
        for (hipStream_t st : pr)
            if (st && st != c->aux[0] && st != c->aux[1]) (void)hipStreamSynchronize(st), (void)hipStreamDestroy(st);
}

}  // namespace vslam

using namespace vslam;

extern "C" {

int vslam_kernel_timing_enable(vslam_ctx* c, const char* name) {
    if (!c) return VSLAM_ERR_INVALID;
    c->timing_name = name ? name : "";
    c->timing_tag = -1;
    const size_t at = c->timing_name.find('@');  // "k_pyr_octave@1": the launches of octave 1 only
    if (at != std::string::npos) {
        c->timing_tag = std::atoi(c->timing_name.c_str() + at + 1);
        c->timing_name.resize(at);
    }
    c->timing_used = 0;
    return VSLAM_OK;
}
int vslam_kernel_timing_read(vslam_ctx* c, int* launches, double* total_ms) {
    TRY(bind_device(c));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // every batch call joins its side streams back: their events are complete too
    double tot = 0;
    for (size_t i = 0; i < c->timing_used; ++i) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->timing_ev[i].first, c->timing_ev[i].second));
        tot += ms;
    }
    if (launches) *launches = (int)c->timing_used;
    if (total_ms) *total_ms = tot;
    c->timing_used = 0;
    return VSLAM_OK;
}

int vslam_ctx_side_stream_report(const vslam_ctx* c, int* pair, int* state) {
    if (!c) return VSLAM_ERR_INVALID;
    if (pair) *pair = c->tuner.chosen;
    if (state) *state = c->tuner.done ? 2 : ((c->tuner.enabled && c->tuner.calls > 1) ? 1 : 0);
    return VSLAM_OK;
}

int vslam_ctx_set_side_stream_priority(vslam_ctx* c, int low) {
    if (!c) return VSLAM_ERR_INVALID;
    if (c->ev_fork) return fail(c, VSLAM_ERR_UNSUPPORTED, "set_side_stream_priority: the side streams exist already - call it before the context's first batch call");
    if (!c->watch.pinned) c->watch.level = low ? 0 : 1;
    return VSLAM_OK;
}

int vslam_ctx_join_watch_report(const vslam_ctx* c, int* level, int* done, float* last_lag_fraction) {
    if (!c) return VSLAM_ERR_INVALID;
    if (level) *level = c->watch.level;
    if (done) *done = c->watch.done ? 1 : 0;
    if (last_lag_fraction) *last_lag_fraction = c->watch.last_lag_frac;
    return VSLAM_OK;
}

int vslam_ctx_tune_side_streams(vslam_ctx* c, int on) {
    if (!c) return VSLAM_ERR_INVALID;
    if (c->tuner.done) return VSLAM_OK;  // a finished comparison stays finished
    if (on && c->ev_fork)  // (the watchdog has cached the pair in use by then, and the pairs compared are created beside it)
        return fail(c, VSLAM_ERR_UNSUPPORTED, "tune_side_streams: the side streams exist already - call it before the context's first batch call");
    c->tuner.enabled = on != 0;
    // the comparison is between pairs of YIELDING streams: asking for it asks for those (a pinned level stays; the tuner then ends at
    // its first call and the join watchdog runs as if it had never been asked)
    if (on && !c->watch.pinned) c->watch.level = 0;
    return VSLAM_OK;
}

int vslam_ctx_set_join_watch(vslam_ctx* c, int on) {
    if (!c) return VSLAM_ERR_INVALID;
    c->watch.disabled = on == 0;
    return VSLAM_OK;
}

int vslam_ctx_pin_side_streams(vslam_ctx* c, int level) {
    if (!c || level < 0 || level > 2) return VSLAM_ERR_INVALID;
    if (c->ev_fork) return fail(c, VSLAM_ERR_UNSUPPORTED, "pin_side_streams: the side streams exist already - call it before the context's first batch call");
    c->watch.level = level;
    c->watch.done = c->watch.pinned = true;
    return VSLAM_OK;
}

}  // extern "C"
