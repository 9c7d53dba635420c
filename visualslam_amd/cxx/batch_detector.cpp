#include "batch_detector.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstring>
#include <string>

#include "vslam_cxx.hpp"

namespace vslam {

namespace {
void hip_check(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw Error(e == hipErrorOutOfMemory ? VSLAM_ERR_NOMEM : VSLAM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPX(expr) hip_check((expr), #expr)
}  // namespace

FrameKeypoints BatchResult::frame(int f) const {
    if (f < 0 || f >= n_frames) throw Error(VSLAM_ERR_RANGE, "BatchResult::frame: index out of range");
    FrameKeypoints k;
    // records beyond the host budget are missing from the END of the packed list: clip every frame to it
    const uint64_t h0 = std::min(harris_offsets[f], harris_records), h1 = std::min(harris_offsets[f + 1], harris_records);
    const uint64_t d0 = std::min(dog_offsets[f], dog_records), d1 = std::min(dog_offsets[f + 1], dog_records);
    k.harris = harris + h0;
    k.n_harris = (size_t)(h1 - h0);
    k.n_dog = (size_t)(d1 - d0);
    if (dog16) {  // the list came down as 16-byte records: this frame's SLAM::points are rebuilt here
        k.dog_expanded.resize(k.n_dog);
        vslam_points16_expand(dog16 + d0, k.n_dog, k.dog_expanded.data());
        k.dog = k.dog_expanded.data();
    } else
        k.dog = dog + d0;
    k.harris_total = harris_counts[f];
    k.dog_total = dog_counts[f];
    if (oriented_offsets) {
        const uint64_t o0 = std::min(oriented_offsets[f], oriented_records), o1 = std::min(oriented_offsets[f + 1], oriented_records);
        k.n_oriented = (size_t)(o1 - o0);
        if (oriented16) {
            k.oriented_expanded.resize(k.n_oriented);
            vslam_points16_expand(oriented16 + o0, k.n_oriented, k.oriented_expanded.data());
            k.oriented = k.oriented_expanded.data();
        } else
            k.oriented = oriented + o0;
        k.oriented_total = oriented_counts[f];
        k.oriented_survivors = oriented_survivors[f];
        if (descriptors && o1 <= descriptor_records) k.descriptors = descriptors + o0 * 128;
        if (descriptor_defined) k.descriptor_defined = descriptor_defined + (size_t)f * oriented_cap;
    }
    return k;
}

void* BatchDetector::alloc_pinned(size_t bytes) {
    void* p = nullptr;
    HIPX(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
    return p;
}
void BatchDetector::free_pinned(void* p) {
    if (p) (void)hipHostFree(p);
}

BatchDetector::BatchDetector(const Options& opt) : opt_(opt) {
    try {
        init(opt);
    } catch (...) {  // a constructor that throws runs no destructor: give back what was allocated so far
        release();
        throw;
    }
}

void BatchDetector::init(const Options& opt) {
    if (opt.batch <= 0 || opt.rows <= 0 || opt.cols <= 0 || opt.slots < 1 || opt.pipelines < 1 || opt.pipelines > 4)
        throw Error(VSLAM_ERR_INVALID, "BatchDetector: bad options");
    if (opt.custom_params)
        p_ = opt.params;
    else
        vslam_params_default(&p_, opt.rows, opt.cols);
    p_.rows = opt.rows;
    p_.cols = opt.cols;
    if (opt.describe) opt_.orient = true;
    if (opt_.orient) opt_.localize = true;
    if (opt_.localize) p_.localize = 1;
    if (opt_.orient) p_.orient = 1;
    int rc = vslam_batch_layout_query(&p_, &L_);
    if (rc != VSLAM_OK) throw Error(rc, "BatchDetector: vslam_batch_layout_query rejected the parameters");
    HIPX(hipSetDevice(opt.device));
    // Streams map onto a small number of hardware queues (4 by default): every extra stream can end up
    // sharing a queue - and therefore serialising - with the library's side streams, so the copy streams
    // exist only in host-fed use (device-resident Stream runs lost 4 % to two idle streams).
    hipStream_t us = nullptr, ds = nullptr, ps = nullptr;
    pipes_.resize((size_t)opt.pipelines);
    for (Pipe& pp : pipes_) {
        hipStream_t cs;
        HIPX(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        pp.stream = cs;
        rc = vslam_ctx_create(opt.device, cs, &pp.ctx);
        if (rc != VSLAM_OK) throw Error(rc, std::string("vslam_ctx_create: ") + vslam_status_string(rc) + " (no usable HIP device: there is no CPU fallback)");
        if (opt.yielding_side_streams || opt.tune_side_streams) (void)vslam_ctx_set_side_stream_priority(pp.ctx, 1);  // (the tuner compares low-priority pairs)
        (void)vslam_ctx_tune_side_streams(pp.ctx, opt.tune_side_streams ? 1 : 0);
        if (opt.f32_fused) (void)vslam_ctx_set_f32_fused(pp.ctx, 1);  // (VSLAM_F32_FUSED=1 sets it for every context of the process)
    }
    if (opt.host_fed) {
        HIPX(hipStreamCreateWithFlags(&us, hipStreamNonBlocking));
        HIPX(hipStreamCreateWithFlags(&ds, hipStreamNonBlocking));
        HIPX(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
    }
    up_ = us, down_ = ds, pack_ = ps;
    if (opt.host_fed) {
        rc = vslam_ctx_create(opt.device, pack_, &ctx_pack_);
        if (rc != VSLAM_OK) throw Error(rc, "vslam_ctx_create (pack stream)");
    }

    vslam_batch_out need{};
    check(vslam_batch_out_required(&p_, opt.batch, &need), pipes_[0].ctx, "vslam_batch_out_required");
    auto dmalloc = [&](size_t bytes) {
        void* p = nullptr;
        HIPX(hipMalloc(&p, bytes ? bytes : 256));
        dev_allocs_.push_back(p);
        return p;
    };
    auto pinned = [&](size_t bytes) {
        void* p = alloc_pinned(bytes);
        pinned_allocs_.push_back(p);
        return p;
    };
    // image outputs: one set per pipeline (the kernels of the batches of one pipeline are ordered on its stream)
    const bool dog = p_.n_octaves > 0;
    for (Pipe& pp : pipes_) {
        vslam_batch_out& img = pp.img;
        img.struct_size = sizeof(vslam_batch_out);
        if (p_.do_harris) {
            img.response = (float*)dmalloc(need.response_bytes), img.response_bytes = need.response_bytes;
            img.nms_mask = (uint8_t*)dmalloc(need.nms_mask_bytes), img.nms_mask_bytes = need.nms_mask_bytes;
        }
        if (dog) {
            img.pyramid = (uint8_t*)dmalloc(need.pyramid_bytes), img.pyramid_bytes = need.pyramid_bytes;
            img.extrema_bits = (uint64_t*)dmalloc(need.extrema_bits_bytes), img.extrema_bits_bytes = need.extrema_bits_bytes;
        }
    }
    // host-fed: `slots` batches between submit() and collect(); device-resident: one set of lists per pipeline
    const int nslots = opt.host_fed ? std::max(opt.slots, opt.pipelines) : opt.pipelines;
    d_totals_all_ = (uint64_t*)dmalloc(2 * sizeof(uint64_t) * (size_t)nslots);
    HIPX(hipMemset(d_totals_all_, 0, 2 * sizeof(uint64_t) * (size_t)nslots));
    d_totals_ = d_totals_all_;
    const size_t n = (size_t)opt.batch;
    packed_cap_h_ = std::min<size_t>(n * p_.harris_cap, n * opt.host_records_per_frame);
    packed_cap_p_ = std::min<size_t>(n * p_.dog_cap, n * opt.host_records_per_frame);
    packed_cap_o_ = std::min<size_t>(n * p_.oriented_cap, n * opt.host_oriented_per_frame);
    packed_cap_d_ = std::min<size_t>(n * p_.oriented_cap, n * opt.host_descriptors_per_frame);
    const bool orient = dog && opt_.orient, describe = orient && opt_.describe;
    slots_.resize(nslots);
    for (Slot& s : slots_) {
        s.out = pipes_[0].img;
        if (p_.do_harris) {
            s.out.harris_kps = (vslam_kp*)dmalloc(need.harris_kps_bytes), s.out.harris_kps_bytes = need.harris_kps_bytes;
            s.out.harris_counts = (uint32_t*)dmalloc(need.harris_counts_bytes), s.out.harris_counts_bytes = need.harris_counts_bytes;
            HIPX(hipMemset(s.out.harris_counts, 0, need.harris_counts_bytes));
        }
        if (dog) {
            s.out.dog_points = (vslam_point*)dmalloc(need.dog_points_bytes), s.out.dog_points_bytes = need.dog_points_bytes;
            s.out.dog_counts = (uint32_t*)dmalloc(need.dog_counts_bytes), s.out.dog_counts_bytes = need.dog_counts_bytes;
            HIPX(hipMemset(s.out.dog_counts, 0, need.dog_counts_bytes));
        }
        if (orient) {
            s.out.oriented_points = (vslam_point*)dmalloc(need.oriented_points_bytes), s.out.oriented_points_bytes = need.oriented_points_bytes;
            s.out.oriented_counts = (uint32_t*)dmalloc(need.oriented_counts_bytes), s.out.oriented_counts_bytes = need.oriented_counts_bytes;
            s.out.oriented_survivors = (uint32_t*)dmalloc(need.oriented_survivors_bytes), s.out.oriented_survivors_bytes = need.oriented_survivors_bytes;
            HIPX(hipMemset(s.out.oriented_counts, 0, need.oriented_counts_bytes));
            HIPX(hipMemset(s.out.oriented_survivors, 0, need.oriented_survivors_bytes));
        }
        if (describe) {
            s.out.descriptors = (float*)dmalloc(need.descriptors_bytes), s.out.descriptors_bytes = need.descriptors_bytes;
            s.out.descriptor_defined = (uint8_t*)dmalloc(need.descriptor_defined_bytes), s.out.descriptor_defined_bytes = need.descriptor_defined_bytes;
        }
        hipEvent_t e;
        HIPX(hipEventCreateWithFlags(&e, hipEventDisableTiming)), s.up_done = e;
        HIPX(hipEventCreateWithFlags(&e, hipEventDisableTiming)), s.comp_done = e;
        HIPX(hipEventCreateWithFlags(&e, hipEventDisableTiming)), s.down_done = e;
        HIPX(hipEventCreateWithFlags(&e, hipEventDisableTiming)), s.det_done = e;
        if (!opt.host_fed) continue;
        s.d_frames = (uint8_t*)dmalloc(n * (size_t)p_.rows * p_.cols);
        s.d_hpacked = (vslam_kp*)dmalloc(packed_cap_h_ * sizeof(vslam_kp));
        s.d_ppacked = (vslam_point*)dmalloc(packed_cap_p_ * sizeof(vslam_point));
        s.d_off = (uint64_t*)dmalloc(3 * (n + 1) * sizeof(uint64_t));
        HIPX(hipMemset(s.d_off, 0, 3 * (n + 1) * sizeof(uint64_t)));
        s.h_off = (uint64_t*)pinned(3 * (n + 1) * sizeof(uint64_t));
        s.h_cnt = (uint32_t*)pinned(4 * n * sizeof(uint32_t));
        std::memset(s.h_off, 0, 3 * (n + 1) * sizeof(uint64_t));
        std::memset(s.h_cnt, 0, 4 * n * sizeof(uint32_t));
        if (orient) {
            s.d_opacked = (vslam_point*)dmalloc(packed_cap_o_ * sizeof(vslam_point));
            s.h_opacked = (vslam_point*)pinned(packed_cap_o_ * sizeof(vslam_point));
        }
        if (describe) {
            s.d_dpacked = (float*)dmalloc(packed_cap_d_ * 128 * sizeof(float));
            s.h_dpacked = (float*)pinned(packed_cap_d_ * 128 * sizeof(float));
            s.h_defined = (uint8_t*)pinned(n * p_.oriented_cap);
        }
        s.h_hpacked = (vslam_kp*)pinned(packed_cap_h_ * sizeof(vslam_kp));
        s.h_ppacked = (vslam_point*)pinned(packed_cap_p_ * sizeof(vslam_point));
    }
    HIPX(hipDeviceSynchronize());
}

BatchDetector::~BatchDetector() { release(); }

void BatchDetector::release() {
    (void)hipSetDevice(opt_.device);
    (void)hipDeviceSynchronize();
    if (ctx_pack_) (void)vslam_ctx_destroy(ctx_pack_);
    ctx_pack_ = nullptr;
    for (Pipe& pp : pipes_)
        if (pp.ctx) (void)vslam_ctx_destroy(pp.ctx), pp.ctx = nullptr;
    for (Slot& s : slots_)
        for (void* e : {s.up_done, s.comp_done, s.down_done, s.det_done})
            if (e) (void)hipEventDestroy((hipEvent_t)e);
    slots_.clear();
    for (void* p : dev_allocs_) (void)hipFree(p);
    for (void* p : pinned_allocs_) (void)hipHostFree(p);
    dev_allocs_.clear();
    pinned_allocs_.clear();
    for (void** s : {&up_, &down_, &pack_}) {
        if (*s) (void)hipStreamDestroy((hipStream_t)*s);
        *s = nullptr;
    }
    for (Pipe& pp : pipes_)
        if (pp.stream) (void)hipStreamDestroy((hipStream_t)pp.stream), pp.stream = nullptr;
    pipes_.clear();
}

void BatchDetector::sync() {
    HIPX(hipSetDevice(opt_.device));
    for (Pipe& pp : pipes_) HIPX(hipStreamSynchronize((hipStream_t)pp.stream));
}

// The slot's lists + the image buffers of the pipeline whose turn it is.  The pair of totals of the slot may still be
// read by the caller's collective (hold_totals_until): the pipeline's stream waits for that first.
BatchDetector::Slot& BatchDetector::begin_batch(int slot, int n) {
    Slot& s = slots_[(size_t)slot];
    const int prev = last_pipe_;
    const bool first = batches_ == 0;
    last_pipe_ = next_pipe();
    last_slot_ = slot;
    ++batches_;
    const Pipe& pp = pipes_[(size_t)last_pipe_];
    s.out.response = pp.img.response, s.out.nms_mask = pp.img.nms_mask, s.out.pyramid = pp.img.pyramid, s.out.extrema_bits = pp.img.extrema_bits;
    // two batches in flight run staggered: this one starts when the previous one (other pipeline) is past its octave 0
    if (!first && prev != last_pipe_) check(vslam_ctx_follow(pp.ctx, pipes_[(size_t)prev].ctx), pp.ctx, "vslam_ctx_follow");
    if (s.totals_read) HIPX(hipStreamWaitEvent((hipStream_t)pp.stream, (hipEvent_t)s.totals_read, 0));
    s.totals_read = nullptr;
    s.n = n;
    d_totals_ = d_totals_all_ + 2 * (size_t)slot;
    return s;
}

void BatchDetector::detect_device(const uint8_t* d_frames, size_t frame_stride, int n) {
    if (n <= 0 || n > opt_.batch) throw Error(VSLAM_ERR_INVALID, "BatchDetector::detect_device: n outside 1..batch");
    HIPX(hipSetDevice(opt_.device));
    // device-resident batches use the lists of slot = pipeline: a batch's outputs stay valid until `pipelines` calls later
    Slot& s = begin_batch(next_pipe(), n);
    vslam_ctx* ctx = pipes_[(size_t)last_pipe_].ctx;
    if (opt_.host_fed) {  // a host-fed batch may still be packing or downloading this slot's lists
        const hipStream_t cs = (hipStream_t)pipes_[(size_t)last_pipe_].stream;
        if (s.comp_done) HIPX(hipStreamWaitEvent(cs, (hipEvent_t)s.comp_done, 0));
        if (s.down_done) HIPX(hipStreamWaitEvent(cs, (hipEvent_t)s.down_done, 0));
    }
    check(vslam_detect_batch_dev(ctx, &p_, d_frames, frame_stride, n, &s.out), ctx, "vslam_detect_batch_dev");
    check(vslam_count_totals_dev(ctx, s.out.harris_counts, s.out.dog_counts, n, d_totals_), ctx, "vslam_count_totals_dev");
}

void BatchDetector::submit(const uint8_t* host_frames, int n) {
    if (!opt_.host_fed) throw Error(VSLAM_ERR_INVALID, "BatchDetector::submit: constructed with host_fed = false");
    if (!host_frames || n <= 0 || n > opt_.batch) throw Error(VSLAM_ERR_INVALID, "BatchDetector::submit: bad arguments");
    if (in_flight() >= (int)slots_.size()) throw Error(VSLAM_ERR_INVALID, "BatchDetector::submit: every slot is in flight - collect() first");
    HIPX(hipSetDevice(opt_.device));
    Slot& s = begin_batch((int)(submitted_ % slots_.size()), n);
    vslam_ctx* ctx = pipes_[(size_t)last_pipe_].ctx;
    const hipStream_t up = (hipStream_t)up_, cs = (hipStream_t)pipes_[(size_t)last_pipe_].stream;
    const size_t N = (size_t)p_.rows * p_.cols;
    // the slot's frame buffer is free once the kernels of its previous batch have run (comp_done), its list
    // buffers once that batch's lists have been downloaded (down_done; collect() has waited for it already)
    HIPX(hipStreamWaitEvent(up, (hipEvent_t)s.comp_done, 0));
    HIPX(hipMemcpyAsync(s.d_frames, host_frames, (size_t)n * N, hipMemcpyHostToDevice, up));
    HIPX(hipEventRecord((hipEvent_t)s.up_done, up));
    HIPX(hipStreamWaitEvent(cs, (hipEvent_t)s.up_done, 0));
    HIPX(hipStreamWaitEvent(cs, (hipEvent_t)s.down_done, 0));
    check(vslam_detect_batch_dev(ctx, &p_, s.d_frames, N, n, &s.out), ctx, "vslam_detect_batch_dev");
    // this batch's {harris, dog} totals stay on the device (one pair per slot): the count all-gather can be
    // enqueued right behind submit() without the host seeing them
    check(vslam_count_totals_dev(ctx, s.out.harris_counts, s.out.dog_counts, n, d_totals_), ctx, "vslam_count_totals_dev");
    const size_t nb = (size_t)opt_.batch;
    // The packing (bandwidth-bound, 0.45 ms for a 256-frame 1080p batch) and the small downloads go to the pack stream:
    // they read only this slot's lists, so they run beside the first kernels of the NEXT batch instead of in front of them.
    const hipStream_t ps = (hipStream_t)pack_;
    HIPX(hipEventRecord((hipEvent_t)s.det_done, cs));
    HIPX(hipStreamWaitEvent(ps, (hipEvent_t)s.det_done, 0));
    if (s.out.harris_kps)
        check(vslam_pack_lists_dev(ctx_pack_, s.out.harris_kps, sizeof(vslam_kp), p_.harris_cap, s.out.harris_counts, n, s.d_hpacked,
                                   packed_cap_h_ * sizeof(vslam_kp), s.d_off),
              ctx_pack_, "vslam_pack_lists_dev (harris)");
    if (s.out.dog_points && opt_.compact_points)
        check(vslam_pack_points16_dev(ctx_pack_, s.out.dog_points, p_.dog_cap, s.out.dog_counts, n, reinterpret_cast<vslam_point16*>(s.d_ppacked),
                                      packed_cap_p_ * sizeof(vslam_point16), s.d_off + (nb + 1)),
              ctx_pack_, "vslam_pack_points16_dev (dog)");
    else if (s.out.dog_points)
        check(vslam_pack_lists_dev(ctx_pack_, s.out.dog_points, sizeof(vslam_point), p_.dog_cap, s.out.dog_counts, n, s.d_ppacked,
                                   packed_cap_p_ * sizeof(vslam_point), s.d_off + (nb + 1)),
              ctx_pack_, "vslam_pack_lists_dev (dog)");
    if (s.out.oriented_points && opt_.compact_points)
        check(vslam_pack_points16_dev(ctx_pack_, s.out.oriented_points, p_.oriented_cap, s.out.oriented_counts, n, reinterpret_cast<vslam_point16*>(s.d_opacked),
                                      packed_cap_o_ * sizeof(vslam_point16), s.d_off + 2 * (nb + 1)),
              ctx_pack_, "vslam_pack_points16_dev (oriented)");
    else if (s.out.oriented_points)
        check(vslam_pack_lists_dev(ctx_pack_, s.out.oriented_points, sizeof(vslam_point), p_.oriented_cap, s.out.oriented_counts, n, s.d_opacked,
                                   packed_cap_o_ * sizeof(vslam_point), s.d_off + 2 * (nb + 1)),
              ctx_pack_, "vslam_pack_lists_dev (oriented)");
    if (s.out.descriptors)  // same counts, same capacity: the same offsets, 512-byte records
        check(vslam_pack_lists_dev(ctx_pack_, s.out.descriptors, 128 * sizeof(float), p_.oriented_cap, s.out.oriented_counts, n, s.d_dpacked,
                                   packed_cap_d_ * 128 * sizeof(float), s.d_off + 2 * (nb + 1)),
              ctx_pack_, "vslam_pack_lists_dev (descriptors)");
    // the small things travel right behind the packing: offsets and true counts
    HIPX(hipMemcpyAsync(s.h_off, s.d_off, 3 * (nb + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, ps));
    if (s.out.oriented_counts) HIPX(hipMemcpyAsync(s.h_cnt + 2 * nb, s.out.oriented_counts, (size_t)n * 4, hipMemcpyDeviceToHost, ps));
    if (s.out.oriented_survivors) HIPX(hipMemcpyAsync(s.h_cnt + 3 * nb, s.out.oriented_survivors, (size_t)n * 4, hipMemcpyDeviceToHost, ps));
    if (s.out.harris_counts) HIPX(hipMemcpyAsync(s.h_cnt, s.out.harris_counts, (size_t)n * 4, hipMemcpyDeviceToHost, ps));
    if (s.out.dog_counts) HIPX(hipMemcpyAsync(s.h_cnt + nb, s.out.dog_counts, (size_t)n * 4, hipMemcpyDeviceToHost, ps));
    HIPX(hipEventRecord((hipEvent_t)s.comp_done, ps));
    ++submitted_;
}

const BatchResult& BatchDetector::collect() {
    if (in_flight() <= 0) throw Error(VSLAM_ERR_INVALID, "BatchDetector::collect: nothing in flight");
    HIPX(hipSetDevice(opt_.device));
    Slot& s = slots_[collected_ % slots_.size()];
    const hipStream_t ds = (hipStream_t)down_;
    const size_t nb = (size_t)opt_.batch;
    HIPX(hipEventSynchronize((hipEvent_t)s.comp_done));  // offsets and counts are on the host now
    BatchResult& r = s.res;
    r = BatchResult{};
    r.n_frames = s.n;
    // out of the pinned staging buffers: this slot's next submit() - possible right after this call, with `slots` - 1
    // batches still in flight - overwrites them from the pack stream while the caller reads the result
    s.res_off.assign(s.h_off, s.h_off + 3 * (nb + 1));
    s.res_cnt.assign(s.h_cnt, s.h_cnt + 4 * nb);
    const uint64_t* h_off = s.res_off.data();
    const uint32_t* h_cnt = s.res_cnt.data();
    r.harris_offsets = h_off;
    r.dog_offsets = h_off + (nb + 1);
    r.harris_counts = h_cnt;
    r.dog_counts = h_cnt + nb;
    r.harris = s.h_hpacked;
    const size_t prec = opt_.compact_points ? sizeof(vslam_point16) : sizeof(vslam_point);  // bytes per SLAM::point on the wire
    if (opt_.compact_points)
        r.dog16 = reinterpret_cast<const vslam_point16*>(s.h_ppacked);
    else
        r.dog = s.h_ppacked;
    const uint64_t th = s.out.harris_kps ? r.harris_offsets[s.n] : 0, tp = s.out.dog_points ? r.dog_offsets[s.n] : 0;
    r.harris_records = std::min<uint64_t>(th, packed_cap_h_);
    r.dog_records = std::min<uint64_t>(tp, packed_cap_p_);
    r.truncated = th > packed_cap_h_ || tp > packed_cap_p_;
    for (int f = 0; f < s.n && !r.truncated; ++f)
        r.truncated = (s.out.harris_kps && r.harris_counts[f] > p_.harris_cap) || (s.out.dog_points && r.dog_counts[f] > p_.dog_cap);
    // only the records that exist
    if (r.harris_records)
        HIPX(hipMemcpyAsync(s.h_hpacked, s.d_hpacked, r.harris_records * sizeof(vslam_kp), hipMemcpyDeviceToHost, ds));
    if (r.dog_records)
        HIPX(hipMemcpyAsync(s.h_ppacked, s.d_ppacked, r.dog_records * prec, hipMemcpyDeviceToHost, ds));
    if (s.out.oriented_points) {
        r.oriented_offsets = h_off + 2 * (nb + 1);
        if (opt_.compact_points)
            r.oriented16 = reinterpret_cast<const vslam_point16*>(s.h_opacked);
        else
            r.oriented = s.h_opacked;
        r.oriented_counts = h_cnt + 2 * nb;
        r.oriented_survivors = h_cnt + 3 * nb;
        r.oriented_cap = p_.oriented_cap;
        const uint64_t to = r.oriented_offsets[s.n];
        r.oriented_records = std::min<uint64_t>(to, packed_cap_o_);
        r.truncated = r.truncated || to > packed_cap_o_;
        for (int f = 0; f < s.n && !r.truncated; ++f) r.truncated = r.oriented_survivors[f] > p_.oriented_cap || r.oriented_counts[f] > p_.oriented_cap;
        if (r.oriented_records)
            HIPX(hipMemcpyAsync(s.h_opacked, s.d_opacked, r.oriented_records * prec, hipMemcpyDeviceToHost, ds));
        if (s.out.descriptors) {
            r.descriptors = s.h_dpacked;
            r.descriptor_defined = s.h_defined;
            r.descriptor_records = std::min<uint64_t>(to, packed_cap_d_);  // descriptors beyond the host budget stay on the device
            if (r.descriptor_records)
                HIPX(hipMemcpyAsync(s.h_dpacked, s.d_dpacked, r.descriptor_records * 128 * sizeof(float), hipMemcpyDeviceToHost, ds));
            HIPX(hipMemcpyAsync(s.h_defined, s.out.descriptor_defined, (size_t)s.n * p_.oriented_cap, hipMemcpyDeviceToHost, ds));
        }
    }
    HIPX(hipEventRecord((hipEvent_t)s.down_done, ds));
    HIPX(hipEventSynchronize((hipEvent_t)s.down_done));
    ++collected_;
    return r;
}

}  // namespace vslam
