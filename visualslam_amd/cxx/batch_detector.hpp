// C++17 host of the THROUGHPUT path: vslam_detect_batch_dev (BASELINE config 4 / 5) behind a class, so
// that the batched detection the benchmark measures is driven from C++ through the C ABI like the
// per-image API of vslam_cxx.hpp (north star: "host code stays C++ calling HIP through a thin C-ABI
// layer").  The reference's callers are C++ main()s that feed one image at a time
// (Harris_corners.cpp:146-193, Diff_of_Gauss.cpp:727-877); a camera stream is the same loop over frames,
// and this class is that loop for batches of frames on one GPU:
//
//   device-resident   detect_device(d_frames, stride, n): frames already in HBM, outputs stay in HBM
//                     (device_outputs()), asynchronous on stream() - what bench.py times.  Consecutive
//                     batches alternate between `pipelines` streams, so two batches' kernels overlap;
//   host-fed          submit(host_frames, n) / collect(): frames start in (pinned) host memory and the two
//                     keypoint lists end there.  Up to `slots` batches are in flight: the upload of batch
//                     k+1 / k+2 (own stream) and the download of batch k-1's lists (own stream) overlap the
//                     kernels of batch k.  Only the records that exist travel back: the per-frame lists
//                     are packed back to back on the device (vslam_pack_lists_dev) and offsets[n] records
//                     are copied, not n x cap.
//
// All device memory is allocated once in the constructor (hipMalloc / hipHostMalloc: nothing is
// allocated per batch).  Errors surface as vslam::Error.  One object per GPU; not thread-safe.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/vslam.h"

namespace vslam {

// One frame's keypoints inside a collected batch (host pointers into the batch's packed lists).
struct FrameKeypoints {
    const vslam_kp* harris = nullptr;   // row-major order (Harris_corners.cpp:139 criterion)
    size_t n_harris = 0;                // records present = min(total, harris_cap)
    const vslam_point* dog = nullptr;   // (octave, level, i, j) order, SLAM::point layout
    size_t n_dog = 0;
    uint32_t harris_total = 0, dog_total = 0;  // what the frame really has (may exceed the caps)
    // Options::orient: filterKeypoints' output {row, col, angle, 0, octave, level}, (octave, keypoint, bin) order
    const vslam_point* oriented = nullptr;
    size_t n_oriented = 0;
    uint32_t oriented_total = 0, oriented_survivors = 0;  // survivors > oriented_cap: the frame's list is truncated
    // Options::describe: SIFT() descriptors of the oriented points (n_oriented rows of 128 floats) and, per point,
    // whether its rotated window was defined (include/vslam.h: vslam_sift_descriptors); may be null
    const float* descriptors = nullptr;
    const uint8_t* descriptor_defined = nullptr;
    // Options::compact_points: the batch's SLAM::point lists travelled as 16-byte records; `dog` / `oriented` then point
    // into these re-expanded copies, which the FrameKeypoints owns (a moved vector keeps its buffer: the pointers stay valid)
    std::vector<vslam_point> dog_expanded, oriented_expanded;

    FrameKeypoints() = default;
    FrameKeypoints(FrameKeypoints&&) = default;
    FrameKeypoints& operator=(FrameKeypoints&&) = default;
    // a COPY owns copies of the expanded lists: its pointers follow them
    FrameKeypoints(const FrameKeypoints& o) { copy_from(o); }
    FrameKeypoints& operator=(const FrameKeypoints& o) {
        if (this != &o) copy_from(o);
        return *this;
    }

private:
    void copy_from(const FrameKeypoints& o) {
        harris = o.harris, n_harris = o.n_harris, n_dog = o.n_dog, harris_total = o.harris_total, dog_total = o.dog_total;
        n_oriented = o.n_oriented, oriented_total = o.oriented_total, oriented_survivors = o.oriented_survivors;
        descriptors = o.descriptors, descriptor_defined = o.descriptor_defined;
        dog_expanded = o.dog_expanded, oriented_expanded = o.oriented_expanded;
        dog = o.dog == o.dog_expanded.data() ? dog_expanded.data() : o.dog;
        oriented = o.oriented == o.oriented_expanded.data() ? oriented_expanded.data() : o.oriented;
    }
};

struct BatchResult {
    int n_frames = 0;
    const uint64_t* harris_offsets = nullptr;  // [n_frames + 1] record offsets into `harris`
    const uint64_t* dog_offsets = nullptr;     // [n_frames + 1]
    const vslam_kp* harris = nullptr;          // packed, pinned host memory
    const vslam_point* dog = nullptr;          // nullptr with Options::compact_points: see dog16
    const uint32_t* harris_counts = nullptr;   // [n_frames] true totals
    const uint32_t* dog_counts = nullptr;
    uint64_t harris_records = 0, dog_records = 0;  // records present in the packed lists
    // orientation / descriptor stage (Options::orient / describe), same packing
    const uint64_t* oriented_offsets = nullptr;
    const vslam_point* oriented = nullptr;
    const float* descriptors = nullptr;          // [oriented_records][128]
    const uint8_t* descriptor_defined = nullptr; // [n_frames][oriented_cap] (not packed)
    const uint32_t* oriented_counts = nullptr;
    const uint32_t* oriented_survivors = nullptr;
    uint64_t oriented_records = 0, descriptor_records = 0;
    uint32_t oriented_cap = 0;
    bool truncated = false;  // a frame exceeded its cap, or the batch the host budget (records beyond it are missing)
    // Options::compact_points: the packed SLAM::point lists as 16-byte records (vslam_pack_points16_dev: a third fewer
    // bytes over PCIe); frame() re-expands a frame's records to SLAM::point (vslam_points16_expand), byte-identical
    const vslam_point16* dog16 = nullptr;
    const vslam_point16* oriented16 = nullptr;
    FrameKeypoints frame(int f) const;
};

class BatchDetector {
public:
    struct Options {
        int device = 0;
        int rows = 1080, cols = 1920;
        int batch = 256;                 // frames per batch (capacity of every buffer)
        bool custom_params = false;      // false: vslam_params_default(rows, cols)
        vslam_params params{};
        int slots = 3;                   // host-fed batches in flight (>= 1); device buffers of frames + lists per slot
        // vslam_ctx_set_f32_fused on every pipeline's context: the f32 stages (orient / describe) with fused multiply-adds, as an
        // OpenCV that dispatches its AVX2 + FMA3 code computes them (include/vslam.h); default: every product and sum rounded
        bool f32_fused = false;
        bool tune_side_streams = false;  // vslam_ctx_tune_side_streams on every pipeline's context (include/vslam.h; Stream --tuner)
        // vslam_ctx_set_side_stream_priority(ctx, 1): the library's side streams at the lowest priority (they yield to the octave
        // kernels).  For a host that controls its hardware-queue layout (GPU_MAX_HW_QUEUES >= the streams of the process): +2-3 %;
        // on an unlucky layout -18 %, which is why the library's default is the main stream's priority
        bool yielding_side_streams = false;
        // Batches whose KERNELS may run at the same time (1..4).  Each pipeline is a context + compute stream + set of
        // image buffers (response, mask, pyramid, bitmask: 33 GB for 256 x 1080p) of its own; consecutive batches
        // alternate between them, each starting once its predecessor is past octave 0 (vslam_ctx_follow), so that the
        // tail of batch k (coarse octaves, scans, lists: short kernels that leave issue slots idle) runs under the
        // octave-0 kernels of batch k + 1.  Worth +3..5 % frames/s when the streams land on hardware queues of their
        // own, and -20 % when they do not: the outcome follows GPU_MAX_HW_QUEUES and the order in which streams are first
        // used (DESIGN section 5.4 has the table), so the default stays at one pipeline.
        int pipelines = 1;
        // host-fed: the DoG and oriented lists cross PCIe as 16-byte records {row, col, value, level | octave << 8 | padding << 16}
        // instead of SLAM::point's 24 bytes (8 of which are constants of the list); BatchResult::frame() re-expands them
        bool compact_points = false;
        size_t host_records_per_frame = (size_t)1 << 17;  // pinned host budget per list, averaged over the batch
        bool host_fed = true;            // false: no frame / list staging buffers at all (device-resident use only)
        // The rest of the reference's DoG executable for every frame of the batch (Diff_of_Gauss.cpp:785-791):
        // localize: the DoG list = FeaturePointLocalization survivors instead of the contrast-8 candidates;
        // orient: + filterKeypoints (implies localize); describe: + SIFT descriptors (implies orient)
        bool localize = false, orient = false, describe = false;
        size_t host_oriented_per_frame = (size_t)1 << 14;    // pinned host budget of the oriented list, averaged over the batch
        size_t host_descriptors_per_frame = (size_t)1 << 11; // ... of the descriptors (512 bytes each)
    };
    explicit BatchDetector(const Options& opt);
    ~BatchDetector();
    BatchDetector(const BatchDetector&) = delete;
    BatchDetector& operator=(const BatchDetector&) = delete;

    const vslam_params& params() const { return p_; }
    const vslam_batch_layout& layout() const { return L_; }
    vslam_ctx* context() const { return pipes_[last_pipe_].ctx; }
    // hipStream_t of the kernels of the most recent batch (detect_device / submit), and of the batch that comes next
    void* stream() const { return pipes_[last_pipe_].stream; }
    void* next_stream() const { return pipes_[next_pipe()].stream; }
    int pipelines() const { return (int)pipes_.size(); }

    // ---- device-resident
    // Harris + DoG over n frames in HBM (frame f at d_frames + f * frame_stride).  Asynchronous on stream().
    void detect_device(const uint8_t* d_frames, size_t frame_stride, int n);
    // The device buffers the most recent detect_device / submit wrote (valid on stream()): the image buffers of its
    // pipeline - overwritten by the batch `pipelines` calls later - and the lists of its slot.
    const vslam_batch_out& device_outputs() const { return slots_[last_slot_].out; }
    // {harris, dog} totals of the last detect_device() / submit() as two uint64 in device memory
    // (vslam_count_totals_dev; one pair per slot): the send buffer of the count all-gather, valid on stream().
    // A reader on another stream (the collective's) names the event that marks its read with hold_totals_until():
    // the batch that overwrites the pair waits for it.
    const uint64_t* device_totals() const { return d_totals_; }
    void hold_totals_until(void* hip_event) { slots_[last_slot_].totals_read = hip_event; }
    void sync();  // every pipeline's stream

    // ---- host-fed pipeline
    // Enqueues upload + detection + list packing of one batch.  host_frames: n dense frames; pinned memory
    // (alloc_pinned) for an asynchronous upload.  Throws if `slots` batches are already in flight.
    void submit(const uint8_t* host_frames, int n);
    // Waits for the oldest batch in flight, downloads its lists and returns them.  The result (offsets, counts and the
    // packed records alike) stays valid until `slots` further collect() calls have been made, whatever is submitted meanwhile.
    const BatchResult& collect();
    int in_flight() const { return (int)(submitted_ - collected_); }

    static void* alloc_pinned(size_t bytes);
    static void free_pinned(void* p);

private:
    struct Slot {
        vslam_batch_out out{};          // shared image buffers + this slot's lists
        uint8_t* d_frames = nullptr;
        vslam_kp* d_hpacked = nullptr;
        vslam_point* d_ppacked = nullptr;
        uint64_t* d_off = nullptr;      // [3][batch + 1]: harris, dog, oriented
        uint64_t* h_off = nullptr;      // pinned
        uint32_t* h_cnt = nullptr;      // pinned [4][batch]: harris, dog, oriented counts, oriented survivors
        vslam_point* d_opacked = nullptr;
        float* d_dpacked = nullptr;
        vslam_point* h_opacked = nullptr;  // pinned
        float* h_dpacked = nullptr;
        uint8_t* h_defined = nullptr;
        vslam_kp* h_hpacked = nullptr;  // pinned
        vslam_point* h_ppacked = nullptr;
        void *up_done = nullptr, *comp_done = nullptr, *down_done = nullptr, *det_done = nullptr;  // hipEvent_t
        void* totals_read = nullptr;    // caller's event (hold_totals_until), not owned
        int n = 0;
        BatchResult res;
        // collect() copies the offsets and counts out of the pinned buffers above: the slot's NEXT submit() overwrites those
        // asynchronously (pack stream) while the caller may still be reading the result
        std::vector<uint64_t> res_off;
        std::vector<uint32_t> res_cnt;
    };
    struct Pipe {
        vslam_ctx* ctx = nullptr;
        void* stream = nullptr;         // hipStream_t
        vslam_batch_out img{};          // image buffers only
    };
    int next_pipe() const { return (int)(batches_ % pipes_.size()); }
    Slot& begin_batch(int slot, int n);  // binds the slot to the next pipeline
    void init(const Options& opt);
    void release();
    vslam_params p_{};
    vslam_batch_layout L_{};
    Options opt_;
    std::vector<Pipe> pipes_;
    int last_pipe_ = 0, last_slot_ = 0;
    uint64_t batches_ = 0;           // detect_device + submit calls so far
    vslam_ctx* ctx_pack_ = nullptr;  // a context on the pack stream: the list packing of batch k runs beside the kernels of batch k+1
    void *up_ = nullptr, *down_ = nullptr, *pack_ = nullptr;  // hipStream_t
    std::vector<Slot> slots_;
    std::vector<void*> dev_allocs_, pinned_allocs_;
    uint64_t *d_totals_ = nullptr, *d_totals_all_ = nullptr;
    size_t packed_cap_h_ = 0, packed_cap_p_ = 0, packed_cap_o_ = 0, packed_cap_d_ = 0;  // records
    uint64_t submitted_ = 0, collected_ = 0;
};

}  // namespace vslam
