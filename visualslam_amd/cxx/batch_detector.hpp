// C++17 host of the THROUGHPUT path: vslam_detect_batch_dev (BASELINE config 4 / 5) behind a class, so
// that the batched detection the benchmark measures is driven from C++ through the C ABI like the
// per-image API of vslam_cxx.hpp (north star: "host code stays C++ calling HIP through a thin C-ABI
// layer").  The reference's callers are C++ main()s that feed one image at a time
// (Harris_corners.cpp:146-193, Diff_of_Gauss.cpp:727-877); a camera stream is the same loop over frames,
// and this class is that loop for batches of frames on one GPU:
//
//   device-resident   detect_device(d_frames, stride, n): frames already in HBM, outputs stay in HBM
//                     (device_outputs()), asynchronous on stream() - what bench.py times;
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
};

struct BatchResult {
    int n_frames = 0;
    const uint64_t* harris_offsets = nullptr;  // [n_frames + 1] record offsets into `harris`
    const uint64_t* dog_offsets = nullptr;     // [n_frames + 1]
    const vslam_kp* harris = nullptr;          // packed, pinned host memory
    const vslam_point* dog = nullptr;
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
    vslam_ctx* context() const { return ctx_; }
    void* stream() const { return compute_; }  // hipStream_t of the kernels

    // ---- device-resident
    // Harris + DoG over n frames in HBM (frame f at d_frames + f * frame_stride).  Asynchronous on stream().
    void detect_device(const uint8_t* d_frames, size_t frame_stride, int n);
    // The device buffers detect_device (and the most recent submit) wrote: response, nms_mask, pyramid,
    // extrema_bits shared by all batches; lists of slot 0 for detect_device.
    const vslam_batch_out& device_outputs() const { return slots_[0].out; }
    // {harris, dog} totals of the last detect_device() / submit() as two uint64 in device memory
    // (vslam_count_totals_dev; one pair per slot): the send buffer of the count all-gather, valid on stream()
    const uint64_t* device_totals() const { return d_totals_; }
    void sync();

    // ---- host-fed pipeline
    // Enqueues upload + detection + list packing of one batch.  host_frames: n dense frames; pinned memory
    // (alloc_pinned) for an asynchronous upload.  Throws if `slots` batches are already in flight.
    void submit(const uint8_t* host_frames, int n);
    // Waits for the oldest batch in flight, downloads its lists and returns them.  The result stays valid
    // until `slots` further collect() calls have been made.
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
        int n = 0;
        BatchResult res;
    };
    void init(const Options& opt);
    void release();
    void run_on_slot(Slot& s, const uint8_t* d_frames, size_t stride, int n);
    vslam_params p_{};
    vslam_batch_layout L_{};
    Options opt_;
    vslam_ctx* ctx_ = nullptr;
    vslam_ctx* ctx_pack_ = nullptr;  // a second context on the pack stream: the list packing of batch k runs beside the kernels of batch k+1
    void *compute_ = nullptr, *up_ = nullptr, *down_ = nullptr, *pack_ = nullptr;  // hipStream_t
    std::vector<Slot> slots_;
    std::vector<void*> dev_allocs_, pinned_allocs_;
    uint64_t *d_totals_ = nullptr, *d_totals_all_ = nullptr;
    size_t packed_cap_h_ = 0, packed_cap_p_ = 0, packed_cap_o_ = 0, packed_cap_d_ = 0;  // records
    uint64_t submitted_ = 0, collected_ = 0;
};

}  // namespace vslam
