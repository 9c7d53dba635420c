/*
 * vslam.h -- C ABI of the MI355X-native keypoint-detection front end.
 *
 * The reference (JacobYoung115/VisualSLAM, KeyPointDetection/) has no FFI layer: its hot
 * path is plain C++ functions over cv::Mat.  This header is the
 * drop-in boundary a maintainer binds instead: every entry point names the reference
 * function / OpenCV call site it replaces (paths relative to KeyPointDetection/).
 * The C++ mirror of the reference signatures (HarrisCorner, NMS2, GaussPyramid,
 * initialKeypointDetection ...) sits on top of this ABI in visualslam_amd/cxx/.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes, int status return (0 = VSLAM_OK, negative =
 *     error); no exceptions cross the ABI; vslam_last_error(ctx) gives the message.
 *   - Images are row-major, `step` = bytes between rows (cv::Mat::step).
 *   - "host" entry points take host pointers, copy in/out and are synchronous.
 *     "_dev" entry points take DEVICE pointers, are asynchronous on the context's HIP
 *     stream and never synchronise: they are the path bench.py measures.
 *   - One context per GPU/stream.  A context is not thread-safe; distinct contexts are
 *     independent.  All kernels are hand-written HIP for gfx950; there is no CPU
 *     fallback: without a GPU every compute entry point fails with VSLAM_ERR_HIP.
 *   - Semantics ("intended" vs "literal" for the reference's defects) follow SURVEY.md
 *     Appendix B and are restated per function below.
 */
#ifndef VSLAM_H
#define VSLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSLAM_VERSION 200 /* 0.2.0: vslam_batch_out carries buffer sizes (round 3) */

enum {
    VSLAM_OK = 0,
    VSLAM_ERR_INVALID = -1,     /* bad argument (null pointer, non-positive size, even window ...) */
    VSLAM_ERR_HIP = -2,         /* a HIP runtime call or kernel launch failed (incl. no GPU) */
    VSLAM_ERR_NOMEM = -3,       /* device or host allocation failed */
    VSLAM_ERR_UNSUPPORTED = -4, /* parameter combination not implemented */
    VSLAM_ERR_RANGE = -5        /* octave / level index out of range */
};

#define VSLAM_MAX_OCTAVES 16
#define VSLAM_NUM_LEVELS 6 /* scaleSamples_ + 3, include/src/GaussPyramid/GaussPyramid.hpp:65-66 */
#define VSLAM_NUM_DOGS 5   /* include/src/GaussPyramid/GaussPyramid.cpp:193 */

typedef struct vslam_ctx vslam_ctx;
typedef struct vslam_pyramid vslam_pyramid;

/* Binary-compatible with SLAM::point (Diff_of_Gauss.cpp:27-35): six ints, 24 bytes.
 * row/col are in 1-padded coordinates exactly as the reference produces them (:289). */
typedef struct {
    int32_t row, col, value, padding, octave, level;
} vslam_point;

/* Harris keypoint: NMS2 survivor whose 8-bit view exceeds 253 (Harris_corners.cpp:139,181). */
typedef struct {
    int32_t row, col;
    float response;
} vslam_kp;

/* ---------------------------------------------------------------- lifecycle */

int vslam_version(void);
const char* vslam_status_string(int status);

/* device: HIP device ordinal.
 * stream: the hipStream_t every call of this context is enqueued on:
 *   - a stream handle of the caller: work is ordered with whatever else the caller enqueues there;
 *   - VSLAM_STREAM_LEGACY: the device's legacy NULL stream (the library then launches on stream 0).
 *     A caller whose own work runs on the NULL stream (handle 0, e.g. torch's default stream) must
 *     pass THIS, not NULL;
 *   - NULL: the context creates and owns a non-blocking stream.  It is NOT ordered against the NULL
 *     stream or any other stream: device buffers handed to vslam_detect_batch_dev must be complete
 *     before the call, and its results are complete only after vslam_ctx_sync.
 * The batched path forks internal side streams from this stream and joins them back before it
 * returns, so ordering on this stream covers all of its work. */
#define VSLAM_STREAM_LEGACY ((void*)1)
int vslam_ctx_create(int device, void* stream, vslam_ctx** out);
int vslam_ctx_destroy(vslam_ctx* ctx);
int vslam_ctx_sync(vslam_ctx* ctx);
/* The batched path runs its Harris chain and its scans / lists on two side streams.  When those YIELD to the octave kernels
 * (lowest stream priority, vslam_ctx_set_side_stream_priority below) HIP's choice of hardware queue decides how they fare:
 * on an unlucky queue they crawl (up to -20 %).  A host that wants the library to look for a better
 * pair of yielding streams opts in with vslam_ctx_tune_side_streams(ctx, 1) BEFORE the context's first batch call
 * (VSLAM_ERR_UNSUPPORTED afterwards; off by default; VSLAM_STREAM_TUNER=1 does the same for contexts created afterwards).
 * Opting in also selects yielding side streams (as set_side_stream_priority(ctx, 1) would), unless a level has been pinned:
 * at any other level there is nothing to compare, the tuner ends at its first call (state 2, pair 0) and the join watchdog
 * below runs as usual.  The context then times its 2nd to 5th full-size batch call of one shape on three candidate pairs of
 * side streams and adopts the fastest at the first later call that finds those calls finished (an event query: no call
 * ever waits on the host, and nothing is timed while the context's stream is being captured; results never depend on the
 * pair); the join watchdog starts once the comparison has ended.
 * vslam_ctx_side_stream_report: the index of the pair in use (0 = the first created) and the state of the comparison
 * (0 off or not started, 1 measuring, 2 decided).  Diagnostic only. */
int vslam_ctx_tune_side_streams(vslam_ctx* ctx, int on);
/* Priority of the batched path's two side streams (Harris chain; scans and lists).  Default (low = 0): the context stream's
 * own priority - within 3.5 % of the best schedule for every hardware-queue layout measured.  low = 1:
 * the device's LOWEST priority, so that the side work yields to the octave kernels - 2-3 % faster when every stream has a
 * hardware queue to itself, up to 18 % slower when one does not: for a host that controls GPU_MAX_HW_QUEUES and its own
 * stream count (vslam::BatchDetector::Options::yielding_side_streams; Stream's host-fed mode).  Before the context's first
 * batch call only (VSLAM_ERR_UNSUPPORTED afterwards); VSLAM_SIDE_PRIORITY=low|main sets the default of new contexts.
 * Results never depend on it. */
int vslam_ctx_set_side_stream_priority(vslam_ctx* ctx, int low);
/* Always on (VSLAM_JOIN_WATCH=0 disables; idle while a comparison of the tuner above runs and during stream captures):
 * the first full-size batch calls of a context measure how long the context's stream waits at the end of the call for the
 * side streams (three events on the stream, read by a later call once complete - no host wait).  Three calls whose median
 * wait exceeds the level's limit (3 % of the call with yielding side streams = level 0, 10 % at the default level 1) start a
 * trial of the next level - 1: side streams at the context stream's priority, 2: side work on the context's stream itself -
 * which is kept only if its fastest call is 1 % faster than the previous level's; then the watch ends (done).  Results
 * never depend on the level.  last_lag_fraction: the most recent measurement (-1: none yet).
 * CAPTURE NOTE: the watch (like the tuner) calls hipEventQuery / hipEventElapsedTime and records timing events on the
 * context's stream.  It checks the capture status of THAT stream only: a process that captures in GLOBAL mode on another
 * stream (torch.cuda.graph's default) while this context makes un-captured batch calls should switch the watch off for the
 * context - vslam_ctx_set_join_watch(ctx, 0), or VSLAM_JOIN_WATCH=0 for every context - or pin a level. */
int vslam_ctx_join_watch_report(const vslam_ctx* ctx, int* level, int* done, float* last_lag_fraction);
/* on = 0: this context takes no more measurements and stays at its current level; on = 1 (the default) resumes. */
int vslam_ctx_set_join_watch(vslam_ctx* ctx, int on);
/* Pins the side-stream level of a context and ends its watch: 0 yielding side streams, 1 side streams at the context
 * stream's priority, 2 no side streams (every kernel of a batch call on the context's stream, in order).  Before the
 * context's first batch call only (VSLAM_ERR_UNSUPPORTED afterwards).  Results never depend on the level. */
int vslam_ctx_pin_side_streams(vslam_ctx* ctx, int level);
int vslam_ctx_side_stream_report(const vslam_ctx* ctx, int* pair, int* state);
/* Two batches in flight: a second context (own stream, own output buffers) whose batch starts when `leader`'s most
 * recent vslam_detect_batch_dev call is past its octave-0 kernels - the long, issue-bound part - instead of beside
 * them.  The coarse octaves, scans and list kernels that follow are short and leave issue slots idle; the follower's
 * octave 0 can fill them (+3..5 % frames/s measured for two contexts taking alternate batches, each following the
 * other, when their streams get hardware queues of their own; see DESIGN section 5.4 for when they do not).
 * Enqueues one event wait on ctx's stream; a no-op if the leader has not run a batch yet.
 * visualslam_amd/cxx/batch_detector.hpp (Options::pipelines) uses it. */
int vslam_ctx_follow(vslam_ctx* ctx, const vslam_ctx* leader);
/* OPT-IN, off by default (environment: VSLAM_MX=1 switches it on for every context created afterwards): the
 * Gaussian levels of the LDS-tiled octaves (GaussVector + Diff_of_Gauss, GaussPyramid.cpp:166-200) as banded
 * matrix products on the matrix cores (v_mfma_i32_32x32x32_i8) instead of packed dot instructions.  Results are
 * bit-identical (exact integer arithmetic either way); BASELINE.json's north star rules MFMA out for this path, so
 * the default, and everything bench.py reports as `value` / `roofline`, stays on the dot kernels. */
int vslam_ctx_set_matrix_path(vslam_ctx* ctx, int on);
int vslam_ctx_get_matrix_path(const vslam_ctx* ctx);
/* Which OpenCV the f32 stages reproduce bit for bit.  The reference's f32 arithmetic runs inside OpenCV: cv::phase
 * (GaussPyramid.cpp:96) and GaussianBlur on CV_32F windows (Diff_of_Gauss.cpp:348, :616-618).  OpenCV dispatches both at run
 * time: its SSE2 baseline rounds every product and sum, its AVX2 + FMA3 code fuses the multiply-adds of fastAtan32f's
 * polynomial and of the separable filter's row / column passes.  on = 0 (default): the baseline; on = 1 (environment:
 * VSLAM_F32_FUSED=1 for contexts created afterwards): the fused form - in vslam_pyramid_get_gradients' orientation image,
 * vslam_filter_keypoints, vslam_sift_descriptors and the batched orient / descriptor stages.  The CPU oracle carries the same
 * switch (vo_set_fma_variant) and the parity tests hold bit for bit under either; which one a given OpenCV build computes is
 * what tools/pin_with_opencv.sh reports.  Between the two (profiles/r06_fma_risk.json, and on the GPU at scale
 * tests/test_gpu_f32_variant.py): orientation values differ by at most one unit in the last place, no histogram bin and no
 * oriented point changes, descriptor entries differ by at most 5e-7.  The integer rows (Harris, pyramid, DoG, extrema,
 * localization) do not depend on it. */
int vslam_ctx_set_f32_fused(vslam_ctx* ctx, int on);
int vslam_ctx_get_f32_fused(const vslam_ctx* ctx);
const char* vslam_last_error(const vslam_ctx* ctx);

/* ------------------------------------------ host-side parameter helpers (no GPU) */

/* Kernel width cv::GaussianBlur derives for CV_8U when ksize = Size(0,0):
 * cvRound(sigma*6+1)|1  (call site GaussPyramid.cpp:177). */
int vslam_gauss_ksize_u8(double sigma);
/* The 8.8 fixed-point taps cv::GaussianBlur uses on CV_8U (sum == 256); taps[n]. */
int vslam_gauss_taps_q8(int n, double sigma, uint16_t* taps);
/* GaussPyramid::calculateSigma(octave, level), GaussPyramid.cpp:160-162. */
double vslam_sigma_at(double sigma0, int octave, int level);
/* GaussPyramid::calculateNumOctaves, GaussPyramid.cpp:150-152. */
int vslam_auto_num_octaves(int rows, int cols);
/* Size cv::resize(..., 0.5, 0.5, INTER_NEAREST) produces (GaussPyramid.cpp:126). */
void vslam_half_size(int rows, int cols, int* out_rows, int* out_cols);
/* Lattice of initialKeypointDetection (Diff_of_Gauss.cpp:267-268): sites
 * i = pad, pad+window, ... < rows (likewise cols). */
void vslam_extrema_lattice(int rows, int cols, int window, int* lat_rows, int* lat_cols);

/* ------------------------------------------------- host-buffer primitives (GPU) */

/* cv::GaussianBlur(src, dst, Size(ksize,ksize) or Size(0,0), sigma, 0, BORDER_DEFAULT)
 * on CV_8U: Harris_corners.cpp:158, GaussPyramid.cpp:177. */
int vslam_gaussian_blur_u8(vslam_ctx* ctx, const uint8_t* src, int rows, int cols, size_t step, int ksize,
                           double sigma, uint8_t* dst, size_t dst_step);
/* cv::Sobel(src, dst, CV_32F, dx, dy, ksize=1, 1, 0, BORDER_DEFAULT):
 * Harris_corners.cpp:163-164, GaussPyramid.cpp:87,90.  (dx,dy) = (1,0) or (0,1). */
int vslam_sobel_k1_u8_f32(vslam_ctx* ctx, const uint8_t* src, int rows, int cols, size_t step, int dx, int dy,
                          float* dst, size_t dst_step);
/* cv::resize(src, dst, Size(), 2, 2, INTER_LINEAR), GaussPyramid.cpp:110. */
int vslam_resize_linear2x_u8(vslam_ctx* ctx, const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst,
                             size_t dst_step);
/* cv::resize(src, dst, Size(), 0.5, 0.5, INTER_NEAREST), GaussPyramid.cpp:126. */
int vslam_resize_nearest_half_u8(vslam_ctx* ctx, const uint8_t* src, int rows, int cols, size_t step,
                                 uint8_t* dst, size_t dst_step);
/* cv::convertScaleAbs(src f32, dst u8), Harris_corners.cpp:176,181, as the reference's x86-64 OpenCV
 * build evaluates it: round_half_even(|x|) saturated to 255 for |x| < 2^31; NaN and |x| >= 2^31 give 0
 * (cvRound = cvtss2si / cvtps2dq returns INT_MIN there and saturate_cast<uchar> maps it to 0). */
int vslam_convert_scale_abs_f32(vslam_ctx* ctx, const float* src, int rows, int cols, size_t step, uint8_t* dst,
                                size_t dst_step);

/* --------------------------------------------------------------- Harris (host) */

/* Mat HarrisCorner(Mat& Ix, Mat& Iy), Harris_corners.cpp:31-68 (with StructureMatrix
 * :10-29): arbitrary f32 gradients, literal f32 accumulation order, double determinant.
 * Intended rows x cols output (Appendix B-1).  Reference literals: k=0.04f, window=3. */
int vslam_harris_from_grad_f32(vslam_ctx* ctx, const float* ix, const float* iy, int rows, int cols,
                               size_t step, float k, int window, float* resp, size_t resp_step);
/* Front end of Harris main(): GaussianBlur 3x3 -> Sobel x,y -> HarrisCorner
 * (Harris_corners.cpp:158-172) as ONE fused kernel over the 8-bit frame. window must be 3. */
int vslam_harris_response_u8(vslam_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step, float k,
                             int window, float* resp, size_t resp_step);
/* Mat NonMaximumSuppression(Mat& response, int windowSize), Harris_corners.cpp:70-81:
 * mask(255/0) = response > max(neighbours, centre excluded).  windowSize odd. */
int vslam_nms_strict_u8(vslam_ctx* ctx, const uint8_t* src, int rows, int cols, size_t step, int window,
                        uint8_t* mask, size_t mask_step);
int vslam_nms_strict_f32(vslam_ctx* ctx, const float* src, int rows, int cols, size_t step, int window,
                         uint8_t* mask, size_t mask_step);
/* Mat NMS2(Mat& response, int windowSize), Harris_corners.cpp:83-129: half-open window
 * (literal, Appendix B-4), f32 map out (intended, B-3).  true_max (may be NULL) receives
 * the value the reference prints at :127. */
int vslam_nms2_f32(vslam_ctx* ctx, const float* resp, int rows, int cols, size_t step, int window, float* out,
                   size_t out_step, float* true_max);
/* Whole Harris executable minus display: response -> NMS2(5) -> 8-bit view > 253
 * (Harris_corners.cpp:158-182).  Row-major keypoint list; *count = total found (may
 * exceed cap; only cap entries are written). */
int vslam_harris_keypoints_u8(vslam_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step, float k,
                              vslam_kp* out, size_t cap, size_t* count);

/* ------------------------------------------------------------ DoG pyramid (host) */

typedef struct {
    int n_octaves;
    int n_levels; /* 6 */
    int n_dogs;   /* 5 */
    double sigma0;
    int rows[VSLAM_MAX_OCTAVES], cols[VSLAM_MAX_OCTAVES];
    double sigma[VSLAM_MAX_OCTAVES][VSLAM_NUM_LEVELS];
    int ksize[VSLAM_MAX_OCTAVES][VSLAM_NUM_LEVELS];
} vslam_pyramid_info;

/* GaussPyramid(Mat& img, int numOctaves, double sigma) -> createPyramid
 * (GaussPyramid.hpp:17, GaussPyramid.cpp:106-131): 2x bilinear base, per octave 6
 * Gaussians (each blurred from the octave base, :166-185) and 5 saturating DoGs
 * (:191-200), next base = Gaussian[3] decimated (:123-126).  The images stay in HBM;
 * the getters copy one image to the host.  n_octaves <= 0 selects the automatic count
 * of the second constructor (GaussPyramid.hpp:18-21). */
int vslam_pyramid_build_u8(vslam_ctx* ctx, const uint8_t* img, int rows, int cols, size_t step, int n_octaves,
                           double sigma0, vslam_pyramid** out);
int vslam_pyramid_destroy(vslam_pyramid* pyr);
int vslam_pyramid_get_info(const vslam_pyramid* pyr, vslam_pyramid_info* out);
/* octaveImage / octaveBlur / octaveDiff (GaussPyramid.hpp:29,31,32). */
int vslam_pyramid_get_base(const vslam_pyramid* pyr, int octave, uint8_t* dst, size_t dst_step);
int vslam_pyramid_get_gauss(const vslam_pyramid* pyr, int octave, int level, uint8_t* dst, size_t dst_step);
int vslam_pyramid_get_dog(const vslam_pyramid* pyr, int octave, int level, uint8_t* dst, size_t dst_step);

/* GaussPyramid::processGradients for one Gaussian level (GaussPyramid.cpp:65-104; getters
 * octaveGradX / octaveGradY / octaveGradMag / octaveGradOrient, GaussPyramid.hpp:33-36):
 * Sobel x, Sobel y (ksize 1), cv::magnitude, cv::phase in degrees (OpenCV's fastAtan2
 * polynomial, ~0.01 deg from atan2).  Computed on demand from the HBM-resident Gaussian
 * (the reference materialises all 96 bytes per pyramid pixel in its constructor).  Any of the
 * four CV_32F destinations may be NULL; dst_step in bytes. */
int vslam_pyramid_get_gradients(const vslam_pyramid* pyr, int octave, int level, float* grad_x, float* grad_y,
                                float* mag, float* orient, size_t dst_step);

/* void initialKeypointDetection(vector<SLAM::point>&, GaussPyramid&, int octave, int
 * windowSize), Diff_of_Gauss.cpp:254-297, up to the FeaturePointLocalization call.
 * Literal stride-`window` lattice and half-open window (Appendix B-7), padded
 * coordinates (B-8).
 *   bits (may be NULL): candidate bitmask, 3 levels x lat_rows x words_per_row uint64
 *     words, words_per_row = (lat_cols+63)/64, bit (lj & 63) of word
 *     [((level-1)*lat_rows + li)*words_per_row + lj/64].
 *   out/cap/count: candidates with value >= min_contrast in the reference's loop order
 *     (level, i, j); *count = total (may exceed cap). */
int vslam_dog_extrema(vslam_ctx* ctx, const vslam_pyramid* pyr, int octave, int window, int min_contrast,
                      uint64_t* bits, vslam_point* out, size_t cap, size_t* count);

/* EXTENSION, not a reference function (note under the table): the dense
 * 3x3x3 scale-space test the north star's wording names.  The rule of Diff_of_Gauss.cpp:282-287
 * (candidate iff the value equals the minimum or the maximum of its window, ties included) and the
 * replicate border of padOctave (:260), applied to EVERY pixel of DoG levels 1..3 with the full
 * 3x3x3 neighbourhood (the reference tests a half-open 2x2x3 window on a stride-3 lattice:
 * vslam_dog_extrema above is that, and is what parity is judged on).
 *   bits (may be NULL): 3 levels x rows x words_per_row uint64 words, words_per_row = (cols+63)/64,
 *     bit (x & 63) of word [((level-1)*rows + y)*words_per_row + x/64].
 *   out/cap/count: candidates with value >= min_contrast in (level, y, x) order as
 *     SLAM::point(y+1, x+1, value, 1, octave, level) - padded coordinates like :289;
 *     *count = total (may exceed cap). */
int vslam_dog_extrema_dense(vslam_ctx* ctx, const vslam_pyramid* pyr, int octave, int min_contrast, uint64_t* bits,
                            vslam_point* out, size_t cap, size_t* count);

/* The same function through its call to FeaturePointLocalization (Diff_of_Gauss.cpp:290,
 * :223-251): exactly the points the reference appends to `keypoints`, in its order, with the
 * value rewritten at :246.  The contrast test is evaluated for EVERY candidate with the
 * reference's own arithmetic (A*A^T is singular; cv::invert's closed form decides, see
 * kernels_localize.hip.h) - no min_contrast shortcut. */
int vslam_dog_keypoints(vslam_ctx* ctx, const vslam_pyramid* pyr, int octave, int window, vslam_point* out,
                        size_t cap, size_t* count);
/* bool FeaturePointLocalization(vector<Mat>& dogs_padded, vector<SLAM::point>&, int level,
 * SLAM::point&), Diff_of_Gauss.cpp:223-251, for n independent candidates.  diffs = n x
 * (d_x, d_y, d_scale, value) as read at :226-228 and from point.value; keep[i] = the function's
 * return value, value[i] = point.value afterwards (unchanged when not kept). */
int vslam_localize_points(vslam_ctx* ctx, const int* diffs, size_t n, int* keep, int* value);

/* void filterKeypoints(GaussPyramid&, int octave, vector<SLAM::point>& keypoints,
 * vector<SLAM::point>& reducedKeypoints), Diff_of_Gauss.cpp:301-372, with computeEdgeResponse
 * (:79-109) and orientationHistogram (:112-133): edge rejection tr^2/det < 12.1 on the level's
 * Sobel gradients, then the 36-bin histogram of the 16x16 window of the 8-padded magnitude /
 * orientation images, magnitudes weighted by GaussianBlur(sigma = 1.5 * sigma(octave, level))
 * evaluated as the reference does on a non-isolated ROI (parent pixels, reflect-101 at the
 * parent's edge).  kps: the octave's keypoints (vslam_dog_keypoints output: 1-padded
 * coordinates, level 1..3).  out: one SLAM::point{row, col, angle = bin*10, 0, octave, level}
 * per histogram bin above 0.8 * max, in keypoint order then ascending bin.  *count = total
 * (may exceed cap).  A keypoint the reference would throw on (level outside 0..5, window
 * outside the padded image, other octave) gives VSLAM_ERR_RANGE. */
int vslam_filter_keypoints(vslam_ctx* ctx, const vslam_pyramid* pyr, int octave, const vslam_point* kps, size_t n,
                           vslam_point* out, size_t cap, size_t* count);
/* float computeEdgeResponse(const SLAM::point&, const Mat& grad_x, const Mat& grad_y),
 * Diff_of_Gauss.cpp:79-109, for n points whose gradient windows the caller has gathered in the
 * reference's loop order (:93-94): gx_windows / gy_windows = n x window_elems f32. */
int vslam_edge_response_windows(vslam_ctx* ctx, const float* gx_windows, const float* gy_windows, int window_elems,
                                size_t n, float* response);

/* void StructureMatrix(Mat& M, Mat& Ix, Mat& Iy, int padding, int i, int j), Harris_corners.cpp:10-29,
 * for n pixels whose (2*padding+1)^2 gradient windows the caller has gathered in the
 * reference's loop order (:16-17): sums = n x (Ix2, IxIy, Iy2) = M(0,0), M(0,1) = M(1,0), M(1,1). */
int vslam_structure_matrix_windows(vslam_ctx* ctx, const float* gx_windows, const float* gy_windows, int window_elems,
                                   size_t n, float* sums);

/* ------------------------------------------- SIFT descriptor stage (host buffers) */

/* const Point2f Rotation::cos_sin_of_angle(float theta, bool degrees = true), rotation.cpp:9-17, with
 * convertToRadians (:5-7): theta * (CV_PI / 180.0f) in double, narrowed to float, cos / sin of it
 * (host libm, like the reference).  Pure host computation. */
void vslam_cos_sin_deg(float theta_deg, float* cos_out, float* sin_out);
/* std::vector<Point2i> Rotation::getRotatedWindowPoints(Mat& I, const Point2i& center, int windowSize,
 * float theta, bool degrees = true), rotation.cpp:112-130 (with rotate_pt_CW :19-27): the
 * (windowSize+1)^2 points of the square around `center` rotated clockwise, rows outer;
 * xy[2q] = x, xy[2q+1] = y.  The Mat argument of the reference is unused.  Pure host computation. */
int vslam_rotated_window_points(int cx, int cy, int window, float theta_deg, int32_t* xy);
/* void SIFT(vector<SLAM::point>& reducedKeypoints, vector<vector<float>>& featureDescriptors_vec,
 * GaussPyramid&, int octave), Diff_of_Gauss.cpp:561-693, with rotateImageSection (:528-559): one
 * 128-float descriptor per oriented keypoint of the octave (vslam_filter_keypoints output: value =
 * angle in degrees), desc = n x 128.  Literal reference behaviour, including the stride-16 walk of
 * the 17 x 17 point list (:545) and Mat::at<>(point.x, point.y) with x as the ROW (:549-554) on the
 * 20-padded level images.  Mat::at checks nothing, so a sample is linear element
 * x * (cols + 40) + y of the padded Mat; a keypoint whose 256 samples all lie inside that buffer is
 * DEFINED (always the case for square and portrait octaves away from the last rows), otherwise the
 * reference reads foreign memory: defined[k] = 0 and a zero descriptor.  defined may be NULL, in
 * which case any undefined keypoint makes the call return VSLAM_ERR_RANGE (after filling desc).
 * A flat window gives the reference's all-NaN descriptor (0 / 0 at :661).
 * Histogram bins: the reference indexes histo.at((int)(orientation * 8/360.f)) (:126), which throws for
 * an orientation of exactly 360.0; the kernels clamp the bin to 0..7 instead.  The gradients here come
 * from integer Sobel differences, for which cv::phase never returns 360.0, so the two agree on every
 * input this entry point can be given; the clamp only keeps an impossible value from indexing outside LDS. */
int vslam_sift_descriptors(vslam_ctx* ctx, const vslam_pyramid* pyr, int octave, const vslam_point* oriented, size_t n,
                           float* desc, uint8_t* defined);
/* featureDescriptors.dat, Diff_of_Gauss.cpp:837-863: int32 {n, 128, 24} (24 = sizeof(std::vector<float>)
 * on LP64, what `sizeof(featureDescriptors_vec.front())` writes) followed by n x 128 float32,
 * native byte order.  Pure host computation. */
int vslam_descriptor_file_write(const char* path, const float* desc, size_t n);

/* ------------------------------------------- device-resident batched detection */

typedef struct {
    int rows, cols;      /* frame size */
    int n_octaves;       /* DoG octaves (reference literal 4, Diff_of_Gauss.cpp:742); 0 = no DoG */
    double sigma0;       /* 1.6, Diff_of_Gauss.cpp:743 */
    float harris_k;      /* 0.04f, Harris_corners.cpp:36 */
    int do_harris;       /* run the Harris path */
    int extrema_window;  /* 3, Diff_of_Gauss.cpp:772 */
    int min_contrast;    /* list threshold on the 8-bit DoG value; 8 */
    int localize;        /* 1: dog_points = FeaturePointLocalization survivors (vslam_dog_keypoints),
                          * min_contrast unused; 0 (default): candidates with value >= min_contrast */
    int orient;          /* 1: also run filterKeypoints (Diff_of_Gauss.cpp:301-372) on every frame's
                          * keypoint list -> oriented_points / oriented_counts; needs localize = 1 */
    uint32_t harris_cap; /* per-frame capacity of the Harris keypoint list */
    uint32_t dog_cap;    /* per-frame capacity of the DoG point list */
    uint32_t oriented_cap; /* per-frame capacity of the oriented keypoint list (and of the
                            * edge-test survivors it is made from) */
    int extrema_dense;   /* EXTENSION (0 = the reference's lattice test, the default and the parity path): 1 runs the
                          * dense 3x3x3 scale-space test of vslam_dog_extrema_dense on every pixel of DoG levels 1..3
                          * of every frame instead.  extrema_bits then holds, per octave, 3 levels x rows x
                          * ceil(cols/64) words (layout.lat_rows / lat_cols / lat_words report rows / cols / words per
                          * row) and dog_points the candidates with value >= min_contrast in (octave, level, y, x)
                          * order.  Needs extrema_window = 3, localize = 0, orient = 0. */
} vslam_params;

/* Byte layout of the per-frame output blocks, so that a caller can allocate them. */
typedef struct {
    int n_octaves;
    int rows[VSLAM_MAX_OCTAVES], cols[VSLAM_MAX_OCTAVES];
    int lat_rows[VSLAM_MAX_OCTAVES], lat_cols[VSLAM_MAX_OCTAVES], lat_words[VSLAM_MAX_OCTAVES];
    /* pyramid block of one frame: for octave o, Gaussian l at octave_offset[o] + l*P_o and
     * DoG l at octave_offset[o] + (6+l)*P_o, P_o = rows[o]*pitch[o]; row r of a plane starts at
     * r*pitch[o].  pitch[o] = cols[o] rounded up to a multiple of 16 (so it equals cols[o] for the
     * usual frame sizes); the padding bytes are unspecified. */
    int pitch[VSLAM_MAX_OCTAVES];
    size_t octave_offset[VSLAM_MAX_OCTAVES];
    size_t pyramid_frame_bytes;
    /* candidate bitmask block of one frame: octave o starts at word bits_offset[o] */
    size_t bits_offset[VSLAM_MAX_OCTAVES];
    size_t bits_frame_words;
    /* algorithmic HBM bytes per frame: input + response + mask +
     * Gaussian and DoG stacks */
    size_t algorithmic_bytes_harris, algorithmic_bytes_dog;
} vslam_batch_layout;

/* Device pointers; any may be NULL to skip that output (it is then neither computed for its own sake
 * nor written).  Every pointer travels with the size of the buffer behind it: `x_bytes` is the number of
 * bytes the caller allocated at `x`.  vslam_detect_batch_dev checks each non-NULL buffer against what
 * n_frames frames need (vslam_batch_out_required fills in exactly those numbers) and returns
 * VSLAM_ERR_INVALID - before anything is launched - if one is too small or struct_size is not
 * sizeof(vslam_batch_out); a kernel never writes past a size stated here. */
typedef struct {
    size_t struct_size;      /* = sizeof(vslam_batch_out) */
    float* response;         /* [n][rows][cols]  HarrisCorner output */
    size_t response_bytes;
    uint8_t* nms_mask;       /* [n][rows][cols]  NonMaximumSuppression(8-bit view, 3) */
    size_t nms_mask_bytes;
    float* nms2;             /* [n][rows][cols]  NMS2(response, 5) map (optional) */
    size_t nms2_bytes;
    vslam_kp* harris_kps;    /* [n][harris_cap] */
    size_t harris_kps_bytes;
    uint32_t* harris_counts; /* [n] totals (may exceed cap) */
    size_t harris_counts_bytes;
    uint8_t* pyramid;        /* [n][pyramid_frame_bytes], 16-byte aligned, planes pitched (layout.pitch) */
    size_t pyramid_bytes;
    uint64_t* extrema_bits;  /* [n][bits_frame_words] */
    size_t extrema_bits_bytes;
    vslam_point* dog_points; /* [n][dog_cap], order (octave, level, i, j) */
    size_t dog_points_bytes;
    uint32_t* dog_counts;    /* [n] totals (may exceed cap) */
    size_t dog_counts_bytes;
    vslam_point* oriented_points; /* [n][oriented_cap] filterKeypoints output: {row, col, angle, 0, octave, level},
                                   * order (octave, keypoint, histogram bin); params.orient = 1 */
    size_t oriented_points_bytes;
    uint32_t* oriented_counts;    /* [n] oriented points of the EVALUATED survivors (may exceed cap): the true total
                                   * iff oriented_survivors[f] <= oriented_cap */
    size_t oriented_counts_bytes;
    uint32_t* oriented_survivors; /* [n] optional: keypoints of the frame that pass the edge test (Diff_of_Gauss.cpp:336).
                                   * Only the first oriented_cap of them (list order) get their histogram evaluated, so
                                   * oriented_survivors[f] > oriented_cap flags a truncated frame */
    size_t oriented_survivors_bytes;
    float* descriptors;           /* [n][oriented_cap][128] optional: SIFT() descriptors (Diff_of_Gauss.cpp:561-693) of the
                                   * oriented points, same order; needs the oriented outputs.  Row q of frame f is
                                   * valid for q < min(oriented_counts[f], oriented_cap) */
    size_t descriptors_bytes;
    uint8_t* descriptor_defined;  /* [n][oriented_cap] optional, with descriptors: 0 where the rotated window leaves the
                                   * padded level (zero descriptor, see vslam_sift_descriptors) */
    size_t descriptor_defined_bytes;
} vslam_batch_out;

/* Defaults of the reference's literals; the three list capacities scale with the frame area
 * (harris_cap = dog_cap = rows*cols/8, oriented_cap = rows*cols/32, each rounded up to a multiple
 * of 4096 and not below 65536 / 65536 / 16384: 262144 / 262144 / 65536 at 1920x1080, 1040384 / 1040384 /
 * 262144 at 3840x2160). */
void vslam_params_default(vslam_params* p, int rows, int cols);
/* Pure host computation (no GPU needed). */
int vslam_batch_layout_query(const vslam_params* p, vslam_batch_layout* out);
/* Bytes each output buffer needs for n_frames frames under *p: sets struct_size and every x_bytes field
 * of *sizes (also for outputs the parameters do not produce - a buffer is only needed where its pointer
 * will be non-NULL); the pointers are left as they are.  Pure host computation. */
int vslam_batch_out_required(const vslam_params* p, int n_frames, vslam_batch_out* sizes);
/* Harris + DoG over n frames already resident in HBM (frame f at d_frames +
 * f*frame_stride, dense rows).  Asynchronous on the context stream.  This is the fused
 * path of BASELINE config 4: every frame is read from HBM once per path and every
 * output written once; frames are independent, so a multi-GPU job shards frames across
 * ranks with no data-path collective.
 * Stream capture: a call may be recorded into a hipGraph (hipStreamBeginCapture on the context's stream ... EndCapture)
 * once ONE call with the same parameters, the same batch size AND the same matrix-path setting
 * (vslam_ctx_set_matrix_path) has run outside a capture - the workspace, the blur taps (each path has its own tables) and
 * the side streams are created by the first call, and a capture can allocate nothing: a captured call that finds a table
 * missing returns VSLAM_ERR_UNSUPPORTED and launches nothing more.  The side-stream forks all start from
 * and join back to the capturing stream; the stream tuner does not time captured calls
 * (tests/test_gpu_batch.py::test_batch_call_captured_into_a_graph). */
int vslam_detect_batch_dev(vslam_ctx* ctx, const vslam_params* p, const uint8_t* d_frames, size_t frame_stride,
                           int n_frames, const vslam_batch_out* out);

/* The same detection for callers with HOST memory and no device code of their own (plain C, ctypes + numpy, ...):
 * n_frames dense frames in host memory in, the two keypoint lists out, packed frame after frame; synchronous.
 * Harris list (when p->do_harris) and DoG list (when p->n_octaves > 0; p->localize chooses candidates with value >=
 * min_contrast or FeaturePointLocalization's survivors, as in vslam_detect_batch_dev).  For each list: `x` receives
 * the records of all frames back to back (x_bytes = capacity of the caller's buffer; records beyond it are not
 * written), x_offsets[f] (n_frames + 1 entries) the record index where frame f starts, x_counts[f] the frame's true
 * total (may exceed the per-frame capacity p->harris_cap / p->dog_cap, only that many are listed).  A list whose
 * three pointers are all NULL is skipped.  The images (response, pyramid ...) stay on the device and are dropped:
 * use the per-image entry points or vslam_detect_batch_dev for them.  This is a convenience path - it allocates its
 * device buffers per call and copies over PCIe synchronously; the throughput path is vslam_detect_batch_dev
 * (visualslam_amd/cxx/batch_detector.hpp pipelines it from host memory). */
typedef struct {
    size_t struct_size; /* = sizeof(vslam_host_lists) */
    vslam_kp* harris;
    size_t harris_bytes;
    uint64_t* harris_offsets;
    uint32_t* harris_counts;
    vslam_point* dog;
    size_t dog_bytes;
    uint64_t* dog_offsets;
    uint32_t* dog_counts;
} vslam_host_lists;
int vslam_detect_batch_host(vslam_ctx* ctx, const vslam_params* p, const uint8_t* frames, size_t frame_stride, int n_frames,
                            const vslam_host_lists* out);

/* Packs the first min(counts[f], cap) records of every frame's list ([n_frames][cap] records of
 * record_bytes each: harris_kps, dog_points or oriented_points of vslam_detect_batch_dev) back to back
 * into `packed`, frame after frame, and writes offsets[f] = sum over g < f of min(counts[g], cap) for
 * f = 0..n_frames (offsets[n_frames] = total records): a host-fed caller then downloads
 * offsets[n_frames] * record_bytes bytes instead of n_frames * cap records of mostly padding.  All
 * pointers are DEVICE pointers; asynchronous on the context stream, behind the call that wrote the lists.
 * Records that do not fit packed_bytes are not written (offsets still give their positions, so
 * offsets[n_frames] * record_bytes > packed_bytes tells the caller).  record_bytes: a multiple of 4. */
int vslam_pack_lists_dev(vslam_ctx* ctx, const void* lists, size_t record_bytes, uint32_t cap, const uint32_t* counts,
                         int n_frames, void* packed, size_t packed_bytes, uint64_t* offsets);

/* The same for SLAM::point lists (dog_points, oriented_points) with every record squeezed from 24 to 16 bytes on the way:
 * a host-fed caller downloads a third fewer bytes (the lists are what saturates PCIe on the batched path).  Lossless for
 * what the detector writes: `padding` is 0 or 1, `octave` < VSLAM_MAX_OCTAVES, `level` < 6 - the three small fields share
 * one word, tag = level | octave << 8 | padding << 16.  vslam_points16_expand (host, pure arithmetic) is the inverse:
 * expand(pack(list)) is byte-identical to the list (tests/test_gpu_batch.py::test_pack_points16).  packed_bytes counts
 * bytes of `packed`; records that do not fit are not written, offsets are as for vslam_pack_lists_dev.  `lists` 8-byte,
 * `packed` 16-byte aligned (VSLAM_ERR_INVALID otherwise: the records move as 8- and 16-byte words). */
typedef struct {
    int32_t row, col, value;
    uint32_t tag; /* level | octave << 8 | padding << 16 */
} vslam_point16;
int vslam_pack_points16_dev(vslam_ctx* ctx, const vslam_point* lists, uint32_t cap, const uint32_t* counts, int n_frames,
                            vslam_point16* packed, size_t packed_bytes, uint64_t* offsets);
/* Host: out[i] = the SLAM::point of in[i], i < n (no GPU involved). */
void vslam_points16_expand(const vslam_point16* in, size_t n, vslam_point* out);

/* The sending side of the one collective of the multi-GPU path: totals[0] = sum of
 * harris_counts[0..n_frames), totals[1] = the same for dog_counts (either list may be NULL -> 0), as two
 * uint64 in DEVICE memory, asynchronous on the context stream - so that the rank's {harris, dog} pair
 * can go into ncclAllGather on the same stream without a host round trip. */
int vslam_count_totals_dev(vslam_ctx* ctx, const uint32_t* harris_counts, const uint32_t* dog_counts, int n_frames,
                           uint64_t* totals);

/* Timing hook for bench.py: when enabled, the context brackets every launch of the
 * named kernel with HIP events on the stream the launch goes to (the context's stream or one of the
 * batched path's side streams); vslam_kernel_timing_read synchronises and returns launches and total
 * milliseconds since the last reset.  "name@N" restricts the hook to the launches of octave N
 * ("k_pyr_octave@0", "k_gauss_h_strip@3", "k_extrema_w3@1"); NULL or "" switches it off. */
int vslam_kernel_timing_enable(vslam_ctx* ctx, const char* kernel_name);
int vslam_kernel_timing_read(vslam_ctx* ctx, int* launches, double* total_ms);
/* Names of the kernels a batch launches, '\n'-separated (for profiles and the hook). */
const char* vslam_kernel_names(void);

#ifdef __cplusplus
}
#endif
#endif
