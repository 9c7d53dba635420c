/*
 * vslam_oracle.h -- CPU restatement of the reference keypoint-detection hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and there only as the checker / the timed CPU baseline.  The product
 * path (visualslam_amd/) never links, imports or falls back to it.
 *
 * PARITY UNPINNED: the arithmetic of the reference lives in OpenCV (un-vendored,
 * un-pinned: KeyPointDetection/CMakeLists.txt:3, README.md:31), OpenCV is absent
 * from this image and the reference holds no golden vectors or asserting tests
 * (SURVEY.md section 8c).  The functions below restate the reference's own loops
 * literally and OpenCV 4.x's published 8-bit fixed-point algorithms from memory;
 * they are pinned only by analytic known answers (tests/test_oracle_kat.py).
 *
 * All citations are relative to /root/reference/KeyPointDetection/.
 */
#ifndef VSLAM_ORACLE_H
#define VSLAM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VO_MAX_OCTAVES 16
#define VO_NUM_LEVELS 6 /* scaleSamples_ + 3, GaussPyramid.hpp:65-66 */
#define VO_NUM_DOGS 5   /* GaussPyramid.cpp:193 */

/* SLAM::point, Diff_of_Gauss.cpp:27-35 (six ints, 24 bytes). */
typedef struct {
    int32_t row, col, value, padding, octave, level;
} vo_point;

/* Harris keypoint: NMS2 survivor whose 8-bit view is > 253 (Harris_corners.cpp:139). */
typedef struct {
    int32_t row, col;
    float response;
} vo_kp;

/* ---- OpenCV primitive semantics ---- */

/* A1: BORDER_REFLECT_101 index map, repeated until inside [0,len). */
int vo_reflect101(int p, int len);

/* A2-i: automatic kernel width for 8-bit input: cvRound(sigma*3*2+1)|1. */
int vo_gauss_ksize_u8(double sigma);

/* A2-ii/iii: Gaussian taps quantised to unsigned 8.8 fixed point with error
 * diffusion; taps[n]; sum is exactly 256.  sigma<=0 uses the fixed small tables /
 * the n-derived sigma.  Returns 0, or -1 for invalid n. */
int vo_gauss_taps_q8(int n, double sigma, uint16_t* taps);

/* A2-iv: GaussianBlur on CV_8U: exact 2-D integer sum, (acc + 32768) >> 16,
 * BORDER_REFLECT_101.  ksize 0 => from sigma.  Square kernels only (both
 * reference call sites: Harris_corners.cpp:158, GaussPyramid.cpp:177). */
int vo_gaussian_blur_u8(const uint8_t* src, int rows, int cols, size_t step, int ksize, double sigma,
                        uint8_t* dst, size_t dst_step);

/* A3: Sobel(ddepth=CV_32F, ksize=1): [-1 0 1] along x (dx=1) or y (dy=1). */
int vo_sobel_k1_u8_f32(const uint8_t* src, int rows, int cols, size_t step, int dx, int dy, float* dst,
                       size_t dst_step_bytes);

/* A4: resize x2 INTER_LINEAR (8-bit fixed point) and x0.5 INTER_NEAREST. */
int vo_resize_linear2x_u8(const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst, size_t dst_step);
void vo_half_size(int rows, int cols, int* out_rows, int* out_cols);
int vo_resize_nearest_half_u8(const uint8_t* src, int rows, int cols, size_t step, uint8_t* dst,
                              size_t dst_step);

/* A7: convertScaleAbs f32 -> u8 as x86-64 OpenCV: round_half_even(|x|) saturated to 255 below 2^31;
 * NaN and |x| >= 2^31 give 0 (cvRound returns INT_MIN there). */
int vo_convert_scale_abs_f32(const float* src, int rows, int cols, size_t step_bytes, uint8_t* dst,
                             size_t dst_step);

/* ---- Harris path ---- */

/* HarrisCorner, Harris_corners.cpp:31-68 with StructureMatrix :10-29.  Literal
 * f32 accumulation order, double determinant, three separately rounded f32 ops.
 * Intended (rows x cols) indexing, Appendix B-1.  window odd >= 1. */
int vo_harris_from_grad_f32(const float* ix, const float* iy, int rows, int cols, size_t step_bytes, float k,
                            int window, float* resp, size_t resp_step_bytes);

/* Front end of main(): GaussianBlur 3x3 -> Sobel x/y -> HarrisCorner
 * (Harris_corners.cpp:158-172). */
int vo_harris_response_u8(const uint8_t* img, int rows, int cols, size_t step, float k, int window,
                          float* resp, size_t resp_step_bytes);

/* NonMaximumSuppression, Harris_corners.cpp:70-81: mask = v > max(neighbours in
 * window, centre excluded, out-of-image ignored); 255/0. */
int vo_nms_strict_u8(const uint8_t* src, int rows, int cols, size_t step, int window, uint8_t* mask,
                     size_t mask_step);
int vo_nms_strict_f32(const float* src, int rows, int cols, size_t step_bytes, int window, uint8_t* mask,
                      size_t mask_step);

/* NMS2, Harris_corners.cpp:83-129: half-open window [i-p,i+p) x [j-p,j+p),
 * >= test, f32 map (Appendix B-3/B-4).  true_max may be NULL. */
int vo_nms2_f32(const float* resp, int rows, int cols, size_t step_bytes, int window, float* out,
                size_t out_step_bytes, float* true_max);

/* H8: row-major list of pixels whose NMS2 map value converts to an 8-bit value
 * > 253 (i.e. >= 253.5).  Returns total count; writes at most cap. */
size_t vo_harris_keypoints(const float* nms2, int rows, int cols, size_t step_bytes, vo_kp* out, size_t cap);

/* ---- DoG pyramid path ---- */

typedef struct {
    int n_octaves;
    double sigma0;
    int rows[VO_MAX_OCTAVES], cols[VO_MAX_OCTAVES];
    double sigma[VO_MAX_OCTAVES][VO_NUM_LEVELS];
    int ksize[VO_MAX_OCTAVES][VO_NUM_LEVELS];
    uint8_t* base[VO_MAX_OCTAVES];                 /* img_pyramid, GaussPyramid.cpp:119 */
    uint8_t* gauss[VO_MAX_OCTAVES][VO_NUM_LEVELS]; /* gauss_pyramid */
    uint8_t* dog[VO_MAX_OCTAVES][VO_NUM_DOGS];     /* diff_pyramid */
} vo_pyramid;

/* calculateNumOctaves, GaussPyramid.cpp:150-152. */
int vo_auto_num_octaves(int rows, int cols);
/* calculateSigma(octave, level), GaussPyramid.cpp:160-162. */
double vo_sigma(double sigma0, int octave, int level);

/* GaussPyramid ctor + createPyramid, GaussPyramid.cpp:106-131 (gradients excluded). */
vo_pyramid* vo_pyramid_build_u8(const uint8_t* img, int rows, int cols, size_t step, int n_octaves,
                                double sigma0);
void vo_pyramid_free(vo_pyramid* p);

/* processGradients, GaussPyramid.cpp:65-104, for ONE Gaussian image: Sobel x / y (ksize 1,
 * :87,:90), cv::magnitude (:93) = sqrt(x*x + y*y) in f32, cv::phase(..., true) (:96) = OpenCV's
 * fastAtan2 polynomial in degrees (restated from memory of mathfuncs_core: |error| <= 0.3 deg vs
 * atan2; parity with a real OpenCV build is unpinned).  Any output pointer may be NULL.
 * Steps are in bytes. */
float vo_fast_atan2_deg(float y, float x);
int vo_level_gradients(const uint8_t* g, int rows, int cols, size_t step, float* gx, float* gy, float* mag,
                       float* orient, size_t out_step_bytes);

/* Lattice geometry of initialKeypointDetection, Diff_of_Gauss.cpp:267-268:
 * sites i = pad, pad+ws, ... < rows. */
void vo_extrema_lattice(int rows, int cols, int window, int* lat_rows, int* lat_cols);

/* initialKeypointDetection, Diff_of_Gauss.cpp:254-297, up to (not including)
 * FeaturePointLocalization.  For each level 1..3: a byte mask (1/0) per lattice
 * site in mask[(level-1)*lat_rows*lat_cols + li*lat_cols + lj] (may be NULL), and
 * the ordered list of candidates with value >= min_contrast (level, i, j order).
 * Returns the total list length; writes at most cap points. */
size_t vo_dog_extrema(const vo_pyramid* p, int octave, int window, int min_contrast, uint8_t* mask,
                      vo_point* out, size_t cap);

/* Extension (not reference code): dense 3x3x3 test on every pixel of levels 1..3, replicate border,
 * ties count; mask[(level-1)*rows*cols + y*cols + x]; points (row+1, col+1, value, 1, octave, level)
 * with value >= min_contrast in (level, row, col) order. */
size_t vo_dog_extrema_dense(const vo_pyramid* p, int octave, int min_contrast, uint8_t* mask, vo_point* out, size_t cap);

/* Automatic kernel width for CV_32F input (cvRound(sigma*4*2+1)|1) and getGaussianKernel(n, sigma, CV_32F). */
int vo_gauss_ksize_f32(double sigma);
int vo_gauss_kernel_f32(int n, double sigma, float* k);
/* GaussianBlur(Mat(parent, Rect(x0, y0, w, h)), dst, Size(0,0), sigma, 0, BORDER_DEFAULT) on CV_32F (Diff_of_Gauss.cpp:341-348):
 * no BORDER_ISOLATED, so the filter reads the parent around the window and reflects only at the parent's edges. */
int vo_blur_f32_roi(const float* parent, int prows, int pcols, int x0, int y0, int w, int h, double sigma, float* dst);
/* computeEdgeResponse, Diff_of_Gauss.cpp:79-109 (tr^2/det of the 2x2 gradient-product sums). */
float vo_compute_edge_response(const float* gx, const float* gy, int rows, int cols, size_t step_elems, int row,
                               int col, int padding);
/* filterKeypoints + orientationHistogram, Diff_of_Gauss.cpp:301-372,112-133, for one octave's
 * keypoints (output of vo_dog_keypoints).  Returns the total, or (size_t)-1 on invalid keypoints. */
size_t vo_filter_keypoints(const vo_pyramid* p, int octave, const vo_point* kps, size_t n, vo_point* out, size_t cap);

/* FeaturePointLocalization (Diff_of_Gauss.cpp:223-251) on its three finite differences and
 * the candidate value; returns 1 (and the value written at :246) when the point is kept. */
int vo_feature_point_localization(int d_x, int d_y, int d_scale, int value, int* new_value);
/* initialKeypointDetection including the FeaturePointLocalization call (Diff_of_Gauss.cpp:254-297):
 * the keypoints the reference appends, in its loop order.  Returns the total; writes <= cap. */
size_t vo_dog_keypoints(const vo_pyramid* p, int octave, int window, vo_point* out, size_t cap);

/* ---- SIFT descriptor stage ---- */
/* Rotation::cos_sin_of_angle(theta, degrees = true), rotation.cpp:5-17. */
void vo_cos_sin_deg(float theta_deg, float* c, float* s);
/* Rotation::getRotatedWindowPoints, rotation.cpp:112-130: (window+1)^2 points, xy[2q] = x, xy[2q+1] = y. */
int vo_rotated_window_points(int cx, int cy, int window, float theta_deg, int32_t* xy);
/* SIFT() + rotateImageSection, Diff_of_Gauss.cpp:561-693,528-559, for one octave's oriented keypoints:
 * desc n x 128 floats, defined n bytes (optional).  Returns the number of keypoints whose rotated
 * window leaves the padded level (undefined in the reference), or (size_t)-1 on invalid input. */
size_t vo_sift_descriptors(const vo_pyramid* p, int octave, const vo_point* kps, size_t n, float* desc, uint8_t* defined);

/* CPU baseline driver (bench.py's cpu_baseline leg): the whole hot path of BASELINE config 4 --
 * Harris response, both NMS variants, keypoint list, GaussPyramid(img, n_octaves, 1.6), extrema
 * candidates with value >= 8 -- on n dense frames, `threads` OpenMP threads over frames (the
 * reference itself is single-threaded: threads = 1 is its figure).  Returns the keypoint
 * total (Harris + DoG) in *keypoints; 0 on success. */
int vo_baseline_frames(const uint8_t* frames, int n, int rows, int cols, int n_octaves, int threads,
                       unsigned long long* keypoints);

/* The f32 stages' one unpinned arithmetic choice (see the comment at g_fma_variant in vslam_oracle.c): 0 (default) rounds every
 * multiply and add separately - OpenCV's SSE2 baseline code, and what the GPU kernels compute; 1 fuses the multiply-adds of
 * hal::fastAtan32f's polynomial and of the separable f32 filter's row / column passes, as OpenCV's AVX2 + FMA3 dispatch does.
 * A mask (the two OpenCV modules dispatch independently): bit 0 = the arctangent polynomial, bit 1 = the filter passes.
 * Process-wide; not for use while another thread runs the f32 stages. */
void vo_set_fma_variant(int mask);
int vo_get_fma_variant(void);

#ifdef __cplusplus
}
#endif
#endif
