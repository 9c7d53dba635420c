// C++17 mirror of the reference's detector API for the hot path, implemented over the C ABI
// of include/vslam.h (hand-written HIP kernels; no CPU compute here).  Same names, argument
// meaning and return types as the reference's free functions / class, so its callers compile
// against this header unchanged (paths relative to /root/reference/KeyPointDetection/):
//
//   Mat HarrisCorner(Mat& Ix, Mat& Iy)                         Harris_corners.cpp:31
//   Mat NonMaximumSuppression(Mat& response, int windowSize)   Harris_corners.cpp:70
//   Mat NMS2(Mat& response, int windowSize)                    Harris_corners.cpp:83
//   class GaussPyramid                                         include/src/GaussPyramid/GaussPyramid.hpp:14
//   struct SLAM::point                                         Diff_of_Gauss.cpp:27
//   void initialKeypointDetection(std::vector<SLAM::point>&, GaussPyramid&, int, int)
//   bool FeaturePointLocalization(std::vector<Mat>&, std::vector<SLAM::point>&, int, SLAM::point&)
//   float computeEdgeResponse(const SLAM::point&, const Mat&, const Mat&)
//   void filterKeypoints(GaussPyramid&, int, std::vector<SLAM::point>&, std::vector<SLAM::point>&)
//                                                              Diff_of_Gauss.cpp:254
//   void SIFT(std::vector<SLAM::point>&, std::vector<std::vector<float>>&, GaussPyramid&, int)
//                                                              Diff_of_Gauss.cpp:561
//   class SLAM::Rotation (the members the descriptor stage uses)   include/src/Rotation/rotation.h:11
// plus the OpenCV calls on the path (vslamcv::GaussianBlur / Sobel / convertScaleAbs / resize).
// StructureMatrix (:10) is also provided as a per-pixel call; HarrisCorner itself runs the whole
// image in one kernel.  Errors of the C ABI surface as vslam::Error (the reference relies on cv::Exception).
// The per-level gradient getters (octaveGradX/Y/Mag/Orient, processGradients, SURVEY.md section 8f
// row 1) are computed on the GPU on first access instead of in the constructor.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vslam.h"
#include "cvlite.hpp"

namespace vslam {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};

// One process-wide context per device (the reference is single-threaded: SURVEY 8b).
vslam_ctx* default_context(int device = 0);
void check(int status, vslam_ctx* ctx, const char* what);

}  // namespace vslam

namespace SLAM {
struct point {  // Diff_of_Gauss.cpp:27-35
    point(int row_, int col_, int value_, int padding_, int octave_, int level_)
        : row(row_), col(col_), value(value_), padding(padding_), octave(octave_), level(level_) {}
    point() = default;
    int row = 0, col = 0, value = 0, padding = 0, octave = 0, level = 0;
};
static_assert(sizeof(point) == sizeof(vslam_point), "SLAM::point must stay six ints");
}  // namespace SLAM

namespace SLAM {
// include/src/Rotation/rotation.h:8-24: the members on the descriptor path (SIFT -> getRotatedWindowPoints
// -> rotate_pt_CW -> cos_sin_of_angle).  Host arithmetic, identical to the C ABI's
// vslam_cos_sin_deg / vslam_rotated_window_points.
class Transform {};
class Rotation : public Transform {
public:
    static float convertToRadians(float theta);
    static const cv::Point2f cos_sin_of_angle(float theta, bool degrees = true);
    static cv::Point2i rotate_pt_CW(const cv::Point2i& pt, const cv::Point2i& center, const cv::Point2f& angles);
    static cv::Point2i rotate_pt_CW(const cv::Point2i& pt, const cv::Point2i& center, float theta, bool degrees = true);
    static std::vector<cv::Point2i> getRotatedWindowPoints(cv::Mat& I, const cv::Point2i& center, int windowSize, float theta,
                                                           bool degrees = true);
};
}  // namespace SLAM

namespace vslamcv {  // the OpenCV call sites of the hot path
enum { BORDER_DEFAULT = 4, INTER_NEAREST = 0, INTER_LINEAR = 1 };
void GaussianBlur(const cv::Mat& src, cv::Mat& dst, cv::Size ksize, double sigmaX, double sigmaY = 0,
                  int borderType = BORDER_DEFAULT);
void Sobel(const cv::Mat& src, cv::Mat& dst, int ddepth, int dx, int dy, int ksize = 1, double scale = 1,
           double delta = 0, int borderType = BORDER_DEFAULT);
void convertScaleAbs(const cv::Mat& src, cv::Mat& dst);
void resize(const cv::Mat& src, cv::Mat& dst, cv::Size dsize, double fx, double fy, int interpolation);
}  // namespace vslamcv

// void StructureMatrix(Mat& M, Mat& Ix, Mat& Iy, int padding, int i, int j), Harris_corners.cpp:10-29:
// gathers the (2*padding+1)^2 window on the host, sums on the GPU (vslam_structure_matrix_windows).
// HarrisCorner does not call it per pixel - it runs the whole image in one kernel.
void StructureMatrix(cv::Mat& M, cv::Mat& Ix, cv::Mat& Iy, int padding, int i, int j);
cv::Mat HarrisCorner(cv::Mat& Ix, cv::Mat& Iy);
cv::Mat NonMaximumSuppression(cv::Mat& response, int windowSize);
cv::Mat NMS2(cv::Mat& response, int windowSize);
// Whole Harris front end on the 8-bit frame in one fused kernel (Harris_corners.cpp:158-182).
std::vector<vslam_kp> HarrisKeypoints(const cv::Mat& gray, float k = 0.04f);

class GaussPyramid {
public:
    GaussPyramid(cv::Mat& img, int numOctaves, double sigma);
    GaussPyramid(cv::Mat& img, double sigma);
    ~GaussPyramid();
    GaussPyramid(const GaussPyramid&) = delete;
    GaussPyramid& operator=(const GaussPyramid&) = delete;
    int getNumOctaves() const { return info_.n_octaves; }
    int getNumLevels() const { return info_.n_levels; }
    int getNumScaleSamples() const { return info_.n_levels - 3; }
    double getSigmaAt(int octave, int level) const;
    double calculateSigma(int octave, int level) const;
    const cv::Mat& octaveImage(int octave);
    const std::vector<double>& octaveSigma(int octave);
    const std::vector<cv::Mat>& octaveBlur(int octave);
    const std::vector<cv::Mat>& octaveDiff(int octave);
    // processGradients results (GaussPyramid.hpp:33-36), computed on the GPU on first access
    const std::vector<cv::Mat>& octaveGradX(int octave) { return grads(octave, 0); }
    const std::vector<cv::Mat>& octaveGradY(int octave) { return grads(octave, 1); }
    const std::vector<cv::Mat>& octaveGradMag(int octave) { return grads(octave, 2); }
    const std::vector<cv::Mat>& octaveGradOrient(int octave) { return grads(octave, 3); }
    const std::vector<cv::Mat>& imagePyramid();
    const std::map<int, std::vector<cv::Mat>>& pyramidGauss();
    const std::map<int, std::vector<cv::Mat>>& pyramidDiff();
    // GaussPyramid.hpp:41-44.  These materialise every level of every octave (96 bytes per pyramid
    // pixel, 1.06 GB for a 1080p frame) exactly as the reference's constructor does.
    const std::map<int, std::vector<cv::Mat>>& pyramidGradX() { return allGrads(0); }
    const std::map<int, std::vector<cv::Mat>>& pyramidGradY() { return allGrads(1); }
    const std::map<int, std::vector<cv::Mat>>& pyramidGradMag() { return allGrads(2); }
    const std::map<int, std::vector<cv::Mat>>& pyramidGradOrient() { return allGrads(3); }
    static std::vector<cv::Mat> padOctave(int padding, const std::vector<cv::Mat>& images);
    const vslam_pyramid* handle() const { return pyr_; }  // the HBM-resident pyramid
    // The reference runs processGradients INSIDE the constructor (GaussPyramid.cpp:118): every level's gradX / gradY /
    // magnitude / orientation exist when it returns.  Here they are formed on first access (same values through the same
    // getters; 96 bytes per pyramid pixel saved for callers that never ask).  A caller that wants the reference's timing -
    // all the work in the constructor, getters free - switches this on before constructing (or sets VSLAM_EAGER_GRADIENTS=1).
    static void setEagerGradients(bool on);
    static bool eagerGradients();

private:
    void build(cv::Mat& img, int numOctaves, double sigma);
    void checkOctave(int octave) const;
    vslam_pyramid* pyr_ = nullptr;
    vslam_pyramid_info info_{};
    // host copies are fetched lazily, one octave at a time (the stacks stay in HBM)
    std::vector<cv::Mat> img_pyramid_;
    std::map<int, std::vector<double>> sigmas_;
    std::map<int, std::vector<cv::Mat>> gauss_, diff_, grad_[4];
    const std::vector<cv::Mat>& grads(int octave, int kind);
    const std::map<int, std::vector<cv::Mat>>& allGrads(int kind);
};

// void initialKeypointDetection(...), Diff_of_Gauss.cpp:254-297, as the reference runs it: every
// lattice candidate goes through FeaturePointLocalization (:290) and the survivors are appended
// in the reference's loop order with the value rewritten at :246 (vslam_dog_keypoints).
void initialKeypointDetection(std::vector<SLAM::point>& keypoints, GaussPyramid& pyramid, int octave, int windowSize);
// The candidate stage alone (up to :287): candidates with DoG value >= minContrast (SURVEY 8a;
// 8 = the contrast test for the degenerate, exactly-singular cases).
void scaleSpaceCandidates(std::vector<SLAM::point>& candidates, GaussPyramid& pyramid, int octave, int windowSize,
                          int minContrast = 8);
// EXTENSION (no reference counterpart; SURVEY.md section 8a, note): the dense 3x3x3 scale-space test on
// every pixel of levels 1..3 - rule of :282-287, replicate border of :260, points in (level, row, col)
// order with padded coordinates like :289 (vslam_dog_extrema_dense).
void scaleSpaceExtremaDense(std::vector<SLAM::point>& candidates, GaussPyramid& pyramid, int octave, int minContrast = 8);
// bool FeaturePointLocalization(...), Diff_of_Gauss.cpp:223-251: reads the three finite
// differences from the padded DoG stack (:226-228) on the host, evaluates the contrast test on
// the GPU (vslam_localize_points), updates point.value and appends the point when kept.
bool FeaturePointLocalization(std::vector<cv::Mat>& dogs_padded, std::vector<SLAM::point>& keypoints, int level,
                              SLAM::point& point);
// float computeEdgeResponse(const SLAM::point&, const Mat& grad_x, const Mat& grad_y),
// Diff_of_Gauss.cpp:79-109: gathers the keypoint's window from the two CV_32F gradient images
// on the host (:93-94) and evaluates tr^2/det on the GPU (vslam_edge_response_windows).
float computeEdgeResponse(const SLAM::point& keypoint, const cv::Mat& grad_x, const cv::Mat& grad_y);
// void filterKeypoints(GaussPyramid&, int octave, vector<SLAM::point>& keypoints,
// vector<SLAM::point>& reducedKeypoints), Diff_of_Gauss.cpp:301-372 (vslam_filter_keypoints).
void filterKeypoints(GaussPyramid& pyramid, int octave, std::vector<SLAM::point>& keypoints,
                     std::vector<SLAM::point>& reducedKeypoints);
// void SIFT(vector<SLAM::point>& reducedKeypoints, vector<vector<float>>& featureDescriptors_vec,
// GaussPyramid&, int octave), Diff_of_Gauss.cpp:561-693 (vslam_sift_descriptors): appends one
// 128-float descriptor per oriented keypoint.  A keypoint whose rotated window leaves the padded level
// (the reference then reads foreign memory, :541) throws vslam::Error(VSLAM_ERR_RANGE) unless
// `defined` is given, in which case it receives one flag per keypoint and the descriptor is all zero.
void SIFT(std::vector<SLAM::point>& reducedKeypoints, std::vector<std::vector<float>>& featureDescriptors_vec,
          GaussPyramid& pyramid, int octave, std::vector<unsigned char>* defined = nullptr);
// featureDescriptors.dat exactly as Diff_of_Gauss.cpp:837-863 writes it (vslam_descriptor_file_write).
void writeFeatureDescriptors(const std::string& file_name, const std::vector<std::vector<float>>& featureDescriptors_vec);
