#include "vslam_cxx.hpp"

#include <cmath>
#include <mutex>

using cv::Mat;

namespace vslam {

vslam_ctx* default_context(int device) {
    static std::mutex mu;
    static std::map<int, vslam_ctx*> ctxs;
    std::lock_guard<std::mutex> lock(mu);
    auto it = ctxs.find(device);
    if (it != ctxs.end()) return it->second;
    vslam_ctx* c = nullptr;
    const int rc = vslam_ctx_create(device, nullptr, &c);
    if (rc != VSLAM_OK) throw Error(rc, std::string("vslam_ctx_create: ") + vslam_status_string(rc) + " (no usable HIP device: there is no CPU fallback)");
    ctxs[device] = c;
    return c;
}

void check(int status, vslam_ctx* ctx, const char* what) {
    if (status != VSLAM_OK)
        throw Error(status, std::string(what) + ": " + vslam_status_string(status) + ": " + vslam_last_error(ctx));
}

}  // namespace vslam

using vslam::check;
using vslam::default_context;

static void require(bool ok, const char* msg) {
    if (!ok) throw vslam::Error(VSLAM_ERR_INVALID, msg);
}

namespace vslamcv {

void GaussianBlur(const Mat& src, Mat& dst, cv::Size ksize, double sigmaX, double, int) {
    require(src.type() == cv::CV_8U && ksize.width == ksize.height, "GaussianBlur: CV_8U, square kernels only");
    Mat out(src.rows, src.cols, cv::CV_8U);
    vslam_ctx* c = default_context();
    check(vslam_gaussian_blur_u8(c, src.data, src.rows, src.cols, src.step, ksize.width, sigmaX, out.data, out.step), c, "GaussianBlur");
    dst = out;
}

void Sobel(const Mat& src, Mat& dst, int ddepth, int dx, int dy, int ksize, double scale, double delta, int) {
    require(src.type() == cv::CV_8U && ddepth == cv::CV_32F && ksize == 1 && scale == 1 && delta == 0, "Sobel: CV_8U -> CV_32F, ksize 1 only");
    Mat out(src.rows, src.cols, cv::CV_32F);
    vslam_ctx* c = default_context();
    check(vslam_sobel_k1_u8_f32(c, src.data, src.rows, src.cols, src.step, dx, dy, out.ptr<float>(), out.step), c, "Sobel");
    dst = out;
}

void convertScaleAbs(const Mat& src, Mat& dst) {
    require(src.type() == cv::CV_32F, "convertScaleAbs: CV_32F input only");
    Mat out(src.rows, src.cols, cv::CV_8U);
    vslam_ctx* c = default_context();
    check(vslam_convert_scale_abs_f32(c, src.ptr<float>(), src.rows, src.cols, src.step, out.data, out.step), c, "convertScaleAbs");
    dst = out;
}

void resize(const Mat& src, Mat& dst, cv::Size, double fx, double fy, int interpolation) {
    require(src.type() == cv::CV_8U, "resize: CV_8U only");
    vslam_ctx* c = default_context();
    if (fx == 2 && fy == 2 && interpolation == INTER_LINEAR) {
        Mat out(src.rows * 2, src.cols * 2, cv::CV_8U);
        check(vslam_resize_linear2x_u8(c, src.data, src.rows, src.cols, src.step, out.data, out.step), c, "resize x2");
        dst = out;
    } else if (fx == 0.5 && fy == 0.5 && interpolation == INTER_NEAREST) {
        int r, cc;
        vslam_half_size(src.rows, src.cols, &r, &cc);
        Mat out(r, cc, cv::CV_8U);
        check(vslam_resize_nearest_half_u8(c, src.data, src.rows, src.cols, src.step, out.data, out.step), c, "resize x0.5");
        dst = out;
    } else {
        throw vslam::Error(VSLAM_ERR_UNSUPPORTED, "resize: only (2,2,INTER_LINEAR) and (0.5,0.5,INTER_NEAREST) are on the hot path");
    }
}

}  // namespace vslamcv

Mat HarrisCorner(Mat& Ix, Mat& Iy) {
    require(Ix.type() == cv::CV_32F && Iy.type() == cv::CV_32F && Ix.rows == Iy.rows && Ix.cols == Iy.cols && Ix.step == Iy.step,
            "HarrisCorner: two CV_32F gradients of equal geometry");
    // intended geometry rows x cols (the reference allocates it transposed: SURVEY Appendix B-1)
    Mat image(Ix.rows, Ix.cols, cv::CV_32F);
    vslam_ctx* c = default_context();
    check(vslam_harris_from_grad_f32(c, Ix.ptr<float>(), Iy.ptr<float>(), Ix.rows, Ix.cols, Ix.step, 0.04f, 3, image.ptr<float>(), image.step), c,
          "HarrisCorner");
    return image;
}

Mat NonMaximumSuppression(Mat& response, int windowSize) {
    Mat localMax(response.rows, response.cols, cv::CV_8U);
    vslam_ctx* c = default_context();
    if (response.type() == cv::CV_8U)
        check(vslam_nms_strict_u8(c, response.data, response.rows, response.cols, response.step, windowSize, localMax.data, localMax.step), c,
              "NonMaximumSuppression");
    else
        check(vslam_nms_strict_f32(c, response.ptr<float>(), response.rows, response.cols, response.step, windowSize, localMax.data, localMax.step),
              c, "NonMaximumSuppression");
    return localMax;
}

Mat NMS2(Mat& response, int windowSize) {
    require(response.type() == cv::CV_32F, "NMS2: CV_32F response");
    Mat nms(response.rows, response.cols, cv::CV_32F);  // intended f32 map (Appendix B-3)
    vslam_ctx* c = default_context();
    float true_max = 0;
    check(vslam_nms2_f32(c, response.ptr<float>(), response.rows, response.cols, response.step, windowSize, nms.ptr<float>(), nms.step, &true_max), c,
          "NMS2");
    return nms;
}

std::vector<vslam_kp> HarrisKeypoints(const Mat& gray, float k) {
    require(gray.type() == cv::CV_8U, "HarrisKeypoints: CV_8U frame");
    vslam_ctx* c = default_context();
    std::vector<vslam_kp> out(1 << 16);
    size_t n = 0;
    check(vslam_harris_keypoints_u8(c, gray.data, gray.rows, gray.cols, gray.step, k, out.data(), out.size(), &n), c, "HarrisKeypoints");
    if (n > out.size()) {
        out.resize(n);
        check(vslam_harris_keypoints_u8(c, gray.data, gray.rows, gray.cols, gray.step, k, out.data(), out.size(), &n), c, "HarrisKeypoints");
    }
    out.resize(n);
    return out;
}

// ------------------------------------------------------------------------------- GaussPyramid

GaussPyramid::GaussPyramid(Mat& img, int numOctaves, double sigma) { build(img, numOctaves, sigma); }
GaussPyramid::GaussPyramid(Mat& img, double sigma) { build(img, 0, sigma); }  // automatic octave count

GaussPyramid::~GaussPyramid() {
    if (pyr_) vslam_pyramid_destroy(pyr_);
}

void GaussPyramid::build(Mat& img, int numOctaves, double sigma) {
    require(img.type() == cv::CV_8U && !img.empty(), "GaussPyramid: CV_8U image");
    vslam_ctx* c = default_context();
    check(vslam_pyramid_build_u8(c, img.data, img.rows, img.cols, img.step, numOctaves, sigma, &pyr_), c, "GaussPyramid");
    check(vslam_pyramid_get_info(pyr_, &info_), c, "GaussPyramid info");
    for (int o = 0; o < info_.n_octaves; ++o)
        sigmas_[o] = std::vector<double>(info_.sigma[o], info_.sigma[o] + info_.n_levels);
    if (eagerGradients())  // processGradients for every octave now, as GaussPyramid.cpp:118 does
        for (int kind = 0; kind < 4; ++kind) (void)allGrads(kind);
}

namespace {
int g_eager_gradients = -1;  // -1: not set, follow the environment
}
void GaussPyramid::setEagerGradients(bool on) { g_eager_gradients = on ? 1 : 0; }
bool GaussPyramid::eagerGradients() {
    if (g_eager_gradients >= 0) return g_eager_gradients != 0;
    const char* e = std::getenv("VSLAM_EAGER_GRADIENTS");
    return e && e[0] == '1';
}

void GaussPyramid::checkOctave(int octave) const {
    if (octave < 0 || octave >= info_.n_octaves) throw std::out_of_range("GaussPyramid: octave");  // std::map::at in the reference
}

double GaussPyramid::getSigmaAt(int octave, int level) const {
    checkOctave(octave);
    if (level < 0 || level >= info_.n_levels) throw std::out_of_range("GaussPyramid: level");
    return info_.sigma[octave][level];
}

double GaussPyramid::calculateSigma(int octave, int level) const { return vslam_sigma_at(info_.sigma0, octave, level); }

const std::vector<double>& GaussPyramid::octaveSigma(int octave) {
    checkOctave(octave);
    return sigmas_.at(octave);
}

const std::vector<Mat>& GaussPyramid::octaveBlur(int octave) {
    checkOctave(octave);
    auto it = gauss_.find(octave);
    if (it == gauss_.end()) {
        std::vector<Mat> v;
        for (int l = 0; l < info_.n_levels; ++l) {
            Mat m(info_.rows[octave], info_.cols[octave], cv::CV_8U);
            check(vslam_pyramid_get_gauss(pyr_, octave, l, m.data, m.step), default_context(), "octaveBlur");
            v.push_back(m);
        }
        it = gauss_.emplace(octave, std::move(v)).first;
    }
    return it->second;
}

const std::vector<Mat>& GaussPyramid::octaveDiff(int octave) {
    checkOctave(octave);
    auto it = diff_.find(octave);
    if (it == diff_.end()) {
        std::vector<Mat> v;
        for (int l = 0; l < info_.n_dogs; ++l) {
            Mat m(info_.rows[octave], info_.cols[octave], cv::CV_8U);
            check(vslam_pyramid_get_dog(pyr_, octave, l, m.data, m.step), default_context(), "octaveDiff");
            v.push_back(m);
        }
        it = diff_.emplace(octave, std::move(v)).first;
    }
    return it->second;
}

const std::vector<Mat>& GaussPyramid::grads(int octave, int kind) {
    checkOctave(octave);
    auto it = grad_[kind].find(octave);
    if (it == grad_[kind].end()) {  // one kernel per level fills all four kinds of the octave
        std::vector<Mat> v[4];
        for (int l = 0; l < info_.n_levels; ++l) {
            Mat m[4];
            for (auto& q : m) q.create(info_.rows[octave], info_.cols[octave], cv::CV_32F);
            check(vslam_pyramid_get_gradients(pyr_, octave, l, m[0].ptr<float>(), m[1].ptr<float>(), m[2].ptr<float>(), m[3].ptr<float>(), m[0].step),
                  default_context(), "processGradients");
            for (int q = 0; q < 4; ++q) v[q].push_back(m[q]);
        }
        for (int q = 0; q < 4; ++q) grad_[q].emplace(octave, std::move(v[q]));
        it = grad_[kind].find(octave);
    }
    return it->second;
}

const std::vector<Mat>& GaussPyramid::imagePyramid() {
    if (img_pyramid_.empty())
        for (int o = 0; o < info_.n_octaves; ++o) {
            Mat m(info_.rows[o], info_.cols[o], cv::CV_8U);
            check(vslam_pyramid_get_base(pyr_, o, m.data, m.step), default_context(), "imagePyramid");
            img_pyramid_.push_back(m);
        }
    return img_pyramid_;
}

const Mat& GaussPyramid::octaveImage(int octave) {
    checkOctave(octave);
    return imagePyramid().at((size_t)octave);
}

const std::map<int, std::vector<Mat>>& GaussPyramid::pyramidGauss() {
    for (int o = 0; o < info_.n_octaves; ++o) octaveBlur(o);
    return gauss_;
}

const std::map<int, std::vector<Mat>>& GaussPyramid::allGrads(int kind) {
    for (int o = 0; o < info_.n_octaves; ++o) grads(o, kind);
    return grad_[kind];
}

const std::map<int, std::vector<Mat>>& GaussPyramid::pyramidDiff() {
    for (int o = 0; o < info_.n_octaves; ++o) octaveDiff(o);
    return diff_;
}

// copyMakeBorder(..., BORDER_REPLICATE) + clone (GaussPyramid.cpp:133-141).  Pure data movement
// for callers that want the padded copies; the extrema kernel itself clamps addresses instead.
std::vector<Mat> GaussPyramid::padOctave(int padding, const std::vector<Mat>& images) {
    std::vector<Mat> padded;
    for (const Mat& im : images) {
        Mat p(im.rows + 2 * padding, im.cols + 2 * padding, im.type());
        const size_t es = im.elemSize();
        for (int r = 0; r < p.rows; ++r) {
            const int sr = std::min(std::max(r - padding, 0), im.rows - 1);
            for (int c = 0; c < p.cols; ++c) {
                const int sc = std::min(std::max(c - padding, 0), im.cols - 1);
                std::memcpy(p.data + p.step * r + es * c, im.data + im.step * sr + es * sc, es);
            }
        }
        padded.push_back(p);
    }
    return padded;
}

template <class F>
static void append_points(std::vector<SLAM::point>& dst, F&& call) {
    vslam_ctx* c = default_context();
    size_t n = 0;
    std::vector<vslam_point> buf(1 << 16);
    check(call(c, buf.data(), buf.size(), &n), c, "initialKeypointDetection");
    if (n > buf.size()) {
        buf.resize(n);
        check(call(c, buf.data(), buf.size(), &n), c, "initialKeypointDetection");
    }
    for (size_t i = 0; i < n; ++i)
        dst.emplace_back(buf[i].row, buf[i].col, buf[i].value, buf[i].padding, buf[i].octave, buf[i].level);
}

void initialKeypointDetection(std::vector<SLAM::point>& keypoints, GaussPyramid& pyramid, int octave, int windowSize) {
    append_points(keypoints, [&](vslam_ctx* c, vslam_point* out, size_t cap, size_t* n) {
        return vslam_dog_keypoints(c, pyramid.handle(), octave, windowSize, out, cap, n);
    });
}

void scaleSpaceCandidates(std::vector<SLAM::point>& candidates, GaussPyramid& pyramid, int octave, int windowSize, int minContrast) {
    append_points(candidates, [&](vslam_ctx* c, vslam_point* out, size_t cap, size_t* n) {
        return vslam_dog_extrema(c, pyramid.handle(), octave, windowSize, minContrast, nullptr, out, cap, n);
    });
}

void scaleSpaceExtremaDense(std::vector<SLAM::point>& candidates, GaussPyramid& pyramid, int octave, int minContrast) {
    append_points(candidates, [&](vslam_ctx* c, vslam_point* out, size_t cap, size_t* n) {
        return vslam_dog_extrema_dense(c, pyramid.handle(), octave, minContrast, nullptr, out, cap, n);
    });
}

bool FeaturePointLocalization(std::vector<cv::Mat>& dogs_padded, std::vector<SLAM::point>& keypoints, int level, SLAM::point& point) {
    vslam_ctx* c = default_context();
    if (level < 1 || level + 1 >= (int)dogs_padded.size()) throw vslam::Error(VSLAM_ERR_RANGE, "FeaturePointLocalization: level out of range");
    const cv::Mat& d = dogs_padded[level];
    const int i = point.row, j = point.col;
    if (i < 1 || j < 1 || i + 1 >= d.rows || j + 1 >= d.cols) throw vslam::Error(VSLAM_ERR_RANGE, "FeaturePointLocalization: point outside the padded image");
    const int q[4] = {(int)d.at<cv::uchar>(i, j - 1) - (int)d.at<cv::uchar>(i, j + 1),                                    // :226
                      (int)d.at<cv::uchar>(i - 1, j) - (int)d.at<cv::uchar>(i + 1, j),                                    // :227
                      (int)dogs_padded[level - 1].at<cv::uchar>(i, j) - (int)dogs_padded[level + 1].at<cv::uchar>(i, j),  // :228
                      point.value};
    int keep = 0, value = point.value;
    check(vslam_localize_points(c, q, 1, &keep, &value), c, "FeaturePointLocalization");
    if (!keep) return false;
    point.value = value;  // :246
    keypoints.emplace_back(point);
    return true;
}

float computeEdgeResponse(const SLAM::point& keypoint, const cv::Mat& grad_x, const cv::Mat& grad_y) {
    vslam_ctx* c = default_context();
    const int x = keypoint.col, y = keypoint.row, p = keypoint.padding;
    if (p < 0 || y - p < 0 || x - p < 0 || y + p > grad_x.rows || x + p > grad_x.cols || grad_y.rows != grad_x.rows || grad_y.cols != grad_x.cols)
        throw vslam::Error(VSLAM_ERR_RANGE, "computeEdgeResponse: window outside the gradient images");
    std::vector<float> wx, wy;
    for (int u = y - p; u < y + p; ++u)      // :93
        for (int v = x - p; v < x + p; ++v) {  // :94
            wx.push_back(grad_x.at<float>(u, v));
            wy.push_back(grad_y.at<float>(u, v));
        }
    float response = 0.f;
    check(vslam_edge_response_windows(c, wx.data(), wy.data(), (int)wx.size(), 1, &response), c, "computeEdgeResponse");
    return response;
}

void filterKeypoints(GaussPyramid& pyramid, int octave, std::vector<SLAM::point>& keypoints, std::vector<SLAM::point>& reducedKeypoints) {
    vslam_ctx* c = default_context();
    static_assert(sizeof(SLAM::point) == sizeof(vslam_point), "SLAM::point must stay binary compatible with vslam_point");
    const vslam_point* in = reinterpret_cast<const vslam_point*>(keypoints.data());
    size_t n = 0;
    std::vector<vslam_point> buf(std::max<size_t>(64, 2 * keypoints.size()));
    check(vslam_filter_keypoints(c, pyramid.handle(), octave, in, keypoints.size(), buf.data(), buf.size(), &n), c, "filterKeypoints");
    if (n > buf.size()) {
        buf.resize(n);
        check(vslam_filter_keypoints(c, pyramid.handle(), octave, in, keypoints.size(), buf.data(), buf.size(), &n), c, "filterKeypoints");
    }
    for (size_t i = 0; i < n; ++i)
        reducedKeypoints.emplace_back(buf[i].row, buf[i].col, buf[i].value, buf[i].padding, buf[i].octave, buf[i].level);
}

void StructureMatrix(cv::Mat& M, cv::Mat& Ix, cv::Mat& Iy, int padding, int i, int j) {
    vslam_ctx* c = default_context();
    if (padding < 0 || i - padding < 0 || j - padding < 0 || i + padding >= Ix.rows || j + padding >= Ix.cols || Iy.rows != Ix.rows ||
        Iy.cols != Ix.cols || M.rows < 2 || M.cols < 2)
        throw vslam::Error(VSLAM_ERR_RANGE, "StructureMatrix: window outside the gradient images");
    std::vector<float> wx, wy;
    for (int u = i - padding; u <= i + padding; ++u)      // :16
        for (int v = j - padding; v <= j + padding; ++v) {  // :17
            wx.push_back(Ix.at<float>(u, v));
            wy.push_back(Iy.at<float>(u, v));
        }
    float m[3] = {0.f, 0.f, 0.f};
    check(vslam_structure_matrix_windows(c, wx.data(), wy.data(), (int)wx.size(), 1, m), c, "StructureMatrix");
    M.at<float>(0, 0) = m[0];  // :25-28
    M.at<float>(0, 1) = m[1];
    M.at<float>(1, 0) = m[1];
    M.at<float>(1, 1) = m[2];
}

// ---- Rotation (include/src/Rotation/rotation.cpp) ------------------------------------------------

namespace SLAM {
float Rotation::convertToRadians(float theta) {  // rotation.cpp:5-7
    return (float)((double)theta * (3.1415926535897932384626433832795 / (double)180.0f));
}
const cv::Point2f Rotation::cos_sin_of_angle(float theta, bool degrees) {  // :9-17
    if (degrees) {
        float c, s;
        vslam_cos_sin_deg(theta, &c, &s);
        return cv::Point2f(c, s);
    }
    return cv::Point2f((float)std::cos((double)theta), (float)std::sin((double)theta));
}
cv::Point2i Rotation::rotate_pt_CW(const cv::Point2i& pt, const cv::Point2i& center, const cv::Point2f& angles) {  // :19-27
    cv::Point2i rot(pt.x, pt.y);
    rot -= center;
    const float a = (float)rot.x * angles.x, b = (float)rot.y * angles.y, d = (float)rot.x * angles.y, e = (float)rot.y * angles.x;
    const int x_r = (int)(a - b), y_r = (int)(d + e);
    return cv::Point2i(x_r + center.x, y_r + center.y);
}
cv::Point2i Rotation::rotate_pt_CW(const cv::Point2i& pt, const cv::Point2i& center, float theta, bool degrees) {  // :53-63
    return rotate_pt_CW(pt, center, cos_sin_of_angle(theta, degrees));
}
std::vector<cv::Point2i> Rotation::getRotatedWindowPoints(cv::Mat&, const cv::Point2i& center, int windowSize, float theta, bool degrees) {  // :112-130
    std::vector<cv::Point2i> out;
    if (windowSize <= 0) return out;
    if (!degrees) {
        const int padding = windowSize / 2;
        const cv::Point2f angles = cos_sin_of_angle(theta, false);
        for (int i = center.y - padding; i <= center.y + padding; ++i)
            for (int j = center.x - padding; j <= center.x + padding; ++j) out.push_back(rotate_pt_CW(cv::Point2i(j, i), center, angles));
        return out;
    }
    std::vector<int32_t> xy(2 * (size_t)(windowSize + 1) * (windowSize + 1));
    if (vslam_rotated_window_points(center.x, center.y, windowSize, theta, xy.data()) != VSLAM_OK)
        throw vslam::Error(VSLAM_ERR_INVALID, "getRotatedWindowPoints: bad arguments");
    for (size_t q = 0; q < xy.size() / 2; ++q) out.emplace_back(xy[2 * q], xy[2 * q + 1]);
    return out;
}
}  // namespace SLAM

// ---- SIFT (Diff_of_Gauss.cpp:561-693) ---------------------------------------------------------------

void SIFT(std::vector<SLAM::point>& reducedKeypoints, std::vector<std::vector<float>>& featureDescriptors_vec, GaussPyramid& pyramid,
          int octave, std::vector<unsigned char>* defined) {
    vslam_ctx* c = vslam::default_context();
    const size_t n = reducedKeypoints.size();
    if (defined) defined->assign(n, 1);
    if (n == 0) return;
    std::vector<float> desc(128 * n);
    std::vector<unsigned char> def(n, 1);
    vslam::check(vslam_sift_descriptors(c, pyramid.handle(), octave, reinterpret_cast<const vslam_point*>(reducedKeypoints.data()), n,
                                        desc.data(), defined ? def.data() : nullptr),
                 c, "SIFT");
    for (size_t i = 0; i < n; ++i) featureDescriptors_vec.emplace_back(desc.begin() + 128 * i, desc.begin() + 128 * (i + 1));  // :678
    if (defined) *defined = def;
}

void writeFeatureDescriptors(const std::string& file_name, const std::vector<std::vector<float>>& v) {
    std::vector<float> flat;
    flat.reserve(128 * v.size());
    for (const auto& d : v) {
        if (d.size() != 128) throw vslam::Error(VSLAM_ERR_INVALID, "writeFeatureDescriptors: a descriptor is not 128 floats");
        flat.insert(flat.end(), d.begin(), d.end());
    }
    if (vslam_descriptor_file_write(file_name.c_str(), flat.data(), v.size()) != VSLAM_OK)
        throw vslam::Error(VSLAM_ERR_INVALID, "writeFeatureDescriptors: cannot write " + file_name);
}
