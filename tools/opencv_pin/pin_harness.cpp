// Pin harness: issues, against a REAL OpenCV, the three OpenCV-dependent operations of the reference's hot path that
// cannot be reached from Python's cv2 (VERDICT r4 "What's missing" #2), and dumps what OpenCV returns as raw binary so
// that tests/test_opencv_crosscheck.py can compare it with the CPU oracle:
//
//   roi_blur  <parent.f32> <rows> <cols> <out.f32> (<x> <y> <sigma>)...
//       GaussianBlur on a 16x16 ROI of a larger CV_32F Mat without BORDER_ISOLATED - the filter reads the PARENT
//       around the window (call shape of Diff_of_Gauss.cpp:341-348).  Python cannot do this: a numpy view handed to
//       cv2 becomes a Mat that does not know its parent.
//   det_trace <ix.f32> <iy.f32> <n> <k> <out.f32>
//       per element: M = [ix*ix, ix*iy; ix*iy, iy*iy] (CV_32F 2x2), float det = determinant(M),
//       float tr = trace(M)[0], det - k*(tr*tr): the double-precision route of cv::determinant / cv::trace and
//       the narrowing assignments (call shape of Harris_corners.cpp:54-57).  Writes det, tr, response per element.
//   mat_at    <rows> <cols> <pad> <out.u8> (<x> <y>)...
//       copyMakeBorder(img, padded, pad x4, BORDER_REPLICATE) and padded.at<uchar>(x, y) with x used as the ROW
//       (call shape of Diff_of_Gauss.cpp:549, :775): the element the release build of Mat::at returns for a column
//       index beyond the row (linear addressing x * step + y).  A debug build of OpenCV asserts instead: reported.
//
// Not reference code: a test driver written for this repository.  Never built by build(): OpenCV is absent from the
// image; tools/pin_with_opencv.sh builds it where an OpenCV exists.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>

#include "pin_io.hpp"  // the OpenCV-free half: raw files, argument parsing (checked in the CPU suite by pin_io_check.cpp)

static std::vector<float> read_f32(const char* path, size_t n) {
    std::vector<float> v;
    if (!pin::read_f32(path, n, &v)) {
        std::fprintf(stderr, "cannot read %zu floats from %s\n", n, path);
        std::exit(2);
    }
    return v;
}

static void write_all(const char* path, const void* p, size_t bytes) {
    if (!pin::write_all(path, p, bytes)) {
        std::fprintf(stderr, "cannot write %s\n", path);
        std::exit(2);
    }
}

int main(int argc, char** argv) {
    if (argc < 2) {
        std::fprintf(stderr, "usage: pin_harness roi_blur|det_trace|mat_at|version ...\n");
        return 2;
    }
    const std::string cmd = argv[1];
    if (cmd == "version") {
        std::printf("%s\n", CV_VERSION);
        return 0;
    }
    if (cmd == "roi_blur") {
        pin::RoiBlurArgs a;
        if (!pin::parse_roi_blur(argc, argv, &a)) {
            std::fprintf(stderr, "bad arguments for roi_blur\n");
            return 2;
        }
        std::vector<float> data = read_f32(a.parent.c_str(), (size_t)a.rows * a.cols);
        cv::Mat parent(a.rows, a.cols, CV_32F, data.data());
        std::vector<float> out;
        for (const auto& c : a.cases) {
            cv::Mat win(parent, cv::Rect(c.x, c.y, 16, 16));
            cv::Mat blurred;
            cv::GaussianBlur(win, blurred, cv::Size(0, 0), c.sigma, 0, cv::BORDER_DEFAULT);
            for (int r = 0; r < 16; ++r)
                for (int cc = 0; cc < 16; ++cc) out.push_back(blurred.at<float>(r, cc));
        }
        write_all(a.out.c_str(), out.data(), out.size() * 4);
        return 0;
    }
    if (cmd == "det_trace") {
        pin::DetTraceArgs a;
        if (!pin::parse_det_trace(argc, argv, &a)) {
            std::fprintf(stderr, "bad arguments for det_trace\n");
            return 2;
        }
        const size_t n = a.n;
        const float k = a.k;
        const std::vector<float> ix = read_f32(a.ix.c_str(), n), iy = read_f32(a.iy.c_str(), n);
        std::vector<float> out(3 * n);
        for (size_t i = 0; i < n; ++i) {
            cv::Mat M = cv::Mat::zeros(2, 2, CV_32F);
            M.at<float>(0, 0) += ix[i] * ix[i];
            M.at<float>(1, 1) += iy[i] * iy[i];
            M.at<float>(0, 1) += ix[i] * iy[i];
            M.at<float>(1, 0) = M.at<float>(0, 1);
            float det = cv::determinant(M);
            float tr = cv::trace(M)[0];
            float response = det - k * (tr * tr);
            out[3 * i] = det, out[3 * i + 1] = tr, out[3 * i + 2] = response;
        }
        write_all(a.out.c_str(), out.data(), out.size() * 4);
        return 0;
    }
    if (cmd == "mat_at") {
        pin::MatAtArgs a;
        if (!pin::parse_mat_at(argc, argv, &a)) {
            std::fprintf(stderr, "bad arguments for mat_at\n");
            return 2;
        }
        cv::Mat img(a.rows, a.cols, CV_8U);
        for (int r = 0; r < a.rows; ++r)
            for (int c = 0; c < a.cols; ++c) img.at<uchar>(r, c) = pin::mat_at_pixel(r, c);
        cv::Mat padded;
        cv::copyMakeBorder(img, padded, a.pad, a.pad, a.pad, a.pad, cv::BORDER_REPLICATE);
        if (!padded.isContinuous()) {
            std::fprintf(stderr, "padded Mat is not continuous: the linear-addressing rule does not apply\n");
            return 3;
        }
        std::vector<uchar> out;
        try {
            for (const auto& xy : a.at) out.push_back(padded.at<uchar>(xy.first, xy.second));
        } catch (const cv::Exception& e) {
            std::fprintf(stderr, "Mat::at asserted (debug build of OpenCV): %s\n", e.what());
            return 4;
        }
        write_all(a.out.c_str(), out.data(), out.size());
        return 0;
    }
    std::fprintf(stderr, "bad arguments for %s\n", cmd.c_str());
    return 2;
}
