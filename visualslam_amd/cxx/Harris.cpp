// Headless counterpart of the reference's `Harris` executable (Harris_corners.cpp:146-193):
// same call sequence through the same function names, display replaced by a printed summary.
//   usage: Harris [image.pgm | WxH]
#include <chrono>
#include <cstdio>

#include "imgio.hpp"
#include "vslam_cxx.hpp"

using namespace cv;

int main(int argc, char** argv) {
    try {
        Mat img = imgio::from_arg(argc, argv, 1754, 1240);
        const auto t0 = std::chrono::steady_clock::now();
        Mat blurred;
        vslamcv::GaussianBlur(img, blurred, Size(3, 3), 0, 0);           // :158
        Mat grad_x, grad_y;
        vslamcv::Sobel(blurred, grad_x, CV_32F, 1, 0, 1);                // :163
        vslamcv::Sobel(blurred, grad_y, CV_32F, 0, 1, 1);                // :164
        Mat HResponse = HarrisCorner(grad_x, grad_y);                    // :172
        Mat abs_HResponse;
        vslamcv::convertScaleAbs(HResponse, abs_HResponse);              // :176
        Mat nms = NonMaximumSuppression(abs_HResponse, 3);               // :178
        Mat nms2 = NMS2(HResponse, 5);                                   // :179
        Mat abs_NMS;
        vslamcv::convertScaleAbs(nms2, abs_NMS);                         // :181
        const auto t1 = std::chrono::steady_clock::now();
        long circles = 0, strict = 0;
        for (int r = 0; r < abs_NMS.rows; ++r)
            for (int c = 0; c < abs_NMS.cols; ++c) {
                circles += abs_NMS.at<uchar>(r, c) > 253;                // draw_them_circles criterion, :139
                strict += nms.at<uchar>(r, c) != 0;
            }
        // the same result from the single fused kernel
        const std::vector<vslam_kp> kps = HarrisKeypoints(img, 0.04f);
        const auto t2 = std::chrono::steady_clock::now();
        std::printf("{\"exe\": \"Harris\", \"rows\": %d, \"cols\": %d, \"keypoints_stagewise\": %ld, \"keypoints_fused\": %zu, "
                    "\"nms3_maxima\": %ld, \"ms_stagewise\": %.3f, \"ms_fused\": %.3f}\n",
                    img.rows, img.cols, circles, kps.size(), strict, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(t2 - t1).count());
        return circles == (long)kps.size() ? 0 : 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Harris: %s\n", e.what());
        return EXIT_FAILURE;
    }
}
