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
        // the per-pixel helper of :10-29 reproduces the response at one interior pixel: det(M) - k tr(M)^2
        int smMismatch = 0;
        if (img.rows > 8 && img.cols > 8) {
            const int i = img.rows / 2, j = img.cols / 2;
            Mat M = Mat::zeros(2, 2, CV_32F);
            StructureMatrix(M, grad_x, grad_y, 1, i, j);
            const float det = (float)((double)M.at<float>(0, 0) * M.at<float>(1, 1) - (double)M.at<float>(0, 1) * M.at<float>(1, 0));
            const float tr = (float)((double)M.at<float>(0, 0) + (double)M.at<float>(1, 1));
            const float trtr = tr * tr, ktr = 0.04f * trtr, resp = det - ktr;
            smMismatch = HResponse.at<float>(i, j) != (resp > 0 ? resp : 0.0f);
        }
        // the same result from the single fused kernel
        const std::vector<vslam_kp> kps = HarrisKeypoints(img, 0.04f);
        const auto t2 = std::chrono::steady_clock::now();
        std::printf("{\"exe\": \"Harris\", \"rows\": %d, \"cols\": %d, \"keypoints_stagewise\": %ld, \"keypoints_fused\": %zu, "
                    "\"nms3_maxima\": %ld, \"structure_matrix_mismatch\": %d, \"ms_stagewise\": %.3f, \"ms_fused\": %.3f}\n",
                    img.rows, img.cols, circles, kps.size(), strict, smMismatch, std::chrono::duration<double, std::milli>(t1 - t0).count(),
                    std::chrono::duration<double, std::milli>(t2 - t1).count());
        return circles == (long)kps.size() && !smMismatch ? 0 : 2;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Harris: %s\n", e.what());
        return EXIT_FAILURE;
    }
}
