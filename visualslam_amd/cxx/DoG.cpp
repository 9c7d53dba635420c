// Headless counterpart of the reference's `DoG` executable up to the end of the hot path
// (Diff_of_Gauss.cpp:727-785): build the pyramid, run initialKeypointDetection per octave.
//   usage: DoG [image.pgm | WxH]
#include <chrono>
#include <cstdio>

#include "imgio.hpp"
#include "vslam_cxx.hpp"

using namespace cv;

int main(int argc, char** argv) {
    try {
        Mat img = imgio::from_arg(argc, argv, 512, 384);  // home.jpg is 512x384
        const int numOctaves = 4;       // :742
        const double pyr_sigma = 1.6;   // :743
        const auto t0 = std::chrono::steady_clock::now();
        GaussPyramid pyramid{img, numOctaves, pyr_sigma};  // :746
        const int windowSize = 3;       // :772
        std::vector<SLAM::point> all;
        std::printf("{\"exe\": \"DoG\", \"rows\": %d, \"cols\": %d, \"octaves\": [", img.rows, img.cols);
        for (int octave = 0; octave < pyramid.getNumOctaves(); ++octave) {  // :780
            std::vector<SLAM::point> keypoints;
            initialKeypointDetection(keypoints, pyramid, octave, windowSize);  // :785
            std::printf("%s{\"octave\": %d, \"candidates\": %zu}", octave ? ", " : "", octave, keypoints.size());
            all.insert(all.end(), keypoints.begin(), keypoints.end());
        }
        const auto t1 = std::chrono::steady_clock::now();
        std::printf("], \"keypoints\": %zu, \"ms\": %.3f}\n", all.size(), std::chrono::duration<double, std::milli>(t1 - t0).count());
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "DoG: %s\n", e.what());
        return EXIT_FAILURE;
    }
}
