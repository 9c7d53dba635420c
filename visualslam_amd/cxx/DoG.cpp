// Headless counterpart of the reference's `DoG` executable (Diff_of_Gauss.cpp:727-863): build the
// pyramid, run initialKeypointDetection (with its FeaturePointLocalization filter), filterKeypoints and
// SIFT per octave, write featureDescriptors.dat.  Drawing / imshow (:796-832, :868-873) have no
// counterpart.
//   usage: DoG [image.pgm | WxH] [descriptor file, default featureDescriptors.dat]
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
        size_t perPointMismatch = 0, oriented = 0, undefined = 0;
        std::vector<std::vector<float>> featureDescriptors_vec;  // :779
        std::printf("{\"exe\": \"DoG\", \"rows\": %d, \"cols\": %d, \"octaves\": [", img.rows, img.cols);
        for (int octave = 0; octave < pyramid.getNumOctaves(); ++octave) {  // :780
            std::vector<SLAM::point> keypoints, candidates, reducedKeypoints;
            initialKeypointDetection(keypoints, pyramid, octave, windowSize);  // :785
            filterKeypoints(pyramid, octave, keypoints, reducedKeypoints);     // :787
            std::vector<unsigned char> defined;
            SIFT(reducedKeypoints, featureDescriptors_vec, pyramid, octave, &defined);  // :791
            for (unsigned char d : defined) undefined += d == 0;
            scaleSpaceCandidates(candidates, pyramid, octave, windowSize);
            std::vector<SLAM::point> dense;  // extension: the dense 3x3x3 test (no reference counterpart)
            scaleSpaceExtremaDense(dense, pyramid, octave);
            std::printf("%s{\"octave\": %d, \"candidates\": %zu, \"keypoints\": %zu, \"oriented\": %zu, \"dense_3x3x3\": %zu}", octave ? ", " : "",
                        octave, candidates.size(), keypoints.size(), reducedKeypoints.size(), dense.size());
            if (!keypoints.empty()) {  // the per-point entry point agrees with the fused edge test on the first keypoint
                const SLAM::point& k0 = keypoints.front();
                const float r0 = computeEdgeResponse(k0, pyramid.octaveGradX(octave).at(k0.level), pyramid.octaveGradY(octave).at(k0.level));
                const bool kept = !reducedKeypoints.empty() && reducedKeypoints.front().row == k0.row && reducedKeypoints.front().col == k0.col &&
                                  reducedKeypoints.front().level == k0.level;
                // kept implies r0 < 12.1; a rejected first keypoint may also fail for an empty histogram
                if (kept && !(r0 < 12.1f)) ++perPointMismatch;
            }
            oriented += reducedKeypoints.size();
            if (octave == pyramid.getNumOctaves() - 1) {
                // the per-point entry point on the coarsest octave must reproduce the fused result
                std::vector<SLAM::point> all_candidates, one_by_one;
                scaleSpaceCandidates(all_candidates, pyramid, octave, windowSize, 0);
                std::vector<Mat> dogs_padded = GaussPyramid::padOctave((windowSize - 1) / 2, pyramid.octaveDiff(octave));  // :260
                for (auto& pt : all_candidates) FeaturePointLocalization(dogs_padded, one_by_one, pt.level, pt);          // :290
                perPointMismatch = one_by_one.size() != keypoints.size();
                for (size_t i = 0; i < one_by_one.size() && !perPointMismatch; ++i)
                    perPointMismatch += one_by_one[i].row != keypoints[i].row || one_by_one[i].col != keypoints[i].col ||
                                        one_by_one[i].value != keypoints[i].value || one_by_one[i].level != keypoints[i].level;
            }
            all.insert(all.end(), keypoints.begin(), keypoints.end());
        }
        const auto t1 = std::chrono::steady_clock::now();
        const std::string file_name = argc > 2 ? argv[2] : "featureDescriptors.dat";  // :838
        writeFeatureDescriptors(file_name, featureDescriptors_vec);                    // :839-863
        std::printf("], \"keypoints\": %zu, \"oriented\": %zu, \"descriptors\": %zu, \"undefined_windows\": %zu, \"descriptor_file\": \"%s\", \"per_point_mismatch\": %zu, \"ms\": %.3f}\n",
                    all.size(), oriented, featureDescriptors_vec.size(), undefined, file_name.c_str(), perPointMismatch,
                    std::chrono::duration<double, std::milli>(t1 - t0).count());
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "DoG: %s\n", e.what());
        return EXIT_FAILURE;
    }
}
