// Headless, asserting counterpart of the reference's `Pyramid_Test` (tests/GaussPyramid_Test.cpp):
// the values that program prints are checked against the answers derivable from the reference
// sources.  building.jpg is 868x600; geometry needs only its size.
#include <cmath>
#include <cstdio>
#include <cstring>

#include "imgio.hpp"
#include "vslam_cxx.hpp"

using namespace cv;

static int failures = 0;
#define EXPECT(cond)                                                    \
    do {                                                                \
        if (!(cond)) {                                                  \
            std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                 \
        }                                                               \
    } while (0)

int main() {
    try {
        Mat img = imgio::synthetic(600, 868);
        GaussPyramid pyramid{img, 4, 1.6};                      // tests/GaussPyramid_Test.cpp:83-85
        EXPECT(pyramid.getNumOctaves() == 4);                   // :88
        EXPECT(pyramid.getNumLevels() == 6);                    // :89
        const int want[4][2] = {{1736, 1200}, {868, 600}, {434, 300}, {217, 150}};
        for (int o = 0; o < 4; ++o) {                           // :92-93
            EXPECT(pyramid.octaveBlur(o).size() == 6);
            EXPECT(pyramid.octaveDiff(o).size() == 5);
            EXPECT(pyramid.octaveBlur(o)[0].cols == want[o][0] && pyramid.octaveBlur(o)[0].rows == want[o][1]);
            EXPECT(pyramid.octaveImage(o).cols == want[o][0]);
        }
        EXPECT(pyramid.getSigmaAt(0, 0) == 1.6);                                  // :99
        EXPECT(std::fabs(pyramid.getSigmaAt(0, 5) - 5.079683366298239) < 1e-12);  // :102
        EXPECT(std::fabs(pyramid.getSigmaAt(0, 3) - 3.2) < 1e-12);                // :105
        EXPECT(pyramid.octaveSigma(3).size() == 6);                               // :110
        // DoG = saturating difference of adjacent Gaussians (GaussPyramid.cpp:197)
        const auto& g = pyramid.octaveBlur(2);
        const auto& d = pyramid.octaveDiff(2);
        long bad = 0;
        for (int r = 0; r < g[0].rows; ++r)
            for (int c = 0; c < g[0].cols; ++c) {
                const int diff = (int)g[3].at<uchar>(r, c) - (int)g[2].at<uchar>(r, c);
                bad += d[2].at<uchar>(r, c) != (diff > 0 ? diff : 0);
            }
        EXPECT(bad == 0);
        // gradient images exist for octave 1 level 0, CV_32F, 868x600 (tests/GaussPyramid_Test.cpp:114-117)
        for (const Mat* m : {&pyramid.octaveGradX(1)[0], &pyramid.octaveGradY(1)[0], &pyramid.octaveGradMag(1)[0], &pyramid.octaveGradOrient(1)[0]})
            EXPECT(m->type() == CV_32F && m->cols == 868 && m->rows == 600);
        {
            const Mat& gx = pyramid.octaveGradX(1)[0];
            const Mat& b = pyramid.octaveBlur(1)[0];
            EXPECT(gx.at<float>(10, 10) == (float)((int)b.at<uchar>(10, 11) - (int)b.at<uchar>(10, 9)));
            EXPECT(gx.at<float>(10, 0) == 0.0f);  // reflect-101
        }
        {   // whole-pyramid gradient maps (GaussPyramid.hpp:41-44) on a small pyramid: every octave, 6 levels each
            Mat small_img = imgio::synthetic(48, 64);
            GaussPyramid sp{small_img, 2, 1.6};
            EXPECT(sp.pyramidGradX().size() == 2 && sp.pyramidGradOrient().at(1).size() == 6);
            EXPECT(sp.pyramidGradMag().at(0)[5].rows == 96 && sp.pyramidGradY().at(1)[0].cols == 64);
        }
        {   // the reference's constructor runs processGradients itself (GaussPyramid.cpp:118): with setEagerGradients the
            // mirror does too, and what the getters then return equals what the lazy pyramid computes on demand
            Mat small_img = imgio::synthetic(48, 64, 3);
            GaussPyramid lazy{small_img, 2, 1.6};
            GaussPyramid::setEagerGradients(true);
            GaussPyramid eager{small_img, 2, 1.6};
            GaussPyramid::setEagerGradients(false);
            EXPECT(GaussPyramid::eagerGradients() == false);
            for (int o = 0; o < 2; ++o)
                for (int l = 0; l < 6; ++l) {
                    const Mat &a = eager.octaveGradOrient(o)[l], &b = lazy.octaveGradOrient(o)[l];
                    const Mat &am = eager.octaveGradMag(o)[l], &bm = lazy.octaveGradMag(o)[l];
                    EXPECT(a.rows == b.rows && a.cols == b.cols && std::memcmp(a.data, b.data, (size_t)a.rows * a.step) == 0);
                    EXPECT(std::memcmp(am.data, bm.data, (size_t)am.rows * am.step) == 0);
                }
        }
        GaussPyramid autop{img, 1.6};  // second constructor: floor(log2(600)) - 4 = 5 octaves
        EXPECT(autop.getNumOctaves() == 5);
        std::vector<Mat> padded = GaussPyramid::padOctave(1, d);
        EXPECT(padded.size() == 5 && padded[0].rows == d[0].rows + 2 && padded[0].at<uchar>(0, 0) == d[0].at<uchar>(0, 0));
        std::printf("{\"exe\": \"Pyramid_Test\", \"failures\": %d}\n", failures);
        return failures ? 1 : 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Pyramid_Test: %s\n", e.what());
        return EXIT_FAILURE;
    }
}
