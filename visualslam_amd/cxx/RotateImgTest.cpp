// Headless, asserting counterpart of the reference's `RotateImgTest` (tests/rotate_image_test.cpp): that
// program times Rotation::getRotatedWindow against a warpAffine double crop and shows trackbar
// windows; the part of the Rotation library on the descriptor path (rotation.cpp:5-27,112-130) is
// checked here against known answers.  Host arithmetic only: runs without a GPU.
#include <cstdio>

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
    Mat I;  // getRotatedWindowPoints never touches its Mat argument (rotation.cpp:112-130)
    const Point2i center(128, 128);  // rotate_image_test.cpp:62 uses the image centre of blox.jpg (256 x 256)
    const int windowSize = 16;
    const std::vector<Point2i> id = SLAM::Rotation::getRotatedWindowPoints(I, center, windowSize, 0.0f);
    EXPECT(id.size() == 17u * 17u);
    for (int i = 0; i < 17; ++i)
        for (int j = 0; j < 17; ++j) EXPECT(id[i * 17 + j] == Point2i(center.x - 8 + j, center.y - 8 + i));
    const Point2f a90 = SLAM::Rotation::cos_sin_of_angle(90.0f);
    EXPECT(a90.y == 1.0f && a90.x > -1e-7f && a90.x < 1e-7f);
    // clockwise in image coordinates: (dx, dy) -> (-dy, dx); the float cosine of pi/2 is -4.37e-8, whose
    // products truncate to zero
    EXPECT(SLAM::Rotation::rotate_pt_CW(Point2i(center.x + 5, center.y), center, a90) == Point2i(center.x, center.y + 5));
    EXPECT(SLAM::Rotation::rotate_pt_CW(Point2i(center.x, center.y + 5), center, 90.0f) == Point2i(center.x - 5, center.y));
    EXPECT(SLAM::Rotation::convertToRadians(180.0f) == 3.14159274f);
    int collapsed = 0;  // integer truncation: a rotated window is not a bijection (rotation.cpp:22-23)
    const std::vector<Point2i> r30 = SLAM::Rotation::getRotatedWindowPoints(I, center, windowSize, 30.0f);
    for (size_t q = 1; q < r30.size(); ++q) collapsed += r30[q] == r30[q - 1];
    EXPECT(collapsed > 0);
    for (const Point2i& p : r30) EXPECT(p.x >= center.x - 12 && p.x <= center.x + 12 && p.y >= center.y - 12 && p.y <= center.y + 12);
    std::printf("{\"exe\": \"RotateImgTest\", \"points\": %zu, \"collapsed_neighbours_at_30_deg\": %d, \"failures\": %d}\n", r30.size(), collapsed, failures);
    return failures ? 1 : 0;
}
