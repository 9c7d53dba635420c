// CPU-suite check of the pin harness's OpenCV-free half (pin_io.hpp): compiles wherever g++ exists, needs no OpenCV.
//   pin_io_check <tmpdir>     exit 0 = the parsers and the raw-file helpers behave
#include <cmath>
#include <cstring>

#include "pin_io.hpp"

static int fails = 0;
#define CHECK(c)                                                   \
    do {                                                           \
        if (!(c)) std::fprintf(stderr, "FAILED: %s\n", #c), ++fails; \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 2) return 2;
    const std::string dir = argv[1];
    {   // raw files round-trip
        std::vector<float> v = {1.5f, -2.25f, 3.0f}, r;
        CHECK(pin::write_all((dir + "/v.f32").c_str(), v.data(), 12));
        CHECK(pin::read_f32((dir + "/v.f32").c_str(), 3, &r) && r == v);
        CHECK(!pin::read_f32((dir + "/v.f32").c_str(), 4, &r));  // short file
        CHECK(!pin::read_f32((dir + "/missing.f32").c_str(), 1, &r));
    }
    {
        const char* av[] = {"pin_harness", "roi_blur", "p.f32", "100", "120", "o.f32", "40", "50", "3.02", "0", "0", "9.6", "104", "84", "38.4"};
        pin::RoiBlurArgs a;
        CHECK(pin::parse_roi_blur(15, (char**)av, &a));
        CHECK(a.rows == 100 && a.cols == 120 && a.cases.size() == 3 && a.cases[2].x == 104 && a.cases[2].y == 84 && std::fabs(a.cases[1].sigma - 9.6) < 1e-12);
        pin::RoiBlurArgs b;
        CHECK(!pin::parse_roi_blur(14, (char**)av, &b));  // a case cut short
        const char* bad[] = {"pin_harness", "roi_blur", "p.f32", "100", "120", "o.f32", "105", "50", "3.02"};  // window leaves the parent
        pin::RoiBlurArgs c;
        CHECK(!pin::parse_roi_blur(9, (char**)bad, &c));
    }
    {
        const char* av[] = {"pin_harness", "det_trace", "ix", "iy", "4096", "0.04", "out"};
        pin::DetTraceArgs a;
        CHECK(pin::parse_det_trace(7, (char**)av, &a) && a.n == 4096 && a.k == 0.04f && a.out == "out");
        CHECK(!pin::parse_det_trace(6, (char**)av, &a));
    }
    {
        const char* av[] = {"pin_harness", "mat_at", "30", "40", "20", "o.u8", "5", "70", "0", "0"};
        pin::MatAtArgs a;
        CHECK(pin::parse_mat_at(10, (char**)av, &a) && a.at.size() == 2 && a.at[0].second == 70 && a.pad == 20);
        CHECK(!pin::parse_mat_at(9, (char**)av, &a));
        CHECK(pin::mat_at_pixel(0, 0) == 0 && pin::mat_at_pixel(2, 3) == ((2 * 131 + 3 * 7 + 6) & 255));
    }
    std::printf("pin_io_check: %s\n", fails ? "FAILED" : "ok");
    return fails ? 1 : 0;
}
