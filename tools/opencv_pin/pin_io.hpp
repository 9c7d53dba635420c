// The OpenCV-free half of the pin harness: raw-file I/O and the parsing of each sub-command's arguments into plain case
// lists.  pin_harness.cpp (needs a real OpenCV; built by tools/pin_with_opencv.sh) issues the OpenCV calls on them;
// pin_io_check.cpp compiles and exercises this half in the CPU suite (tests/test_pin_readiness_cpu.py), so that the one
// command that closes the pin does not rot while no OpenCV is around.  No stand-in OpenCV headers anywhere.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace pin {

inline bool read_f32(const char* path, size_t n, std::vector<float>* v) {
    v->assign(n, 0.0f);
    FILE* f = std::fopen(path, "rb");
    if (!f) return false;
    const bool ok = std::fread(v->data(), 4, n, f) == n;
    std::fclose(f);
    return ok;
}

inline bool write_all(const char* path, const void* p, size_t bytes) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    const bool ok = std::fwrite(p, 1, bytes, f) == bytes;
    std::fclose(f);
    return ok;
}

// roi_blur <parent.f32> <rows> <cols> <out.f32> (<x> <y> <sigma>)...
struct RoiBlurArgs {
    std::string parent, out;
    int rows = 0, cols = 0;
    struct Case {
        int x, y;
        double sigma;
    };
    std::vector<Case> cases;
};
inline bool parse_roi_blur(int argc, char** argv, RoiBlurArgs* a) {
    if (argc < 9 || (argc - 6) % 3 != 0) return false;
    a->parent = argv[2], a->rows = std::atoi(argv[3]), a->cols = std::atoi(argv[4]), a->out = argv[5];
    for (int i = 6; i + 2 < argc; i += 3) a->cases.push_back({std::atoi(argv[i]), std::atoi(argv[i + 1]), std::atof(argv[i + 2])});
    if (a->rows < 16 || a->cols < 16) return false;
    for (const auto& c : a->cases)  // the window must lie inside the parent: cv::Rect outside a Mat throws
        if (c.x < 0 || c.y < 0 || c.x + 16 > a->cols || c.y + 16 > a->rows || !(c.sigma > 0)) return false;
    return true;
}

// det_trace <ix.f32> <iy.f32> <n> <k> <out.f32>
struct DetTraceArgs {
    std::string ix, iy, out;
    size_t n = 0;
    float k = 0;
};
inline bool parse_det_trace(int argc, char** argv, DetTraceArgs* a) {
    if (argc != 7) return false;
    a->ix = argv[2], a->iy = argv[3], a->n = (size_t)std::atoll(argv[4]), a->k = (float)std::atof(argv[5]), a->out = argv[6];
    return a->n > 0;
}

// mat_at <rows> <cols> <pad> <out.u8> (<x> <y>)...
struct MatAtArgs {
    int rows = 0, cols = 0, pad = 0;
    std::string out;
    std::vector<std::pair<int, int>> at;
};
inline bool parse_mat_at(int argc, char** argv, MatAtArgs* a) {
    if (argc < 8 || (argc - 6) % 2 != 0) return false;
    a->rows = std::atoi(argv[2]), a->cols = std::atoi(argv[3]), a->pad = std::atoi(argv[4]), a->out = argv[5];
    for (int i = 6; i + 1 < argc; i += 2) a->at.emplace_back(std::atoi(argv[i]), std::atoi(argv[i + 1]));
    return a->rows > 0 && a->cols > 0 && a->pad >= 0;
}
// the test image of mat_at: a pattern whose every byte tells its position (mod 256)
inline unsigned char mat_at_pixel(int r, int c) { return (unsigned char)((r * 131 + c * 7 + (r * c) % 13) & 255); }

}  // namespace pin
