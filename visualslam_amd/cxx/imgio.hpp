// Headless image input for the executables: binary PGM (P5) reader and the seeded synthetic
// frame generator of SURVEY.md section 8d (no libjpeg/libpng headers in the build image).
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#include "cvlite.hpp"

namespace imgio {

inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// 32-px checkerboard (56/200) + uniform noise in -16..15; identical to visualslam_amd/synth.py
inline cv::Mat synthetic(int rows, int cols, int frame = 0, int stream_id = 0) {
    cv::Mat m(rows, cols, cv::CV_8U);
    const uint64_t seed = 0x5EED0000ull + (uint64_t)stream_id;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const int base = (((r / 32) + (c / 32)) & 1) ? 200 : 56;
            const int noise = (int)(splitmix64(seed ^ ((uint64_t)frame << 40) ^ (uint64_t)((uint64_t)r * cols + c)) & 31) - 16;
            const int v = base + noise;
            m.at<cv::uchar>(r, c) = (cv::uchar)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    return m;
}

inline cv::Mat read_pgm(const std::string& path) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    char magic[3] = {0};
    int w = 0, h = 0, maxv = 0;
    if (std::fscanf(f, "%2s %d %d %d", magic, &w, &h, &maxv) != 4 || std::string(magic) != "P5" || maxv != 255 || w <= 0 || h <= 0) {
        std::fclose(f);
        throw std::runtime_error(path + ": not an 8-bit binary PGM");
    }
    std::fgetc(f);  // single whitespace after the header
    cv::Mat m(h, w, cv::CV_8U);
    const size_t got = std::fread(m.data, 1, (size_t)w * h, f);
    std::fclose(f);
    if (got != (size_t)w * h) throw std::runtime_error(path + ": truncated");
    return m;
}

// "<file.pgm>" or "WxH" (synthetic); default 1754x1240, the size of the reference's chessboard.png
inline cv::Mat from_arg(int argc, char** argv, int def_cols, int def_rows) {
    if (argc < 2) return synthetic(def_rows, def_cols);
    const std::string a = argv[1];
    int w = 0, h = 0;
    if (std::sscanf(a.c_str(), "%dx%d", &w, &h) == 2 && w > 0 && h > 0) return synthetic(h, w);
    return read_pgm(a);
}

}  // namespace imgio
