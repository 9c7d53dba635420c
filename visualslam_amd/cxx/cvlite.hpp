// Minimal cv::Mat-compatible image type for the subset of OpenCV the reference's hot path uses
// (SURVEY.md section 7 step 1).  OpenCV is not available in the build image; when it is
// (define VSLAM_USE_OPENCV and link it) the real cv::Mat is used instead and this header only
// provides the small adaptors.  Ref-counted header + shared buffer, like cv::Mat.
#pragma once
#ifdef VSLAM_USE_OPENCV
#include <opencv2/core.hpp>
#else
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

namespace cv {

enum { CV_8U = 0, CV_32F = 5 };
typedef unsigned char uchar;

struct Size {
    int width = 0, height = 0;
    Size() = default;
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
};

template <typename T>
struct Point_ {
    T x{}, y{};
    Point_() = default;
    Point_(T x_, T y_) : x(x_), y(y_) {}
    Point_& operator-=(const Point_& o) {
        x -= o.x, y -= o.y;
        return *this;
    }
    bool operator==(const Point_& o) const { return x == o.x && y == o.y; }
};
typedef Point_<int> Point2i;
typedef Point_<float> Point2f;
typedef Point2i Point;

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;  // bytes per row
    uchar* data = nullptr;

    Mat() = default;
    Mat(int r, int c, int type) { create(r, c, type); }
    static Mat zeros(int r, int c, int type) {
        Mat m(r, c, type);
        if (m.data) std::memset(m.data, 0, m.step * (size_t)r);
        return m;
    }
    void create(int r, int c, int type) {
        if (r < 0 || c < 0 || (type != CV_8U && type != CV_32F)) throw std::invalid_argument("cv::Mat::create");
        rows = r, cols = c, type_ = type;
        step = (size_t)c * elemSize();
        buf_ = std::shared_ptr<uchar>(new uchar[step * (size_t)r + 16], std::default_delete<uchar[]>());
        data = buf_.get();
    }
    int type() const { return type_; }
    int depth() const { return type_; }
    size_t elemSize() const { return type_ == CV_32F ? 4 : 1; }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    bool isContinuous() const { return step == (size_t)cols * elemSize(); }
    Size size() const { return Size(cols, rows); }
    size_t total() const { return (size_t)rows * cols; }
    Mat clone() const {
        Mat m(rows, cols, type_);
        for (int r = 0; r < rows; ++r) std::memcpy(m.data + m.step * r, data + step * r, (size_t)cols * elemSize());
        return m;
    }
    template <typename T>
    T& at(int r, int c) { return *reinterpret_cast<T*>(data + step * r + sizeof(T) * c); }
    template <typename T>
    const T& at(int r, int c) const { return *reinterpret_cast<const T*>(data + step * r + sizeof(T) * c); }
    template <typename T>
    T* ptr(int r = 0) { return reinterpret_cast<T*>(data + step * r); }
    template <typename T>
    const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + step * r); }

private:
    int type_ = CV_8U;
    std::shared_ptr<uchar> buf_;
};

}  // namespace cv
#endif
