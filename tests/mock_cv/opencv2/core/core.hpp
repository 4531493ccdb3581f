// TEST INFRASTRUCTURE ONLY -- a minimal stand-in for <opencv2/core/core.hpp>, just large enough to COMPILE the shims of
// include/shims/ in an image without OpenCV (tests/test_shims_cpu.py, tests/shim_driver.cpp).  It is not used to build any part of
// the reference and nothing of the product includes it: with the real OpenCV the shims compile against the real header.
// Layouts that matter to the shims are the real ones: cv::KeyPoint = 28 bytes, cv::Point2f = 2 floats, Mat::data / step / rows / cols.
#ifndef MOCK_OPENCV_CORE_HPP
#define MOCK_OPENCV_CORE_HPP
#include <cmath>
#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_8UC1 0
#define CV_8UC3 16

namespace cv
{
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<float> Point2f;
typedef Point_<int> Point;
typedef Point_<int> Point2i;
struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
    bool operator!=(const Size& o) const { return !(*this == o); }
};
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {} };

class Exception : public std::runtime_error {
public:
    int code;
    Exception(int c, const std::string& err, const std::string&, const std::string&, int) : std::runtime_error(err), code(c) {}
};

class Mat {
public:
    int rows = 0, cols = 0, type_ = CV_8U;
    size_t step = 0;
    unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int t) { create(r, c, t); }
    Mat(int r, int c, int t, void* ext, size_t st) : rows(r), cols(c), type_(t), step(st), data((unsigned char*)ext) {}
    void create(int r, int c, int t)
    {
        rows = r; cols = c; type_ = t; step = (size_t)c * esz();
        buf = std::make_shared<std::vector<unsigned char>>((size_t)r * step, (unsigned char)0);
        data = buf->data();
    }
    void release() { rows = cols = 0; data = nullptr; buf.reset(); }
    int type() const { return type_; }
    bool empty() const { return data == nullptr || rows * cols == 0; }
    size_t total() const { return (size_t)rows * cols; }
    Size size() const { return Size(cols, rows); }
    template <class T> T* ptr(int r = 0) { return (T*)(data + (size_t)r * step); }
    template <class T> const T* ptr(int r = 0) const { return (const T*)(data + (size_t)r * step); }
    unsigned char* ptr(int r = 0) { return data + (size_t)r * step; }
    template <class T> T& at(int r, int c) { return ((T*)(data + (size_t)r * step))[c]; }
    template <class T> const T& at(int r, int c) const { return ((const T*)(data + (size_t)r * step))[c]; }
    template <class T> T& at(int i) { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    template <class T> const T& at(int i) const { return rows == 1 ? at<T>(0, i) : at<T>(i, 0); }
    Mat sub(int r0, int r1, int c0, int c1) const
    {
        Mat m(r1 - r0, c1 - c0, type_);
        for (int r = r0; r < r1; r++) std::memcpy(m.data + (size_t)(r - r0) * m.step, data + (size_t)r * step + (size_t)c0 * esz(), (size_t)(c1 - c0) * esz());
        return m;
    }
    Mat rowRange(int a, int b) const { return sub(a, b, 0, cols); }
    Mat colRange(int a, int b) const { return sub(0, rows, a, b); }
    Mat row(int r) const { return sub(r, r + 1, 0, cols); }
    Mat col(int c) const { return sub(0, rows, c, c + 1); }
    Mat clone() const { return sub(0, rows, 0, cols); }
    Mat t() const
    {
        Mat m(cols, rows, CV_32F);
        for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) m.at<float>(c, r) = at<float>(r, c);
        return m;
    }
    double dot(const Mat& o) const
    {
        double s = 0;
        for (int r = 0; r < rows; r++) for (int c = 0; c < cols; c++) s += (double)at<float>(r, c) * o.at<float>(r, c);
        return s;
    }
    void convertTo(Mat& dst, int t) const
    {
        if (empty()) { dst = Mat(); return; }
        if (t != CV_32F || type_ != CV_32F) throw std::runtime_error("mock cv::Mat::convertTo: CV_32F only");
        dst = clone();
    }
    void copyTo(Mat& dst) const { dst = clone(); }
private:
    size_t esz() const { return type_ == CV_32F ? 4 : type_ == CV_8UC3 ? 3 : 1; }
    std::shared_ptr<std::vector<unsigned char>> buf;
};
inline Mat operator*(const Mat& a, const Mat& b)
{
    Mat m(a.rows, b.cols, CV_32F);
    for (int r = 0; r < a.rows; r++) for (int c = 0; c < b.cols; c++) {
        float s = 0;
        for (int k = 0; k < a.cols; k++) s += a.at<float>(r, k) * b.at<float>(k, c);
        m.at<float>(r, c) = s;
    }
    return m;
}
inline Mat scale(const Mat& a, double s)
{
    Mat m = a.clone();
    for (int r = 0; r < m.rows; r++) for (int c = 0; c < m.cols; c++) m.at<float>(r, c) = (float)(m.at<float>(r, c) * s);
    return m;
}
inline Mat operator*(double s, const Mat& a) { return scale(a, s); }
inline Mat operator*(const Mat& a, double s) { return scale(a, s); }
inline Mat operator/(const Mat& a, double s) { return scale(a, 1.0 / s); }
inline Mat operator-(const Mat& a) { return scale(a, -1.0); }
inline Mat operator+(const Mat& a, const Mat& b)
{
    Mat m = a.clone();
    for (int r = 0; r < m.rows; r++) for (int c = 0; c < m.cols; c++) m.at<float>(r, c) += b.at<float>(r, c);
    return m;
}

// the proxies of operator()'s signature: an input wraps a Mat (or nothing), an output wraps the caller's Mat
class _InputArray {
public:
    _InputArray() : m(nullptr) {}
    _InputArray(const Mat& mm) : m(&mm) {}
    bool empty() const { return !m || m->empty(); }
    Mat getMat() const { return m ? *m : Mat(); }
private:
    const Mat* m;
};
class _OutputArray {
public:
    _OutputArray(Mat& mm) : m(&mm) {}
    void create(int r, int c, int t) const { m->create(r, c, t); }
    void release() const { m->release(); }
    Mat getMat() const { return *m; } // shares the buffer
private:
    Mat* m;
};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
} // namespace cv
#endif
