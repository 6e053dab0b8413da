// oracle/ref_shim/opencv2/core.hpp -- TEST INFRASTRUCTURE ONLY.
// A from-scratch, minimal stand-in for the handful of cv:: names that
// /root/reference/3rdparty/DBoW2/src/DBoW2/FORB.cpp and
// /root/reference/3rdparty/line_descriptor/src/binary_descriptor_matcher.cpp (+ the class declarations in
// include/line_descriptor/descriptor_custom.hpp) touch, so that those files can be compiled from where they lie
// (OpenCV is not installed in this image).  It is NOT OpenCV and implements no OpenCV algorithm: Mat owns a
// row-major byte buffer, the other types are plain records or empty tags that let declarations parse.
#ifndef PLSLAM_ORACLE_REF_SHIM_OPENCV_CORE
#define PLSLAM_ORACLE_REF_SHIM_OPENCV_CORE
#include <cfloat>
#include <math.h>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define CV_EXPORTS
#define CV_EXPORTS_W
#define CV_WRAP
#define CV_OUT
#define CV_IN_OUT
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert failed: " #expr); } while (0)
typedef unsigned char uchar;

#define CV_8U 0
#define CV_32F 5

namespace cv {
class Mat {
public:
    int rows = 0, cols = 0, type_ = CV_8U;
    Mat() {}
    static size_t esz(int type) { return type == CV_32F ? 4 : 1; }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        buf_ = std::shared_ptr<unsigned char>(new unsigned char[(size_t)r * c * esz(type) + 8],
                                              std::default_delete<unsigned char[]>());
    }
    static Mat zeros(int r, int c, int type) {
        Mat m; m.create(r, c, type);
        std::memset(m.buf_.get(), 0, (size_t)r * c * esz(type));
        return m;
    }
    void release() { buf_.reset(); rows = cols = 0; }
    bool empty() const { return !buf_ || rows * cols == 0; }
    Mat clone() const {
        Mat m; m.create(rows, cols, type_);
        if (buf_) std::memcpy(m.buf_.get(), buf_.get(), (size_t)rows * cols * esz(type_));
        return m;
    }
    template <class T> T* ptr() { return reinterpret_cast<T*>(buf_.get()); }
    template <class T> const T* ptr() const { return reinterpret_cast<const T*>(buf_.get()); }
    unsigned char* ptr() { return buf_.get(); }
    const unsigned char* ptr() const { return buf_.get(); }
    unsigned char* ptr(int r) { return buf_.get() + (size_t)r * cols * esz(type_); }
    const unsigned char* ptr(int r) const { return buf_.get() + (size_t)r * cols * esz(type_); }
    template <class T> T& at(int r, int c) { return reinterpret_cast<T*>(ptr(r))[c]; }
    template <class T> const T& at(int r, int c) const { return reinterpret_cast<const T*>(ptr(r))[c]; }
    template <class T> T& at(int i) { return reinterpret_cast<T*>(buf_.get())[i]; }
    template <class T> const T& at(int i) const { return reinterpret_cast<const T*>(buf_.get())[i]; }
    int type() const { return type_; }
    Mat row(int r) const {
        Mat m; m.create(1, cols, type_);
        std::memcpy(m.buf_.get(), ptr(r), (size_t)cols * esz(type_));
        return m;
    }
    void push_back(const Mat& o) {                   // append rows (same width and type)
        if (o.rows == 0) return;
        Mat m; m.create(rows + o.rows, o.cols, o.type_);
        if (rows) std::memcpy(m.buf_.get(), buf_.get(), (size_t)rows * cols * esz(type_));
        std::memcpy(m.buf_.get() + (size_t)rows * o.cols * esz(o.type_), o.buf_.get(), (size_t)o.rows * o.cols * esz(o.type_));
        *this = m;
    }
    void convertTo(Mat& dst, int type) const {
        dst.create(rows, cols, type);
        if (type == CV_32F && type_ == CV_8U) {
            float* o = dst.ptr<float>();
            const unsigned char* s = ptr<unsigned char>();
            for (size_t i = 0; i < (size_t)rows * cols; ++i) o[i] = (float)s[i];
        }
    }
private:
    std::shared_ptr<unsigned char> buf_;
};
template <class T> class Mat_ : public Mat {};
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<float> Point2f;
typedef Point_<int> Point;
struct Size { int width, height; Size() : width(0), height(0) {} Size(int w, int h) : width(w), height(h) {} };
struct Scalar { double v[4]; static Scalar all(double a) { Scalar s; s.v[0] = s.v[1] = s.v[2] = s.v[3] = a; return s; } };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.4e38f) {}
    bool operator<(const DMatch& m) const { return distance < m.distance; }
};
typedef std::string String;
class FileStorage {};
class FileNode {};
class _InputArray {};
class _OutputArray {};
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
class Algorithm { public: virtual ~Algorithm() {} };
template <class T> using Ptr = std::shared_ptr<T>;
template <class T> Ptr<T> makePtr() { return std::make_shared<T>(); }
}  // namespace cv
#endif
