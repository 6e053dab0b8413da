// oracle/ref_shim/opencv2/core.hpp -- TEST INFRASTRUCTURE ONLY.
// A from-scratch, minimal stand-in for the handful of cv:: names that
// /root/reference/3rdparty/DBoW2/src/DBoW2/FORB.cpp and
// /root/reference/3rdparty/line_descriptor/src/binary_descriptor_matcher.cpp (+ the class declarations in
// include/line_descriptor/descriptor_custom.hpp) touch, so that those files can be compiled from where they lie
// (OpenCV is not installed in this image).  It is NOT OpenCV and implements no OpenCV algorithm: Mat owns a
// row-major byte buffer, the other types are plain records or empty tags that let declarations parse.
#ifndef PLSLAM_ORACLE_REF_SHIM_OPENCV_CORE
#define PLSLAM_ORACLE_REF_SHIM_OPENCV_CORE
#include <algorithm>
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
#define CV_8S 1
#define CV_16S 3
#define CV_32S 4
#define CV_32F 5
#define CV_8UC1 CV_8U
#define CV_8SC1 CV_8S
#define CV_16SC1 CV_16S
#define CV_32FC1 CV_32F
#define PLSLAM_SHIM_NOT_IMPLEMENTED(what) throw std::logic_error("oracle/ref_shim: " what " is a declaration-only stand-in (no OpenCV here)")

namespace cv {
class Mat {
public:
    int rows = 0, cols = 0, type_ = CV_8U;
    unsigned char* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    static size_t esz(int type) { return type == CV_32F || type == CV_32S ? 4 : (type == CV_16S ? 2 : 1); }
    void create(int r, int c, int type) {
        rows = r; cols = c; type_ = type;
        buf_ = std::shared_ptr<unsigned char>(new unsigned char[(size_t)r * c * esz(type) + 8],
                                              std::default_delete<unsigned char[]>());
        data = buf_.get();
    }
    static Mat zeros(int r, int c, int type) {
        Mat m; m.create(r, c, type);
        std::memset(m.buf_.get(), 0, (size_t)r * c * esz(type));
        return m;
    }
    void release() { buf_.reset(); rows = cols = 0; data = nullptr; }
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
    int depth() const { return type_; }
    int channels() const { return 1; }
    template <class T> T* ptr(int r) { return reinterpret_cast<T*>(ptr(r)); }
    template <class T> const T* ptr(int r) const { return reinterpret_cast<const T*>(ptr(r)); }
    struct MSize {
        int width, height;
        bool operator==(const MSize& o) const { return width == o.width && height == o.height; }
        bool operator!=(const MSize& o) const { return !(*this == o); }
    };
    MSize size() const { MSize z; z.width = cols; z.height = rows; return z; }
    void setTo(int v) { if (data) std::memset(data, v, (size_t)rows * cols * esz(type_)); }
    template <class O> void copyTo(O&) const { PLSLAM_SHIM_NOT_IMPLEMENTED("Mat::copyTo"); }
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
inline Mat operator/(const Mat&, int) { PLSLAM_SHIM_NOT_IMPLEMENTED("Mat / int"); }
template <class T> class Mat_ : public Mat {
public:
    Mat_() {}
    Mat_(int r, int c) { create(r, c, sizeof(T) == 4 ? CV_32F : CV_8U); }
    template <class U> Mat_(const Mat_<U>& o) { create(o.rows, o.cols, sizeof(T) == 4 ? CV_32F : CV_8U); }
    T* operator[](int r) { return reinterpret_cast<T*>(ptr(r)); }
    const T* operator[](int r) const { return reinterpret_cast<const T*>(ptr(r)); }
    Mat_ t() const { PLSLAM_SHIM_NOT_IMPLEMENTED("Mat_::t"); }
};
template <class T> Mat_<T> operator+(const Mat_<T>&, const Mat_<T>&) { PLSLAM_SHIM_NOT_IMPLEMENTED("Mat_ + Mat_"); }
template <class T> Mat_<T> operator*(const Mat_<T>&, const Mat_<T>&) { PLSLAM_SHIM_NOT_IMPLEMENTED("Mat_ * Mat_"); }
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<float> Point2f;
typedef Point_<int> Point;
struct Size {
    int width, height;
    Size() : width(0), height(0) {}
    Size(int w, int h) : width(w), height(h) {}
    Size(const Mat::MSize& m) : width(m.width), height(m.height) {}
};
struct Scalar { double v[4]; static Scalar all(double a) { Scalar s; s.v[0] = s.v[1] = s.v[2] = s.v[3] = a; return s; } };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };
struct DMatch {
    int queryIdx, trainIdx, imgIdx;
    float distance;
    DMatch() : queryIdx(-1), trainIdx(-1), imgIdx(-1), distance(3.4e38f) {}
    bool operator<(const DMatch& m) const { return distance < m.distance; }
};
typedef std::string String;
class FileStorage {
public:
    template <class T> FileStorage& operator<<(const T&) { return *this; }
};
class FileNode {
public:
    FileNode operator[](const char*) const { return FileNode(); }
    operator int() const { return 0; }
};
class _InputArray { public: Mat getMat() const { PLSLAM_SHIM_NOT_IMPLEMENTED("InputArray::getMat"); } };
class _OutputArray {};
enum { COLOR_BGR2GRAY = 6, THRESH_TOZERO = 3, CMP_LT = 3, NORM_HAMMING = 6 };
using std::max;
// The one OpenCV routine this stand-in IMPLEMENTS (src/mapFeatures.cpp:63,133 call it): the Hamming norm of two byte
// rows = bit count of their XOR.  The distance itself is pinned separately against the reference's own popcount code.
inline double norm(const Mat& a, const Mat& b, int type) {
    if (type != NORM_HAMMING || a.rows * a.cols != b.rows * b.cols) PLSLAM_SHIM_NOT_IMPLEMENTED("norm (only NORM_HAMMING)");
    int d = 0;
    for (int i = 0; i < a.rows * a.cols; ++i) d += __builtin_popcount((unsigned)(a.ptr()[i] ^ b.ptr()[i]));
    return d;
}
// image-processing entry points the reference's detector code mentions: declarations that let it compile; the tests
// only ever run code that does not reach them
inline void Sobel(const Mat&, Mat&, int, int, int, int) { PLSLAM_SHIM_NOT_IMPLEMENTED("Sobel"); }
inline void GaussianBlur(const Mat&, Mat&, Size, double) { PLSLAM_SHIM_NOT_IMPLEMENTED("GaussianBlur"); }
inline void resize(const Mat&, Mat&, Size, double, double) { PLSLAM_SHIM_NOT_IMPLEMENTED("resize"); }
inline void cvtColor(const Mat&, Mat&, int) { PLSLAM_SHIM_NOT_IMPLEMENTED("cvtColor"); }
inline void pyrDown(const Mat&, Mat&, Size) { PLSLAM_SHIM_NOT_IMPLEMENTED("pyrDown"); }
inline double threshold(const Mat&, Mat&, double, double, int) { PLSLAM_SHIM_NOT_IMPLEMENTED("threshold"); }
inline void compare(const Mat&, const Mat&, Mat&, int) { PLSLAM_SHIM_NOT_IMPLEMENTED("compare"); }
inline void add(const Mat&, const Mat&, Mat&) { PLSLAM_SHIM_NOT_IMPLEMENTED("add"); }
inline Mat abs(const Mat&) { PLSLAM_SHIM_NOT_IMPLEMENTED("abs"); }
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
class Algorithm { public: virtual ~Algorithm() {} };
template <class T> using Ptr = std::shared_ptr<T>;
template <class T> Ptr<T> makePtr() { return std::make_shared<T>(); }
}  // namespace cv
#endif
