// oracle/ref_shim/opencv2/core.hpp -- TEST INFRASTRUCTURE ONLY.
// A from-scratch, minimal stand-in for the handful of cv::Mat members that
// /root/reference/3rdparty/DBoW2/src/DBoW2/FORB.cpp touches, so that FORB.cpp can be
// compiled from where it lies (OpenCV is not installed in this image).  It is NOT OpenCV
// and implements no OpenCV algorithm; it only owns a row-major byte buffer.
#ifndef PLSLAM_ORACLE_REF_SHIM_OPENCV_CORE
#define PLSLAM_ORACLE_REF_SHIM_OPENCV_CORE
#include <cstddef>
#include <cstring>
#include <memory>
#include <vector>

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
}  // namespace cv
#endif
