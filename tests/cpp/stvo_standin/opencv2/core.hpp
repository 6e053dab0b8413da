// Stand-in for <opencv2/core.hpp> -- TEST INFRASTRUCTURE for tools/pin_stvo (tests/test_pin_tool_cpu.py): just enough of
// cv::Mat for the harness to compile and run without OpenCV.  Not used by the product.
#pragma once
#include <cstdint>
#define CV_8U 0
namespace cv {
struct Mat {
    int rows = 0, cols = 0;
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int /*type*/, void* d) : rows(r), cols(c), data(static_cast<uint8_t*>(d)) {}
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + (size_t)r * cols); }
    bool isContinuous() const { return true; }
};
}  // namespace cv
