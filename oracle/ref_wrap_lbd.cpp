// oracle/ref_wrap_lbd.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/).
// Runs the reference's OWN line-band-descriptor code on caller-supplied gradient images:
//   cv::line_descriptor::BinaryDescriptor::BinaryDescriptor(Params)   (Gaussian weight tables)
//       /root/reference/3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:217-259
//   BinaryDescriptor::computeLBD(ScaleLines&, useDetectionData = false)                       :1026-1372
//   BinaryDescriptor::binaryConversion(float*, float*)                                         :401-412
// The Makefile compiles binary_descriptor_custom.cpp itself from where it lies against the cv:: stand-in
// oracle/ref_shim/ (its image-processing calls -- Sobel, GaussianBlur, ... -- are declaration-only there and are
// never reached: the tests hand computeLBD the gradient images directly, exactly the state computeSobel would have
// left in dxImg_vector / dyImg_vector / images_sizes).  The members involved are private: this TU (test
// infrastructure) opens them with the usual #define, which does not change the class layout under GCC.
#include <stdint.h>
#include <string.h>
#include <vector>
#define private public
#define protected public
#include "line_descriptor/descriptor_custom.hpp"
#undef private
#undef protected

using cv::line_descriptor::BinaryDescriptor;

struct ref_lbd_line {            // = plo_lbd_line (oracle/plslam_oracle.h)
    int32_t num_pixels;
    float sx, sy, ex, ey, direction;
};

static BinaryDescriptor::Params make_params(int width_of_band)
{
    BinaryDescriptor::Params p;
    p.numOfOctave_ = 1;
    p.widthOfBand_ = width_of_band;
    return p;
}

extern "C" int ref_lbd_compute(const int16_t* dx, const int16_t* dy, int width, int height, const ref_lbd_line* lines, int n,
                               int width_of_band, float* lbd /* n x 72 */)
{
    BinaryDescriptor bd(make_params(width_of_band));
    bd.dxImg_vector.resize(1);
    bd.dyImg_vector.resize(1);
    bd.dxImg_vector[0].create(height, width, CV_16SC1);
    bd.dyImg_vector[0].create(height, width, CV_16SC1);
    memcpy(bd.dxImg_vector[0].ptr(), dx, (size_t)width * height * 2);
    memcpy(bd.dyImg_vector[0].ptr(), dy, (size_t)width * height * 2);
    bd.images_sizes.assign(1, cv::Size(width, height));
    BinaryDescriptor::ScaleLines sl((size_t)n);
    for (int i = 0; i < n; ++i) {
        BinaryDescriptor::OctaveSingleLine l;
        l.startPointX = l.sPointInOctaveX = lines[i].sx;
        l.startPointY = l.sPointInOctaveY = lines[i].sy;
        l.endPointX = l.ePointInOctaveX = lines[i].ex;
        l.endPointY = l.ePointInOctaveY = lines[i].ey;
        l.direction = lines[i].direction;
        l.salience = 0.f;
        l.lineLength = 0.f;
        l.numOfPixels = (unsigned int)lines[i].num_pixels;
        l.octaveCount = 0;
        sl[i].push_back(l);
    }
    bd.computeLBD(sl, false);
    for (int i = 0; i < n; ++i) {
        const std::vector<float>& d = sl[i][0].descriptor;
        if (d.size() != 72) return -1;
        memcpy(lbd + (size_t)i * 72, d.data(), 72 * sizeof(float));
    }
    return 0;
}

extern "C" int ref_lbd_gauss_tables(int width_of_band, double* coef_l /* 3 w */, double* coef_g /* 9 w */)
{
    BinaryDescriptor bd(make_params(width_of_band));
    if ((int)bd.gaussCoefL_.size() != 3 * width_of_band || (int)bd.gaussCoefG_.size() != 9 * width_of_band) return -1;
    memcpy(coef_l, bd.gaussCoefL_.data(), bd.gaussCoefL_.size() * sizeof(double));
    memcpy(coef_g, bd.gaussCoefG_.data(), bd.gaussCoefG_.size() * sizeof(double));
    return 0;
}

extern "C" int ref_lbd_binary_conversion(const float* f1, const float* f2)
{
    BinaryDescriptor bd(make_params(7));
    float a[8], b[8];
    memcpy(a, f1, sizeof a);
    memcpy(b, f2, sizeof b);
    return (int)bd.binaryConversion(a, b);
}
