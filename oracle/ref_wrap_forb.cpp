// oracle/ref_wrap_forb.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/).
// Exposes the reference's DBoW2::FORB::distance
//   /root/reference/3rdparty/DBoW2/src/DBoW2/FORB.cpp:78-101
// (the Makefile compiles FORB.cpp itself from where it lies, against the cv::Mat stand-in
// oracle/ref_shim/opencv2/core.hpp).
#include <stdint.h>
#include <string.h>
#include "DBoW2/FORB.h"

extern "C" int ref_forb_distance(const uint8_t* a, const uint8_t* b)
{
    cv::Mat ma, mb;
    ma.create(1, DBoW2::FORB::L, CV_8U);
    mb.create(1, DBoW2::FORB::L, CV_8U);
    memcpy(ma.ptr<unsigned char>(), a, DBoW2::FORB::L);
    memcpy(mb.ptr<unsigned char>(), b, DBoW2::FORB::L);
    return (int)DBoW2::FORB::distance(ma, mb);
}
