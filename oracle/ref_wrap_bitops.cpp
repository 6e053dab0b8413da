// oracle/ref_wrap_bitops.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/).
// Exposes the reference's own 256-bit popcount matcher,
//   /root/reference/3rdparty/line_descriptor/src/bitops_custom.hpp:83-96  cv::line_descriptor::match
// compiled from where it lies (the Makefile passes -I<ref>/3rdparty/line_descriptor/src and
// -D__OPENCV_PRECOMP_H__ so that the OpenCV-dependent precompiled header is skipped).
#include <stdint.h>
#include <stdio.h>
#include "types_custom.hpp"
#include "bitops_custom.hpp"

extern "C" int ref_ld_match(const uint8_t* p, const uint8_t* q, int code_bytes)
{
    return cv::line_descriptor::match((UINT8*)p, (UINT8*)q, code_bytes);
}
