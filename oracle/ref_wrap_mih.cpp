// oracle/ref_wrap_mih.cpp -- TEST INFRASTRUCTURE ONLY (built into oracle/_ref/).
// Exposes the reference's exact multi-index-hashing kNN search over 256-bit descriptors,
//   cv::line_descriptor::BinaryDescriptorMatcher::knnMatch(query, train, matches, k, mask, compact)
//   /root/reference/3rdparty/line_descriptor/src/binary_descriptor_matcher.cpp:258-335 (engine :596-971),
// the only kNN code that exists IN the reference tree (the matcher pl-slam actually calls, cv::BFMatcher via
// stvo-pl, is not vendored).  The Makefile compiles binary_descriptor_matcher.cpp itself from where it lies,
// against the cv:: stand-in oracle/ref_shim/opencv2/core.hpp.  Used by tests/test_oracle_pin.py to pin the
// oracle's k-nearest DISTANCES (an exact search must return the same distance multiset; which of several
// equally distant rows it names is the engine's own business and is checked only for consistency).
#include <stdint.h>
#include <string.h>
#include <vector>
#include "line_descriptor/descriptor_custom.hpp"

extern "C" int ref_mih_knn(const uint8_t* q, int nq, const uint8_t* t, int nt, int k, int32_t* idx, int32_t* dist)
{
    if (nq <= 0 || nt <= 0 || k <= 0 || k > nt) return -1;
    cv::Mat mq, mt, mask;
    mq.create(nq, 32, CV_8U);
    mt.create(nt, 32, CV_8U);
    memcpy(mq.ptr(), q, (size_t)nq * 32);
    memcpy(mt.ptr(), t, (size_t)nt * 32);
    cv::line_descriptor::BinaryDescriptorMatcher m;
    std::vector<std::vector<cv::DMatch> > out;
    m.knnMatch(mq, mt, out, k, mask, false);
    if ((int)out.size() != nq) return -2;
    for (int i = 0; i < nq; ++i) {
        if ((int)out[i].size() != k) return -3;
        for (int j = 0; j < k; ++j) {
            idx[(size_t)i * k + j] = out[i][j].trainIdx;
            dist[(size_t)i * k + j] = (int32_t)out[i][j].distance;
        }
    }
    return 0;
}
