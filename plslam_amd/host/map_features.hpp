// map_features.hpp -- C++ host mirror of two producer steps next to the matcher:
//
//  * PLSLAM::updateAverageDescriptors(): the descriptor part of MapPoint::updateAverageDescDir /
//    MapLine::updateAverageDescDir (src/mapFeatures.cpp:51-84, :121-157) for ALL landmarks of a local
//    map in one call (the reference runs it per landmark on every addMap{Point,Line}Observation, :47, :118).
//  * cv::line_descriptor-side: PLSLAM::binaryDescriptorRows(): the "fill current row with binary
//    descriptor" loop of BinaryDescriptor::computeImpl (3rdparty/line_descriptor/src/
//    binary_descriptor_custom.cpp:653-668, binaryConversion :401-412).
//
// Header-only over the C ABI (include/plslam_hip.h); errors surface as std::runtime_error like the
// reference's own (src/mapHandler.cpp:286-288).  Uses the calling thread's context of stvo_match.hpp.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "stvo_match.hpp"

namespace PLSLAM {

// One landmark's desc_list: n rows of 32 bytes (cv::Mat rows of CV_8U), in observation order.
struct DescList {
    const uint8_t* const* rows = nullptr;   // rows[k] -> 32 bytes   (desc_list[k].ptr<uchar>())
    int n = 0;
};

// med_idx[l] = index into landmark l's desc_list of its representative descriptor (0 for a single
// observation, -1 for an empty list); med_desc (optional) = those rows, n_landmarks x 32, ready to be the
// `med_desc` argument of plslam_map2kf_match_points / _lines.
inline void updateAverageDescriptors(const std::vector<DescList>& landmarks, std::vector<int>& med_idx,
                                     std::vector<uint8_t>* med_desc = nullptr)
{
    std::vector<int32_t> off(landmarks.size() + 1, 0);
    for (size_t l = 0; l < landmarks.size(); ++l) off[l + 1] = off[l] + landmarks[l].n;
    std::vector<uint8_t> lists((size_t)off.back() * 32);
    for (size_t l = 0; l < landmarks.size(); ++l)
        for (int k = 0; k < landmarks[l].n; ++k)
            std::copy(landmarks[l].rows[k], landmarks[l].rows[k] + 32, &lists[((size_t)off[l] + k) * 32]);
    std::vector<int32_t> idx(landmarks.size());
    if (med_desc) med_desc->assign(landmarks.size() * 32, 0);
    const int rc = plslam_median_desc_batched(StVO::detail::ctx(), lists.data(), off.data(), (int32_t)landmarks.size(),
                                              idx.data(), med_desc ? med_desc->data() : nullptr);
    if (rc != PLSLAM_OK)
        throw std::runtime_error(std::string("[MapPoint::updateAverageDescDir] ") + plslam_strerror(rc) + ": " +
                                 plslam_last_error());
    med_idx.assign(idx.begin(), idx.end());
}

// lbd: n x 72 float (the ScaleLines' `descriptor` vectors back to back) -> n x 32 uint8 rows
inline void binaryDescriptorRows(const float* lbd, int n, uint8_t* rows)
{
    const int rc = plslam_lbd_binarise(StVO::detail::ctx(), lbd, n, rows);
    if (rc != PLSLAM_OK)
        throw std::runtime_error(std::string("[BinaryDescriptor::computeImpl] ") + plslam_strerror(rc) + ": " +
                                 plslam_last_error());
}

}  // namespace PLSLAM
