// stvo_match.hpp -- C++ host shim: the reference's matcher interface on top of the C ABI.
//
// Mirrors  int StVO::match(const cv::Mat& desc1, const cv::Mat& desc2, float nnr,
//                          std::vector<int>& matches_12)
// of stvo-pl's matching.h (included at src/mapHandler.cpp:28 of pl-slam; call sites
// src/mapHandler.cpp:277,424,597,712,3223,3249): same name, argument meaning, return value
// (number of matches), output convention (matches_12.size() == desc1.rows, -1 = no match,
// src/mapHandler.cpp:280-283) and error behaviour (std::runtime_error, cf. :286-288).
//
// The descriptor arguments are templates so that cv::Mat (rows/cols/isContinuous()/ptr<uchar>())
// works unchanged where OpenCV exists, and StVO::DescMat -- a plain view -- works where it does
// not (this repository's tests).  Header-only; link with libplslam_hip.so.
//
// Threading: the reference calls match() from the VO thread, the local-mapping thread and the
// loop-closure thread at once.  Each calling thread lazily gets its own plslam_ctx (thread_local),
// so concurrent calls never share device scratch.
#pragma once

#include <stdint.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "plslam_hip.h"

namespace StVO {

// minimal stand-in for the cv::Mat descriptor block (N x 32, CV_8U, continuous)
struct DescMat {
    const uint8_t* data = nullptr;
    int rows = 0;
    int cols = PLSLAM_DESC_BYTES;
    DescMat() {}
    DescMat(const uint8_t* d, int r) : data(d), rows(r) {}
    bool isContinuous() const { return true; }
    template <class T> const T* ptr() const { return reinterpret_cast<const T*>(data); }
};

// whether match() runs both directions and keeps mutual matches: stvo-pl Config::bestLRMatches()
// (config/config/config_kitti.yaml:17 `best_lr_matches`).  A process-wide switch like the
// reference's Config singleton.
inline bool& bestLRMatches()
{
    static bool v = true;
    return v;
}

namespace detail {
struct ThreadCtx {
    plslam_ctx* ctx = nullptr;
    ~ThreadCtx() { if (ctx) plslam_ctx_destroy(ctx); }
};
inline int& deviceOrdinal()
{
    static int d = 0;
    return d;
}
inline plslam_ctx* ctx()
{
    static thread_local ThreadCtx t;
    if (!t.ctx) {
        const int rc = plslam_ctx_create(deviceOrdinal(), &t.ctx);
        if (rc != PLSLAM_OK)
            throw std::runtime_error(std::string("[StVO::match] no MI355X context: ") + plslam_strerror(rc) +
                                     " (" + plslam_last_error() + ")");
    }
    return t.ctx;
}
template <class Mat> inline const uint8_t* rows_of(const Mat& m, const char* what)
{
    if (m.rows < 0 || (m.rows > 0 && m.cols != PLSLAM_DESC_BYTES))
        throw std::runtime_error(std::string("[StVO::match] ") + what + " is not N x 32 bytes");
    if (m.rows > 0 && !m.isContinuous())
        throw std::runtime_error(std::string("[StVO::match] ") + what + " is not continuous");
    return m.rows > 0 ? m.template ptr<uint8_t>() : nullptr;
}
inline void check(int rc, const char* fn)
{
    if (rc != PLSLAM_OK)
        throw std::runtime_error(std::string("[") + fn + "] " + plslam_strerror(rc) + ": " + plslam_last_error());
}
}  // namespace detail

// select the HIP device used by the calling process (before the first match())
inline void setDevice(int ordinal) { detail::deviceOrdinal() = ordinal; }

// stvo-pl matchNNR: one directed kNN-2 + ratio test
template <class Mat1, class Mat2>
inline int matchNNR(const Mat1& desc1, const Mat2& desc2, float nnr, std::vector<int>& matches_12)
{
    matches_12.assign((size_t)desc1.rows, -1);
    int32_t n = 0;
    detail::check(plslam_match(detail::ctx(), detail::rows_of(desc1, "desc1"), desc1.rows,
                               detail::rows_of(desc2, "desc2"), desc2.rows, nnr, 0, matches_12.data(), &n),
                  "StVO::matchNNR");
    return n;
}

// stvo-pl match: ratio test both ways + mutual consistency when bestLRMatches()
template <class Mat1, class Mat2>
inline int match(const Mat1& desc1, const Mat2& desc2, float nnr, std::vector<int>& matches_12)
{
    static_assert(sizeof(int) == sizeof(int32_t), "matches_12 is std::vector<int> in the reference");
    matches_12.assign((size_t)desc1.rows, -1);
    int32_t n = 0;
    detail::check(plslam_match(detail::ctx(), detail::rows_of(desc1, "desc1"), desc1.rows,
                               detail::rows_of(desc2, "desc2"), desc2.rows, nnr, bestLRMatches() ? 1 : 0,
                               matches_12.data(), &n),
                  "StVO::match");
    return n;
}

// A batch of independent match() calls in one launch (e.g. the four per-frame problems of
// StereoFrame L<->R + f2fTracking, or every frame of an offline sequence).
struct MatchJob {
    DescMat desc1, desc2;
};
inline std::vector<int> matchBatch(const std::vector<MatchJob>& jobs, float nnr,
                                   std::vector<std::vector<int>>& matches_12)
{
    std::vector<int32_t> off1(jobs.size() + 1, 0), off2(jobs.size() + 1, 0);
    for (size_t b = 0; b < jobs.size(); ++b) {
        off1[b + 1] = off1[b] + jobs[b].desc1.rows;
        off2[b + 1] = off2[b] + jobs[b].desc2.rows;
    }
    std::vector<uint8_t> d1((size_t)off1.back() * 32), d2((size_t)off2.back() * 32);
    for (size_t b = 0; b < jobs.size(); ++b) {
        const uint8_t* p1 = detail::rows_of(jobs[b].desc1, "desc1");
        const uint8_t* p2 = detail::rows_of(jobs[b].desc2, "desc2");
        if (p1) std::copy(p1, p1 + (size_t)jobs[b].desc1.rows * 32, d1.begin() + (size_t)off1[b] * 32);
        if (p2) std::copy(p2, p2 + (size_t)jobs[b].desc2.rows * 32, d2.begin() + (size_t)off2[b] * 32);
    }
    std::vector<int32_t> flat((size_t)off1.back()), counts(jobs.size());
    detail::check(plslam_match_batched(detail::ctx(), d1.data(), off1.data(), d2.data(), off2.data(),
                                       (int32_t)jobs.size(), nnr, bestLRMatches() ? 1 : 0, flat.data(),
                                       counts.data()),
                  "StVO::matchBatch");
    matches_12.resize(jobs.size());
    for (size_t b = 0; b < jobs.size(); ++b)
        matches_12[b].assign(flat.begin() + off1[b], flat.begin() + off1[b + 1]);
    return std::vector<int>(counts.begin(), counts.end());
}

}  // namespace StVO
