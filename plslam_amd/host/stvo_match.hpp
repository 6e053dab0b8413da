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

#include <cmath>
#include <list>
#include <mutex>
#include <stdexcept>
#include <atomic>
#include <string>
#include <utility>
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
// One context (stream + scratch buffers) per calling thread, as the reference's threads each own their cv::BFMatcher.
// Contexts live in a process-wide pool: a thread that ends hands its context back (no HIP call runs in a thread_local
// destructor -- for the main thread that would be during static destruction, possibly after the HIP runtime is gone),
// the next new thread re-uses it (a short-lived worker does not pay a context create + destroy).  Nothing is destroyed
// at process exit; StVO::shutdown() does it explicitly.
struct CtxPool {
    std::mutex mu;
    std::vector<std::pair<int, plslam_ctx*>> idle;   // (device, context) handed back by finished threads
    std::vector<plslam_ctx*> all;
    unsigned generation = 0;                         // bumped by shutdown(): handles cached by threads are stale then
};
inline CtxPool& pool()
{
    static CtxPool* p = new CtxPool();               // deliberately never deleted
    return *p;
}
struct ThreadCtx {
    plslam_ctx* ctx = nullptr;
    int device = -1;
    unsigned generation = 0;
    void release()
    {
        if (!ctx) return;
        CtxPool& P = pool();
        std::lock_guard<std::mutex> lk(P.mu);
        if (generation == P.generation) P.idle.emplace_back(device, ctx);
        ctx = nullptr;
    }
    ~ThreadCtx() { release(); }
};
// read by every matching thread (the reference calls match() from three: src/mapHandler.cpp:1042-1057), written by setDevice()
inline std::atomic<int>& deviceOrdinal()
{
    static std::atomic<int> d{0};
    return d;
}
inline plslam_ctx* ctx()
{
    static thread_local ThreadCtx t;
    CtxPool& P = pool();
    const int want = deviceOrdinal().load(std::memory_order_relaxed);
    {
        std::lock_guard<std::mutex> lk(P.mu);
        if (t.ctx && t.generation != P.generation) t.ctx = nullptr;      // destroyed by shutdown()
    }
    if (t.ctx && t.device != want) t.release();                          // setDevice() since the last call
    if (!t.ctx) {
        {
            std::lock_guard<std::mutex> lk(P.mu);
            for (size_t k = 0; k < P.idle.size(); ++k)
                if (P.idle[k].first == want) {
                    t.ctx = P.idle[k].second;
                    P.idle.erase(P.idle.begin() + (long)k);
                    break;
                }
            t.generation = P.generation;
        }
        if (!t.ctx) {
            // a library built from another revision of the header reads plslam_match_problem arrays with another stride
            if (plslam_abi_version() != PLSLAM_ABI_VERSION)
                throw std::runtime_error("[StVO::match] libplslam_hip.so has ABI version " + std::to_string(plslam_abi_version()) +
                                         ", this translation unit was compiled against " + std::to_string(PLSLAM_ABI_VERSION));
            const int rc = plslam_ctx_create(want, &t.ctx);
            if (rc != PLSLAM_OK) {
                t.ctx = nullptr;
                throw std::runtime_error(std::string("[StVO::match] no MI355X context: ") + plslam_strerror(rc) +
                                         " (" + plslam_last_error() + ")");
            }
            std::lock_guard<std::mutex> lk(P.mu);
            P.all.push_back(t.ctx);
        }
        t.device = want;
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

// select the HIP device used by the calling process (takes effect on each thread's next call)
inline void setDevice(int ordinal) { detail::deviceOrdinal().store(ordinal, std::memory_order_relaxed); }

// Destroys every context the drop-in created (streams, device and page-locked buffers).  Call it when no thread is
// inside a StVO:: function -- e.g. at the end of main(), before static destruction; without it the contexts are simply
// left to process exit.  A later StVO:: call creates fresh ones.
inline void shutdown()
{
    detail::CtxPool& P = detail::pool();
    std::vector<plslam_ctx*> victims;
    {
        std::lock_guard<std::mutex> lk(P.mu);
        victims.swap(P.all);
        P.idle.clear();
        ++P.generation;
    }
    for (plslam_ctx* c : victims) plslam_ctx_destroy(c);
}

namespace detail {
// [RECALL] stvo-pl matchNNR opens with `matches_12.resize(desc1.rows, -1)`: a vector that already holds entries keeps
// them on rows the ratio test rejects (the fall-back after matchGrid, src/mapHandler.cpp:271+277, :418+424, :591+597,
// :706+712).  A vector without any entry >= 0 takes the plain path.
template <class Mat1, class Mat2>
inline int match_impl(const Mat1& desc1, const Mat2& desc2, float nnr, int mutual, std::vector<int>& matches_12,
                      const char* fn)
{
    static_assert(sizeof(int) == sizeof(int32_t), "matches_12 is std::vector<int> in the reference");
    matches_12.resize((size_t)desc1.rows, -1);
    bool prior = false;
    for (int v : matches_12) prior = prior || v >= 0;
    int32_t n = 0;
    check((prior ? plslam_match_prior : plslam_match)(ctx(), rows_of(desc1, "desc1"), desc1.rows, rows_of(desc2, "desc2"),
                                                      desc2.rows, nnr, mutual, matches_12.data(), &n),
          fn);
    return n;
}
}  // namespace detail

// stvo-pl matchNNR: one directed kNN-2 + ratio test
template <class Mat1, class Mat2>
inline int matchNNR(const Mat1& desc1, const Mat2& desc2, float nnr, std::vector<int>& matches_12)
{
    return detail::match_impl(desc1, desc2, nnr, 0, matches_12, "StVO::matchNNR");
}

// stvo-pl match: ratio test both ways + mutual consistency when bestLRMatches()
template <class Mat1, class Mat2>
inline int match(const Mat1& desc1, const Mat2& desc2, float nnr, std::vector<int>& matches_12)
{
    return detail::match_impl(desc1, desc2, nnr, bestLRMatches() ? 1 : 0, matches_12, "StVO::match");
}

// ---------------------------------------------------------------------------------------------------------------
// The windowed matcher: stvo-pl gridStructure.h / matching.h ([RECALL] -- the dependency is not vendored), as used
// by src/mapHandler.cpp:252-271 (points), :381-418 (lines), :580-591, :683-706.  Same names, same argument meaning.
// ---------------------------------------------------------------------------------------------------------------
#ifndef GRID_ROWS
#define GRID_ROWS 48
#endif
#ifndef GRID_COLS
#define GRID_COLS 64
#endif

typedef std::pair<int, int> point_2d;            // a grid cell (x, y); doubles passed to make_pair truncate toward 0
typedef std::pair<point_2d, point_2d> line_2d;   // start and end cell of a segment

struct GridWindow {
    std::pair<int, int> width, height;           // cells x - width.first .. x + width.second, y likewise
};

// Config::lineSimTh() of stvo-pl (`line_sim_th`, config/config/config_kitti.yaml): minimum |cos| between the
// directions of a query line and a candidate; process-wide like the reference's Config singleton
inline double& lineSimTh()
{
    static double v = 0.75;
    return v;
}

inline void normalize(std::pair<double, double>& v)
{
    const double magnitude = std::sqrt(v.first * v.first + v.second * v.second);
    v.first /= magnitude;
    v.second /= magnitude;
}

// Bresenham cells of a segment given in (real-valued) grid units; the last x is excluded
inline void getLineCoords(double x1, double y1, double x2, double y2, std::list<point_2d>& line_coords)
{
    line_coords.clear();
    const bool steep = std::fabs(y2 - y1) > std::fabs(x2 - x1);
    if (steep) { std::swap(x1, y1); std::swap(x2, y2); }
    if (x1 > x2) { std::swap(x1, x2); std::swap(y1, y2); }
    const double dx = x2 - x1, dy = std::fabs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int y = (int)y1;
    const int maxX = (int)x2;
    for (int x = (int)x1; x < maxX; x++) {
        if (steep) line_coords.push_back(std::make_pair(y, x));
        else line_coords.push_back(std::make_pair(x, y));
        error -= dy;
        if (error < 0) { y += ystep; error += dx; }
    }
}

// grid[x][y], 0 <= x < cols, 0 <= y < rows; at() outside the grid returns a dummy list whose content is never seen
class GridStructure {
public:
    int rows, cols;
    GridStructure(int rows_, int cols_) : rows(rows_), cols(cols_)
    {
        if (rows <= 0 || cols <= 0) throw std::runtime_error("[GridStructure] invalid dimension");
        grid.resize((size_t)cols, std::vector<std::list<int>>((size_t)rows));
    }
    std::list<int>& at(int x, int y)
    {
        if (x >= 0 && x < cols && y >= 0 && y < rows) return grid[(size_t)x][(size_t)y];
        return out_of_bounds;
    }
    // CSR form of the C ABI: cell (x, y) -> id x*rows + y, items in push_back order
    void toCSR(std::vector<int32_t>& cell_start, std::vector<int32_t>& cell_items) const
    {
        cell_start.assign((size_t)cols * rows + 1, 0);
        cell_items.clear();
        for (int x = 0; x < cols; ++x)
            for (int y = 0; y < rows; ++y) {
                for (int i : grid[(size_t)x][(size_t)y]) cell_items.push_back(i);
                cell_start[(size_t)x * rows + y + 1] = (int32_t)cell_items.size();
            }
    }
private:
    std::vector<std::vector<std::list<int>>> grid;
    std::list<int> out_of_bounds;
};

namespace detail {
template <class Mat1, class Mat2>
inline int matchGridCall(const std::vector<int32_t>& centres, int n_centres, const Mat1& desc1, const GridStructure& grid,
                         const Mat2& desc2, const double* dir1, const double* dir2, const GridWindow& w, float nnr,
                         std::vector<int>& matches_12, const char* who)
{
    matches_12.assign((size_t)desc1.rows, -1);
    std::vector<int32_t> cs, items;
    grid.toCSR(cs, items);
    const int32_t win[4] = {w.width.first, w.width.second, w.height.first, w.height.second};
    int32_t n = 0;
    check(plslam_match_grid(ctx(), centres.data(), n_centres, rows_of(desc1, "desc1"), desc1.rows, cs.data(),
                            items.data(), grid.cols, grid.rows, rows_of(desc2, "desc2"), desc2.rows, dir1, dir2,
                            lineSimTh(), win, (double)nnr, bestLRMatches() ? 1 : 0, matches_12.data(), &n),
          who);
    return n;
}
}  // namespace detail

// int matchGrid(points1, desc1, grid, desc2, w, matches_12): nnr is Config::minRatio12P() upstream (a global there;
// an argument with the same default use here: pass SlamConfig::minRatio12P())
template <class Mat1, class Mat2>
inline int matchGrid(const std::vector<point_2d>& points1, const Mat1& desc1, const GridStructure& grid,
                     const Mat2& desc2, const GridWindow& w, std::vector<int>& matches_12, float nnr = 0.75f)
{
    if ((int)points1.size() != desc1.rows)
        throw std::runtime_error("[matchGrid] Each point needs a corresponding descriptor!");
    std::vector<int32_t> centres(points1.size() * 2);
    for (size_t i = 0; i < points1.size(); ++i) {
        centres[2 * i] = points1[i].first;
        centres[2 * i + 1] = points1[i].second;
    }
    return detail::matchGridCall(centres, 1, desc1, grid, desc2, nullptr, nullptr, w, nnr, matches_12, "StVO::matchGrid");
}

// line overload: candidates from the windows around both end points, skipped when the directions disagree
template <class Mat1, class Mat2>
inline int matchGrid(const std::vector<line_2d>& lines1, const Mat1& desc1, const GridStructure& grid, const Mat2& desc2,
                     const std::vector<std::pair<double, double>>& directions2, const GridWindow& w,
                     std::vector<int>& matches_12, float nnr = 0.75f)
{
    if ((int)lines1.size() != desc1.rows)
        throw std::runtime_error("[matchGrid] Each line needs a corresponding descriptor!");
    if ((int)directions2.size() != desc2.rows)
        throw std::runtime_error("[matchGrid] Each candidate line needs a direction!");
    std::vector<int32_t> centres(lines1.size() * 4);
    std::vector<double> dir1(lines1.size() * 2), dir2(directions2.size() * 2);
    for (size_t i = 0; i < lines1.size(); ++i) {
        const point_2d &sp = lines1[i].first, &ep = lines1[i].second;
        centres[4 * i] = sp.first; centres[4 * i + 1] = sp.second;
        centres[4 * i + 2] = ep.first; centres[4 * i + 3] = ep.second;
        std::pair<double, double> v = std::make_pair((double)(ep.first - sp.first), (double)(ep.second - sp.second));
        normalize(v);                                  // a zero vector becomes NaN: the direction test then never skips
        dir1[2 * i] = v.first;
        dir1[2 * i + 1] = v.second;
    }
    for (size_t i = 0; i < directions2.size(); ++i) {
        dir2[2 * i] = directions2[i].first;
        dir2[2 * i + 1] = directions2[i].second;
    }
    return detail::matchGridCall(centres, 2, desc1, grid, desc2, dir1.data(), dir2.data(), w, nnr, matches_12,
                                 "StVO::matchGrid");
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
