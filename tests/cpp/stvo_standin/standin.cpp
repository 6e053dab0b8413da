// TEST INFRASTRUCTURE: the stand-in StVO:: functions of matching.h, forwarding to the CPU restatement (oracle/).
#include <cmath>

#include "config.h"
#include "matching.h"
#include "plslam_oracle.h"

namespace StVO {
namespace {
void to_csr(const GridStructure& g, std::vector<int32_t>& cs, std::vector<int32_t>& items)
{
    cs.assign((size_t)g.cols * g.rows + 1, 0);
    items.clear();
    for (int x = 0; x < g.cols; ++x)
        for (int y = 0; y < g.rows; ++y) {
            for (int v : g.cell(x, y)) items.push_back(v);
            cs[(size_t)x * g.rows + y + 1] = (int32_t)items.size();
        }
    items.push_back(0);
}
}  // namespace

int match(const cv::Mat& desc1, const cv::Mat& desc2, float nnr, std::vector<int>& matches_12)
{
    matches_12.assign((size_t)desc1.rows, -1);
    return plo_match(desc1.data, desc1.rows, desc2.data, desc2.rows, nnr, Config::bestLRMatches() ? 1 : 0, matches_12.data());
}

int matchGrid(const std::vector<point_2d>& points1, const cv::Mat& desc1, const GridStructure& grid, const cv::Mat& desc2,
              const GridWindow& w, std::vector<int>& matches_12)
{
    std::vector<int32_t> cs, items, cen;
    to_csr(grid, cs, items);
    for (const point_2d& p : points1) { cen.push_back(p.first); cen.push_back(p.second); }
    const int32_t win[4] = {w.width.first, w.width.second, w.height.first, w.height.second};
    matches_12.assign(points1.size(), -1);
    return plo_match_grid(cen.data(), 1, desc1.data, desc1.rows, cs.data(), items.data(), grid.cols, grid.rows, desc2.data,
                          desc2.rows, nullptr, nullptr, 0.0, win, Config::minRatio12P(), Config::bestLRMatches() ? 1 : 0,
                          matches_12.data());
}

int matchGrid(const std::vector<line_2d>& lines1, const cv::Mat& desc1, const GridStructure& grid, const cv::Mat& desc2,
              const std::vector<std::pair<double, double>>& directions2, const GridWindow& w, std::vector<int>& matches_12)
{
    std::vector<int32_t> cs, items, cen;
    std::vector<double> d1, d2;
    to_csr(grid, cs, items);
    for (const line_2d& l : lines1) {
        cen.push_back(l.first.first); cen.push_back(l.first.second); cen.push_back(l.second.first); cen.push_back(l.second.second);
        // the query direction from the INTEGER end points, normalised (zero vectors -> NaN), as upstream derives it
        const double vx = l.second.first - l.first.first, vy = l.second.second - l.first.second, m = std::sqrt(vx * vx + vy * vy);
        d1.push_back(vx / m); d1.push_back(vy / m);
    }
    for (const auto& d : directions2) { d2.push_back(d.first); d2.push_back(d.second); }
    const int32_t win[4] = {w.width.first, w.width.second, w.height.first, w.height.second};
    matches_12.assign(lines1.size(), -1);
    return plo_match_grid(cen.data(), 2, desc1.data, desc1.rows, cs.data(), items.data(), grid.cols, grid.rows, desc2.data,
                          desc2.rows, d1.data(), d2.data(), Config::lineSimTh(), win, Config::minRatio12P(),
                          Config::bestLRMatches() ? 1 : 0, matches_12.data());
}
}  // namespace StVO
