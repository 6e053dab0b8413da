// Stand-in for stvo-pl's matching.h -- TEST INFRASTRUCTURE (see opencv2/core.hpp next to it): the same signatures, implemented
// in standin.cpp by forwarding to the CPU restatement (oracle/), so that the harness can be exercised end to end here.
#pragma once
#include <utility>
#include <vector>

#include <opencv2/core.hpp>

#include "gridStructure.h"
namespace StVO {
typedef std::pair<int, int> point_2d;
typedef std::pair<point_2d, point_2d> line_2d;
int match(const cv::Mat& desc1, const cv::Mat& desc2, float nnr, std::vector<int>& matches_12);
int matchGrid(const std::vector<point_2d>& points1, const cv::Mat& desc1, const GridStructure& grid, const cv::Mat& desc2,
              const GridWindow& w, std::vector<int>& matches_12);
int matchGrid(const std::vector<line_2d>& lines1, const cv::Mat& desc1, const GridStructure& grid, const cv::Mat& desc2,
              const std::vector<std::pair<double, double>>& directions2, const GridWindow& w, std::vector<int>& matches_12);
}  // namespace StVO
