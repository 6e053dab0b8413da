// Stand-in for stvo-pl's gridStructure.h -- TEST INFRASTRUCTURE (see opencv2/core.hpp next to it).
#pragma once
#include <list>
#include <utility>
#include <vector>
namespace StVO {
struct GridWindow {
    std::pair<int, int> width, height;
};
class GridStructure {
public:
    int rows, cols;
    GridStructure(int r, int c) : rows(r), cols(c), cells((size_t)r * c) {}
    std::list<int>& at(int x, int y) { return (x >= 0 && x < cols && y >= 0 && y < rows) ? cells[(size_t)x * rows + y] : dummy; }
    const std::list<int>& cell(int x, int y) const { return cells[(size_t)x * rows + y]; }
private:
    std::vector<std::list<int>> cells;
    std::list<int> dummy;
};
}  // namespace StVO
