// Stand-in for stvo-pl's config.h -- TEST INFRASTRUCTURE (see opencv2/core.hpp next to it).
#pragma once
namespace StVO {
struct Config {
    static bool& bestLRMatches() { static bool v = true; return v; }
    static double& minRatio12P() { static double v = 0.75; return v; }
    static double& lineSimTh() { static double v = 0.75; return v; }
};
}  // namespace StVO
