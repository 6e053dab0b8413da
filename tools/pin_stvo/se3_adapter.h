// The stvo-pl helpers of the LBA rows / drivers behind plain arrays, for tools/pin_stvo/pin_stvo.cpp.
//   default build: stvo-pl's own auxiliar.h (inverse_se3, expmap_se3, logmap_se3, robustWeightCauchy: free functions, called
//   unqualified by src/mapHandler.cpp:137-183, :1372, :1407 of pl-slam) and pinholeStereoCamera.h (projection, :255) -- the
//   signatures are recalled; adjust HERE if a checkout differs;
//   -DPIN_STANDIN_SE3: the CPU restatement (oracle/), which is how tests/test_pin_tool_cpu.py exercises the harness itself.
// Matrices are row-major 4 x 4, twists are [t, w].
#pragma once

#ifdef PIN_STANDIN_SE3
#include <cmath>

#include "plslam_oracle.h"
namespace pin {
inline void inverse_se3(const double* T, double* o) { plo_inverse_se3(T, o); }
inline void expmap_se3(const double* x, double* o) { plo_expmap_se3(x, o); }
inline void logmap_se3(const double* T, double* o) { plo_logmap_se3(T, o); }
inline void projection(const double* cam /* fx fy cx cy */, const double* P, double* uv)
{
    uv[0] = cam[2] + cam[0] * P[0] / P[2];          // oracle/plslam_oracle.c:385-390 (static there)
    uv[1] = cam[3] + cam[1] * P[1] / P[2];
}
inline double cauchy(double r) { return 1.0 / (1.0 + r * r); }
}  // namespace pin
#else
#include <Eigen/Core>

#include "auxiliar.h"
#include "pinholeStereoCamera.h"
namespace pin {
typedef Eigen::Matrix<double, 4, 4, Eigen::RowMajor> M4r;
typedef Eigen::Matrix<double, 6, 1> V6;
inline void inverse_se3(const double* T, double* o) { Eigen::Map<M4r>(o) = ::inverse_se3(Eigen::Matrix4d(Eigen::Map<const M4r>(T))); }
inline void expmap_se3(const double* x, double* o) { Eigen::Map<M4r>(o) = ::expmap_se3(V6(Eigen::Map<const V6>(x))); }
inline void logmap_se3(const double* T, double* o) { Eigen::Map<V6>(o) = ::logmap_se3(Eigen::Matrix4d(Eigen::Map<const M4r>(T))); }
inline void projection(const double* cam, const double* P, double* uv)
{
    StVO::PinholeStereoCamera c(752, 480, cam[0], cam[1], cam[2], cam[3], 0.11);    // (width, height, fx, fy, cx, cy, baseline)
    const Eigen::Vector2d r = c.projection(Eigen::Vector3d(P[0], P[1], P[2]));
    uv[0] = r(0);
    uv[1] = r(1);
}
inline double cauchy(double r) { return ::robustWeightCauchy(r); }
}  // namespace pin
#endif
