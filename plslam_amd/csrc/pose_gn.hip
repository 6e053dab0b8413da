// K17 -- the iteration body of MapHandler::computeRelativePoseGN (src/mapHandler.cpp:3324-3424) and of
// computeRelativePoseRobustGN (:3588-3689; the same loops): the pose-only Gauss-Newton system of the loop-closure
// relative pose.  Per inlier point (:3334-3366):
//     P_ = T_inc * P;  err = proj(P_) - pl_obs;  r = |err|;  fgz2 = fx / max(th, gz^2)      (fx for BOTH coordinates)
//     J = fgz2 * [dx gz, dy gz, -(gx dx + gy dy), -(gx gy dx + gy^2 dy + gz^2 dy), gx^2 dx + gz^2 dx + gx gy dy,
//                 gx gz dy - gy gz dx] / max(th, r);     w = robustWeightCauchy(r)
// per inlier line (:3372-3423): err = (l . proj(sP_), l . proj(eP_)), Js / Je as above with (lx, ly) = l_obs(0..1) in place
// of (dx, dy) and each end point's own fgz2, J = (Js ds + Je de) / max(th, r);
//     H_p += J J^T w;  g_p += J r w;  e_p += r^2 w;  N_p++      (H_l, g_l, e_l, N_l likewise);  H = H_p + H_l, ...
// fp64, the source's operation order inside a row, no FMA contraction.  One workgroup: a lane takes the observations
// tid, tid + 256, ... and keeps its partial sums (21 + 6 + 1 per kind) in registers; the 256 partials are then summed
// in a fixed tree through LDS -- the result does not depend on scheduling and differs from the reference's sequential
// sum only by rounding (<= 1e-12 relative; contract 1e-6).  A loop closure has a few hundred observations: launch-bound.
#include <cstring>

#include "common.hpp"

namespace plslam {
namespace {

constexpr int GN_THREADS = 256;
constexpr int GN_TERMS = 21 + 6 + 1;       // upper triangle of H, g, e

struct GnCam { double fx, fy, cx, cy; };

__device__ __forceinline__ double dmaxd(double a, double b) { return a < b ? b : a; }   // std::max

__device__ __forceinline__ void xform(const double* T, const double* X, double o[3])
{
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = (T[4 * i] * X[0] + T[4 * i + 1] * X[1] + T[4 * i + 2] * X[2]) + T[4 * i + 3];
}

__device__ __forceinline__ void jac6(double fgz2, double a, double b, double gx, double gy, double gz, double J[6])
{
    J[0] = +fgz2 * a * gz;
    J[1] = +fgz2 * b * gz;
    J[2] = -fgz2 * (gx * a + gy * b);
    J[3] = -fgz2 * (gx * gy * a + gy * gy * b + gz * gz * b);
    J[4] = +fgz2 * (gx * gx * a + gz * gz * a + gx * gy * b);
    J[5] = +fgz2 * (gx * gz * b - gy * gz * a);
}

__device__ __forceinline__ void accumulate(double acc[GN_TERMS], const double J[6], double r, double w)
{
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) acc[k++] += J[i] * J[j] * w;     // (J J^T) w, evaluated as Eigen does: product, then * w
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] += J[i] * r * w;
    acc[27] += r * r * w;
}

__global__ void __launch_bounds__(GN_THREADS)
k_pose_gn(GnCam K, double th, const double* __restrict__ T, const double* __restrict__ P, const double* __restrict__ pl_obs,
          const uint8_t* __restrict__ pt_inlier, int32_t npt, const double* __restrict__ sPeP,
          const double* __restrict__ le_obs, const uint8_t* __restrict__ ls_inlier, int32_t nls, double* __restrict__ out)
{
    __shared__ double red[GN_THREADS];
    __shared__ int32_t cnt[2];
    const int tid = threadIdx.x;
    if (tid < 2) cnt[tid] = 0;
    double Tm[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) Tm[i] = T[i];
    double ap[GN_TERMS], al[GN_TERMS];
#pragma unroll
    for (int k = 0; k < GN_TERMS; ++k) ap[k] = al[k] = 0.0;
    int np = 0, nl = 0;
    for (int32_t i = tid; i < npt; i += GN_THREADS) {
        if (!pt_inlier[i]) continue;
        double G[3], J[6];
        xform(Tm, P + 3 * (size_t)i, G);
        const double u = K.cx + K.fx * G[0] / G[2], v = K.cy + K.fy * G[1] / G[2];
        const double dx = u - pl_obs[2 * (size_t)i], dy = v - pl_obs[2 * (size_t)i + 1];
        const double r = sqrt(dx * dx + dy * dy);
        const double fgz2 = K.fx / dmaxd(th, G[2] * G[2]);
        jac6(fgz2, dx, dy, G[0], G[1], G[2], J);
        const double den = dmaxd(th, r);
#pragma unroll
        for (int k = 0; k < 6; ++k) J[k] = J[k] / den;
        accumulate(ap, J, r, 1.0 / (1.0 + r * r));
        ++np;
    }
    for (int32_t i = tid; i < nls; i += GN_THREADS) {
        if (!ls_inlier[i]) continue;
        double S[3], E[3], Js[6], Je[6], J[6];
        xform(Tm, sPeP + 6 * (size_t)i, S);
        xform(Tm, sPeP + 6 * (size_t)i + 3, E);
        const double su = K.cx + K.fx * S[0] / S[2], sv = K.cy + K.fy * S[1] / S[2];
        const double eu = K.cx + K.fx * E[0] / E[2], ev = K.cy + K.fy * E[1] / E[2];
        const double lx = le_obs[3 * (size_t)i], ly = le_obs[3 * (size_t)i + 1], lz = le_obs[3 * (size_t)i + 2];
        const double ds = lx * su + ly * sv + lz, de = lx * eu + ly * ev + lz;
        const double r = sqrt(ds * ds + de * de);
        jac6(K.fx / dmaxd(th, S[2] * S[2]), lx, ly, S[0], S[1], S[2], Js);
        jac6(K.fx / dmaxd(th, E[2] * E[2]), lx, ly, E[0], E[1], E[2], Je);
        const double den = dmaxd(th, r);
#pragma unroll
        for (int k = 0; k < 6; ++k) J[k] = (Js[k] * ds + Je[k] * de) / den;
        accumulate(al, J, r, 1.0 / (1.0 + r * r));
        ++nl;
    }
    __syncthreads();
    if (np) atomicAdd(&cnt[0], np);
    if (nl) atomicAdd(&cnt[1], nl);
    // fixed-shape tree per term; term k of the points, then of the lines; out[k] = points + lines (H = H_p + H_l, :3417-3419)
    for (int k = 0; k < GN_TERMS; ++k) {
        double tot[2];
        for (int kind = 0; kind < 2; ++kind) {
            __syncthreads();
            red[tid] = kind ? al[k] : ap[k];
            __syncthreads();
            for (int st = GN_THREADS / 2; st > 0; st >>= 1) {
                if (tid < st) red[tid] = red[tid] + red[tid + st];
                __syncthreads();
            }
            tot[kind] = red[0];
        }
        if (tid == 0) out[k] = tot[0] + tot[1];
    }
    __syncthreads();
    if (tid == 0) {
        out[GN_TERMS] = (double)cnt[0];
        out[GN_TERMS + 1] = (double)cnt[1];
    }
}

}  // namespace
}  // namespace plslam

extern "C" {

int plslam_pose_gn_accumulate(plslam_ctx* ctx, const plslam_cam* K, double homog_th, const double* T_inc, const double* P,
                              const double* pl_obs, const uint8_t* pt_inlier, int32_t npt, const double* sPeP,
                              const double* le_obs, const uint8_t* ls_inlier, int32_t nls, double* H, double* g, double* e,
                              int32_t* n_obs)
{
    using namespace plslam;
    PLSLAM_REQUIRE(ctx && K && T_inc && H && g && e && npt >= 0 && nls >= 0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(npt == 0 || (P && pl_obs && pt_inlier), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(nls == 0 || (sPeP && le_obs && ls_inlier), PLSLAM_EINVAL);
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard(ctx->device);
    Carver ci;
    const size_t oT = ci.take(128), oP = ci.take((size_t)npt * 24), oO = ci.take((size_t)npt * 16), oI = ci.take((size_t)npt),
                 oS = ci.take((size_t)nls * 48), oL = ci.take((size_t)nls * 24), oJ = ci.take((size_t)nls);
    const size_t out_bytes = (GN_TERMS + 2) * 8;
    int rc;
    if ((rc = ctx->pin_in.reserve(ci.off))) return rc;
    if ((rc = ctx->in_a.reserve(ci.off))) return rc;
    if ((rc = ctx->pin_out.reserve(out_bytes))) return rc;
    if ((rc = ctx->out_a.reserve(out_bytes))) return rc;
    char* h = ctx->pin_in.as<char>();
    char* d = ctx->in_a.as<char>();
    memcpy(h + oT, T_inc, 128);
    if (npt) { memcpy(h + oP, P, (size_t)npt * 24); memcpy(h + oO, pl_obs, (size_t)npt * 16); memcpy(h + oI, pt_inlier, (size_t)npt); }
    if (nls) { memcpy(h + oS, sPeP, (size_t)nls * 48); memcpy(h + oL, le_obs, (size_t)nls * 24); memcpy(h + oJ, ls_inlier, (size_t)nls); }
    hipStream_t s = ctx->stream;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d, h, ci.off, hipMemcpyHostToDevice, s));
    const GnCam cam{K->fx, K->fy, K->cx, K->cy};
    hipLaunchKernelGGL(k_pose_gn, dim3(1), dim3(GN_THREADS), 0, s, cam, homog_th, (const double*)(d + oT), (const double*)(d + oP),
                       (const double*)(d + oO), (const uint8_t*)(d + oI), npt, (const double*)(d + oS), (const double*)(d + oL),
                       (const uint8_t*)(d + oJ), nls, ctx->out_a.as<double>());
    PLSLAM_HIP_CHECK(hipGetLastError());
    PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->pin_out.p, ctx->out_a.p, out_bytes, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    const double* o = ctx->pin_out.as<double>();
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j) {
            H[6 * i + j] = o[k];
            H[6 * j + i] = o[k];
            ++k;
        }
    for (int i = 0; i < 6; ++i) g[i] = o[21 + i];
    *e = o[27];
    if (n_obs) {
        n_obs[0] = (int32_t)o[GN_TERMS];
        n_obs[1] = (int32_t)o[GN_TERMS + 1];
    }
    return PLSLAM_OK;
}

}  // extern "C"
