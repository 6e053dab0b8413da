// K15/K16 -- the stereo L<->R gates of StVO::StereoFrame (stvo-pl stereoFrame.cpp: matchStereoPoints /
// matchStereoLines, [RECALL]) that turn a match table of (desc_l, desc_r) into stereo features:
//   points: |pt_l.y - pt_r.y| <= max_dist_epip (float arithmetic on cv::KeyPoint::pt) and disparity
//           pt_l.x - pt_r.x >= min_disp                              (config/config/config_kitti.yaml:25-26)
//   lines:  end-point disparities after sliding the right end points along the right line to the rows of the left
//           ones, their ratio against ls_min_disp_ratio (:36), both >= min_disp, neither line horizontal
//           (line_horiz_th, :34), vertical overlap above stereo_overlap_th (:31)
// One lane per left feature, fp64 with the source's operation order and no FMA contraction (the build uses
// -ffp-contract=off), so the thresholded decisions equal the CPU restatement bit for bit.  HBM-bound: 4 B (table) +
// 8/16 B (left feature) + one scattered 8/16 B read (right feature) in, 12/20 B out per feature; one frame (1500 +
// 200 features) is launch-bound.
#include <cmath>
#include <cstring>

#include "common.hpp"

namespace plslam {
namespace {

__global__ void __launch_bounds__(256)
k_stereo_point_gate(const int32_t* __restrict__ m12, int32_t n_l, const float2* __restrict__ kp_l,
                    const float2* __restrict__ kp_r, int32_t n_r, double max_dist_epip, double min_disp,
                    int32_t* __restrict__ stereo_12, double* __restrict__ disp, int32_t* __restrict__ count)
{
    const int i1 = blockIdx.x * 256 + threadIdx.x;
    int ok = 0;
    if (i1 < n_l) {
        const int32_t i2 = m12[i1];
        double dsp = 0.0;
        if (i2 >= 0 && i2 < n_r) {
            const float2 a = kp_l[i1], b = kp_r[i2];
            const float dy = __fsub_rn(a.y, b.y);
            if ((double)fabsf(dy) <= max_dist_epip) {
                const double d = (double)__fsub_rn(a.x, b.x);
                if (d >= min_disp) {
                    ok = 1;
                    dsp = d;
                }
            }
        }
        stereo_12[i1] = ok ? i2 : -1;
        disp[i1] = dsp;
    }
    const unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(count, (int)__popcll(bal));
}

__device__ __forceinline__ double dmin2(double a, double b) { return b < a ? b : a; }   // std::min
__device__ __forceinline__ double dmax2(double a, double b) { return a < b ? b : a; }   // std::max

// StereoFrame::lineSegmentOverlapStereo
__device__ __forceinline__ double overlap_stereo(double spl_obs, double epl_obs, double spl_proj, double epl_proj,
                                                 double line_horiz_th)
{
    double overlap = 1.f;
    if (fabs(epl_obs - spl_obs) > line_horiz_th) {
        const double sln = dmin2(spl_obs, epl_obs), eln = dmax2(spl_obs, epl_obs);
        const double spn = dmin2(spl_proj, epl_proj), epn = dmax2(spl_proj, epl_proj);
        const double length = eln - spn;
        if ((epn < sln) || (spn > eln))
            overlap = 0.f;
        else if ((epn > eln) && (spn < sln))
            overlap = eln - sln;
        else
            overlap = dmin2(eln, epn) - dmax2(sln, spn);
        if (length > 0.01f)
            overlap = overlap / length;
        else
            overlap = 0.f;
        if (overlap > 1.f) overlap = 1.f;
    }
    return overlap;
}

__global__ void __launch_bounds__(256)
k_stereo_line_gate(const int32_t* __restrict__ m12, int32_t n_l, const float4* __restrict__ seg_l,
                   const float4* __restrict__ seg_r, int32_t n_r, double min_disp, double line_horiz_th,
                   double stereo_overlap_th, double ls_min_disp_ratio, int32_t* __restrict__ stereo_12,
                   double* __restrict__ disp_se, int32_t* __restrict__ count)
{
    const int i1 = blockIdx.x * 256 + threadIdx.x;
    int ok = 0;
    if (i1 < n_l) {
        const int32_t i2 = m12[i1];
        double ds = 0.0, de = 0.0;
        if (i2 >= 0 && i2 < n_r) {
            const float4 L = seg_l[i1], R = seg_r[i2];
            const double sp_l[2] = {L.x, L.y}, ep_l[2] = {L.z, L.w};
            double sp_r[2] = {R.x, R.y}, ep_r[2] = {R.z, R.w};
            const double overlap = overlap_stereo(sp_l[1], ep_l[1], sp_r[1], ep_r[1], line_horiz_th);
            const double sx = (sp_r[0] * (sp_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - sp_l[1])) / (sp_r[1] - ep_r[1]);
            sp_r[0] = sx;
            sp_r[1] = sp_l[1];
            const double ex = (sp_r[0] * (ep_l[1] - ep_r[1]) + ep_r[0] * (sp_r[1] - ep_l[1])) / (sp_r[1] - ep_r[1]);
            ep_r[0] = ex;
            ep_r[1] = ep_l[1];
            double disp_s = sp_l[0] - sp_r[0], disp_e = ep_l[0] - ep_r[0];
            if (dmin2(disp_s, disp_e) / dmax2(disp_s, disp_e) < ls_min_disp_ratio) {
                disp_s = -1.0;
                disp_e = -1.0;
            }
            if (disp_s >= min_disp && disp_e >= min_disp && fabs(sp_l[1] - ep_l[1]) > line_horiz_th &&
                fabs(sp_r[1] - ep_r[1]) > line_horiz_th && overlap > stereo_overlap_th) {
                ok = 1;
                ds = disp_s;
                de = disp_e;
            }
        }
        stereo_12[i1] = ok ? i2 : -1;
        disp_se[2 * (size_t)i1] = ds;
        disp_se[2 * (size_t)i1 + 1] = de;
    }
    const unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(count, (int)__popcll(bal));
}

// one host-pointer call: lines != 0 -> segments (4 floats per feature) and two disparities per feature
int stereo_gate_host(plslam_ctx* ctx, int lines, const int32_t* m12, int32_t n_l, const float* f_l, const float* f_r,
                     int32_t n_r, double max_dist_epip, double min_disp, double line_horiz_th, double stereo_overlap_th,
                     double ls_min_disp_ratio, int32_t* stereo_12, double* disp, int32_t* n_stereo)
{
    PLSLAM_REQUIRE(ctx && n_l >= 0 && n_r >= 0, PLSLAM_EINVAL);
    if (n_stereo) *n_stereo = 0;
    if (n_l == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(m12 && f_l && stereo_12 && disp && (n_r == 0 || f_r), PLSLAM_EINVAL);
    const size_t fw = lines ? 16 : 8, dw = lines ? 16 : 8;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    Carver ci, co;
    const size_t oM = ci.take((size_t)n_l * 4), oL = ci.take((size_t)n_l * fw), oR = ci.take((size_t)n_r * fw + 16);
    const size_t oS = co.take((size_t)n_l * 4), oD = co.take((size_t)n_l * dw), oC = co.take(4);
    int rc;
    if ((rc = ctx->pin_in.reserve(ci.off))) return rc;
    if ((rc = ctx->in_a.reserve(ci.off))) return rc;
    if ((rc = ctx->pin_out.reserve(co.off))) return rc;
    if ((rc = ctx->out_a.reserve(co.off))) return rc;
    char* h = ctx->pin_in.as<char>();
    char* d = ctx->in_a.as<char>();
    char* o = ctx->out_a.as<char>();
    memcpy(h + oM, m12, (size_t)n_l * 4);
    memcpy(h + oL, f_l, (size_t)n_l * fw);
    if (n_r) memcpy(h + oR, f_r, (size_t)n_r * fw);
    hipStream_t s = ctx->stream;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d, h, ci.off, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemsetAsync(o + oC, 0, 4, s));
    const dim3 grid((unsigned)((n_l + 255) / 256)), block(256);
    if (!lines)
        hipLaunchKernelGGL(k_stereo_point_gate, grid, block, 0, s, (const int32_t*)(d + oM), n_l, (const float2*)(d + oL),
                           (const float2*)(d + oR), n_r, max_dist_epip, min_disp, (int32_t*)(o + oS), (double*)(o + oD),
                           (int32_t*)(o + oC));
    else
        hipLaunchKernelGGL(k_stereo_line_gate, grid, block, 0, s, (const int32_t*)(d + oM), n_l, (const float4*)(d + oL),
                           (const float4*)(d + oR), n_r, min_disp, line_horiz_th, stereo_overlap_th, ls_min_disp_ratio,
                           (int32_t*)(o + oS), (double*)(o + oD), (int32_t*)(o + oC));
    PLSLAM_HIP_CHECK(hipGetLastError());
    PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->pin_out.p, o, co.off, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    const char* r = ctx->pin_out.as<char>();
    memcpy(stereo_12, r + oS, (size_t)n_l * 4);
    memcpy(disp, r + oD, (size_t)n_l * dw);
    if (n_stereo) memcpy(n_stereo, r + oC, 4);
    return PLSLAM_OK;
}

}  // namespace
}  // namespace plslam

extern "C" {

int plslam_stereo_point_gate(plslam_ctx* ctx, const int32_t* matches_12, int32_t n_l, const float* kp_l,
                             const float* kp_r, int32_t n_r, double max_dist_epip, double min_disp,
                             int32_t* stereo_12, double* disp, int32_t* n_stereo)
{
    return plslam::stereo_gate_host(ctx, 0, matches_12, n_l, kp_l, kp_r, n_r, max_dist_epip, min_disp, 0.0, 0.0, 0.0,
                                    stereo_12, disp, n_stereo);
}

int plslam_stereo_line_gate(plslam_ctx* ctx, const int32_t* matches_12, int32_t n_l, const float* seg_l,
                            const float* seg_r, int32_t n_r, double min_disp, double line_horiz_th,
                            double stereo_overlap_th, double ls_min_disp_ratio, int32_t* stereo_12, double* disp_se,
                            int32_t* n_stereo)
{
    return plslam::stereo_gate_host(ctx, 1, matches_12, n_l, seg_l, seg_r, n_r, 0.0, min_disp, line_horiz_th,
                                    stereo_overlap_th, ls_min_disp_ratio, stereo_12, disp_se, n_stereo);
}

}  // extern "C"
