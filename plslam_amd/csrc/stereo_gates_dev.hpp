// stereo_gates_dev.hpp -- the per-feature gates of StVO::StereoFrame (matchStereoPoints / matchStereoLines, [RECALL])
// as device functions: used by the stand-alone gate kernels (stereo_gates.hip) and by the finalize kernel (hamming.hip), which
// applies them to a row's match the moment it is decided (plans with a gate stage: no second pass over the tables).
#pragma once
#include "common.hpp"

namespace plslam {

typedef float gfvec2_t __attribute__((ext_vector_type(2)));
typedef float gfvec4_t __attribute__((ext_vector_type(4)));

// one left key point: the gate of matchStereoPoints; returns the kept right index or -1, *dsp = its disparity or 0
__device__ __forceinline__ int32_t point_gate_one(int32_t i2, float2 a, const PLSLAM_AS1 float2* __restrict__ kp_r, int32_t n_r,
                                                  double max_dist_epip, double min_disp, double* dsp)
{
    *dsp = 0.0;
    if (i2 < 0 || i2 >= n_r) return -1;
    const gfvec2_t bv = reinterpret_cast<const PLSLAM_AS1 gfvec2_t*>(kp_r)[i2];
    const float2 b = make_float2(bv.x, bv.y);
    const float dy = __fsub_rn(a.y, b.y);
    if (!((double)fabsf(dy) <= max_dist_epip)) return -1;
    const double d = (double)__fsub_rn(a.x, b.x);
    if (!(d >= min_disp)) return -1;
    *dsp = d;
    return i2;
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

// one left segment: the gate of matchStereoLines (the second end point reads the already overwritten first one, as the
// source does); returns the kept right index or -1, ds / de = the end-point disparities or 0
__device__ __forceinline__ int32_t line_gate_one(int32_t i2, float4 L, const PLSLAM_AS1 float4* __restrict__ seg_r, int32_t n_r,
                                                 double min_disp, double line_horiz_th, double stereo_overlap_th,
                                                 double ls_min_disp_ratio, double* ds, double* de)
{
    *ds = 0.0;
    *de = 0.0;
    if (i2 < 0 || i2 >= n_r) return -1;
    const gfvec4_t R = reinterpret_cast<const PLSLAM_AS1 gfvec4_t*>(seg_r)[i2];
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
        *ds = disp_s;
        *de = disp_e;
        return i2;
    }
    return -1;
}


// one left feature of a gate problem whose match is i2: writes stereo_12 / disp (streaming stores), returns 1 if it is kept
__device__ __forceinline__ int stereo_gate_row(const plslam_stereo_gate_problem& q, int i1, int32_t i2)
{
    if (q.lines) {
        double ds, de;
        const gfvec4_t a = g_(reinterpret_cast<const gfvec4_t*>(q.f_l))[i1];
        const int32_t k = line_gate_one(i2, make_float4(a.x, a.y, a.z, a.w), g_(reinterpret_cast<const float4*>(q.f_r)), q.n_r,
                                        q.min_disp, q.line_horiz_th, q.stereo_overlap_th, q.ls_min_disp_ratio, &ds, &de);
        __builtin_nontemporal_store(k, g_(q.stereo_12) + i1);
        __builtin_nontemporal_store(ds, g_(q.disp) + 2 * (size_t)i1);
        __builtin_nontemporal_store(de, g_(q.disp) + 2 * (size_t)i1 + 1);
        return k >= 0;
    }
    double dsp;
    const gfvec2_t a = g_(reinterpret_cast<const gfvec2_t*>(q.f_l))[i1];
    const int32_t k = point_gate_one(i2, make_float2(a.x, a.y), g_(reinterpret_cast<const float2*>(q.f_r)), q.n_r, q.max_dist_epip,
                                     q.min_disp, &dsp);
    __builtin_nontemporal_store(k, g_(q.stereo_12) + i1);
    __builtin_nontemporal_store(dsp, g_(q.disp) + i1);
    return k >= 0;
}

}  // namespace plslam
