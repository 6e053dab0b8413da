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
#include "stereo_gates_dev.hpp"

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
        double dsp;
        const gfvec2_t a = g_(reinterpret_cast<const gfvec2_t*>(kp_l))[i1];
        const int32_t k = point_gate_one(g_(m12)[i1], make_float2(a.x, a.y), g_(kp_r), n_r, max_dist_epip, min_disp, &dsp);
        ok = k >= 0;
        g_(stereo_12)[i1] = k;
        g_(disp)[i1] = dsp;
    }
    const unsigned long long bal = __ballot(ok);
    if (count && (threadIdx.x & 63) == 0 && bal) (void)atomic_add_global(count, (int)__popcll(bal));
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
        double ds, de;
        const gfvec4_t a = g_(reinterpret_cast<const gfvec4_t*>(seg_l))[i1];
        const int32_t k = line_gate_one(g_(m12)[i1], make_float4(a.x, a.y, a.z, a.w), g_(seg_r), n_r, min_disp, line_horiz_th,
                                        stereo_overlap_th, ls_min_disp_ratio, &ds, &de);
        ok = k >= 0;
        g_(stereo_12)[i1] = k;
        g_(disp_se)[2 * (size_t)i1] = ds;
        g_(disp_se)[2 * (size_t)i1 + 1] = de;
    }
    const unsigned long long bal = __ballot(ok);
    if (count && (threadIdx.x & 63) == 0 && bal) (void)atomic_add_global(count, (int)__popcll(bal));
}

// the gate stage of a match plan: every (gate problem, 256 left features) pair is one workgroup of ONE launch
__global__ void __launch_bounds__(256)
k_stereo_gates_batched(const plslam_stereo_gate_problem* __restrict__ gates, const BlockDesc* __restrict__ blocks)
{
    const BlockDesc bd = blocks[blockIdx.x];
    const plslam_stereo_gate_problem q = gates[bd.item];
    const int i1 = bd.row0 + (int)threadIdx.x;
    int ok = 0;
    if (i1 < q.n_l) {
        const int32_t i2 = __builtin_nontemporal_load(g_(q.matches_12) + i1);   // (tables and outputs stream through once)
        ok = stereo_gate_row(q, i1, i2);
    }
    const unsigned long long bal = __ballot(ok);
    if (q.n_stereo && (threadIdx.x & 63) == 0 && bal) (void)atomic_add_global(q.n_stereo, (int)__popcll(bal));
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

int launch_stereo_gates(const plslam_stereo_gate_problem* d_gates, const BlockDesc* d_blocks, int nblocks, hipStream_t s)
{
    if (nblocks <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_stereo_gates_batched, dim3(nblocks), dim3(256), 0, s, d_gates, d_blocks);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int check_stereo_gate_problem(const plslam_stereo_gate_problem& q)
{
    PLSLAM_REQUIRE(q.n_l >= 0 && q.n_r >= 0, PLSLAM_EINVAL);
    if (q.n_l == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(q.matches_12 && q.f_l && q.stereo_12 && q.disp && (q.n_r == 0 || q.f_r), PLSLAM_EINVAL);
    const uintptr_t al = q.lines ? 15 : 7;            // float4 / float2 rows
    PLSLAM_REQUIRE((reinterpret_cast<uintptr_t>(q.f_l) & al) == 0 && (reinterpret_cast<uintptr_t>(q.f_r) & al) == 0,
                   PLSLAM_EINVAL);
    PLSLAM_REQUIRE((reinterpret_cast<uintptr_t>(q.disp) & 7) == 0, PLSLAM_EINVAL);
    return PLSLAM_OK;
}

// one gate problem with DEVICE pointers on stream `s` (no synchronisation); *n_stereo is zeroed first when given
static int stereo_gate_dev(plslam_ctx* ctx, const plslam_stereo_gate_problem& q, hipStream_t s)
{
    PLSLAM_REQUIRE(ctx != nullptr, PLSLAM_EINVAL);
    int rc = check_stereo_gate_problem(q);
    if (rc) return rc;
    DeviceGuard g(ctx->device);
    if (!s) s = ctx->stream;
    if (q.n_stereo) PLSLAM_HIP_CHECK(hipMemsetAsync(q.n_stereo, 0, 4, s));
    if (q.n_l == 0) return PLSLAM_OK;
    const dim3 grid((unsigned)((q.n_l + 255) / 256)), block(256);
    if (!q.lines)
        hipLaunchKernelGGL(k_stereo_point_gate, grid, block, 0, s, q.matches_12, q.n_l, (const float2*)q.f_l,
                           (const float2*)q.f_r, q.n_r, q.max_dist_epip, q.min_disp, q.stereo_12, q.disp, q.n_stereo);
    else
        hipLaunchKernelGGL(k_stereo_line_gate, grid, block, 0, s, q.matches_12, q.n_l, (const float4*)q.f_l,
                           (const float4*)q.f_r, q.n_r, q.min_disp, q.line_horiz_th, q.stereo_overlap_th,
                           q.ls_min_disp_ratio, q.stereo_12, q.disp, q.n_stereo);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

}  // namespace plslam

extern "C" {

int plslam_stereo_point_gate_dev(plslam_ctx* ctx, const int32_t* matches_12, int32_t n_l, const float* kp_l,
                                 const float* kp_r, int32_t n_r, double max_dist_epip, double min_disp,
                                 int32_t* stereo_12, double* disp, int32_t* n_stereo, void* stream)
{
    plslam_stereo_gate_problem q{};
    q.matches_12 = matches_12; q.f_l = kp_l; q.f_r = kp_r; q.n_l = n_l; q.n_r = n_r; q.lines = 0;
    q.max_dist_epip = max_dist_epip; q.min_disp = min_disp;
    q.stereo_12 = stereo_12; q.disp = disp; q.n_stereo = n_stereo;
    return plslam::stereo_gate_dev(ctx, q, static_cast<hipStream_t>(stream));
}

int plslam_stereo_line_gate_dev(plslam_ctx* ctx, const int32_t* matches_12, int32_t n_l, const float* seg_l,
                                const float* seg_r, int32_t n_r, double min_disp, double line_horiz_th,
                                double stereo_overlap_th, double ls_min_disp_ratio, int32_t* stereo_12,
                                double* disp_se, int32_t* n_stereo, void* stream)
{
    plslam_stereo_gate_problem q{};
    q.matches_12 = matches_12; q.f_l = seg_l; q.f_r = seg_r; q.n_l = n_l; q.n_r = n_r; q.lines = 1;
    q.min_disp = min_disp; q.line_horiz_th = line_horiz_th; q.stereo_overlap_th = stereo_overlap_th;
    q.ls_min_disp_ratio = ls_min_disp_ratio;
    q.stereo_12 = stereo_12; q.disp = disp_se; q.n_stereo = n_stereo;
    return plslam::stereo_gate_dev(ctx, q, static_cast<hipStream_t>(stream));
}

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
