// lba_assemble.hip -- device-side assembly of the local-BA normal equations in BLOCK form.
//
// The reference adds every observation's blocks into a dense MatrixXd H(N,N) and VectorXd g
// (src/mapHandler.cpp:1410-1429 for points, :1519-1538 for lines) and then takes H.sparseView()
// (:1555).  H has a fixed block structure -- one 6x6 block per optimised keyframe, one 3x3 / 6x6
// block per landmark, one 3x6 / 6x6 cross block per observation -- so this file produces exactly
// those blocks (the Schur-complement-ready layout) and g, with the reference's accumulation ORDER:
// each entry is the sequential sum over the observations in list order, so the blocks are bit-exact
// against the dense accumulation.  No atomics (they would make the sums order-dependent).
//   K7  k_landmark_blocks<DL>   one lane per landmark: H_ll (DLxDL), g_l (DL) over its observations
//   K8  k_cross_blocks<DL>      one lane per observation: W = J_lm * J_pose^T * w (DLx6)
//   K9  k_pose_blocks           one lane per (keyframe, entry): H_pp (6x6) and g_p (6) over the
//                                keyframe's observations (points first, then lines, list order)
//   K10 k_weighted_error        err = sum r^2 w (fixed-shape tree: deterministic, not sequential)
#include <vector>

#include "common.hpp"

namespace plslam {

template <int DL>
__global__ void __launch_bounds__(256)
k_landmark_blocks(const int32_t* __restrict__ lm_ptr, const int32_t* __restrict__ lm_obs, int32_t nlm,
                  const double* __restrict__ Jl, const double* __restrict__ r, const double* __restrict__ w,
                  double* __restrict__ Hll, double* __restrict__ gl)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= nlm) return;
    double H[DL * DL], g[DL];
#pragma unroll
    for (int i = 0; i < DL * DL; ++i) H[i] = 0.0;
#pragma unroll
    for (int i = 0; i < DL; ++i) g[i] = 0.0;
    for (int k = lm_ptr[l]; k < lm_ptr[l + 1]; ++k) {
        const int o = lm_obs[k];
        double J[DL];
#pragma unroll
        for (int a = 0; a < DL; ++a) J[a] = Jl[(size_t)o * DL + a];
        const double rr = r[o], ww = w[o];
#pragma unroll
        for (int a = 0; a < DL; ++a) g[a] += J[a] * rr * ww;
#pragma unroll
        for (int a = 0; a < DL; ++a)
#pragma unroll
            for (int b = 0; b < DL; ++b) H[a * DL + b] += J[a] * J[b] * ww;
    }
#pragma unroll
    for (int i = 0; i < DL * DL; ++i) Hll[(size_t)l * DL * DL + i] = H[i];
#pragma unroll
    for (int i = 0; i < DL; ++i) gl[(size_t)l * DL + i] = g[i];
}

template <int DL>
__global__ void __launch_bounds__(256)
k_cross_blocks(const int32_t* __restrict__ kf_loc, int32_t nobs, const double* __restrict__ Jp,
               const double* __restrict__ Jl, const double* __restrict__ w, double* __restrict__ W)
{
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= nobs) return;
    const bool opt = kf_loc[o] >= 0;   // kf_loc == -1: the keyframe is not optimised, no cross block
    double P[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) P[b] = Jp[(size_t)o * 6 + b];
    const double ww = w[o];
#pragma unroll
    for (int a = 0; a < DL; ++a) {
        const double ja = Jl[(size_t)o * DL + a];
#pragma unroll
        for (int b = 0; b < 6; ++b) W[((size_t)o * DL + a) * 6 + b] = opt ? ja * P[b] * ww : 0.0;
    }
}

// entry e of keyframe k: e < 36 -> H_pp[k][e/6][e%6]; e >= 36 -> g_p[k][e-36]
__global__ void __launch_bounds__(64)
k_pose_blocks(const int32_t* __restrict__ kf_ptr, const int32_t* __restrict__ kf_obs, int32_t n_pt_obs,
              const double* __restrict__ Jp_pt, const double* __restrict__ r_pt, const double* __restrict__ w_pt,
              const double* __restrict__ Jp_ls, const double* __restrict__ r_ls, const double* __restrict__ w_ls,
              double* __restrict__ Hpp, double* __restrict__ gp)
{
    const int k = blockIdx.x, e = threadIdx.x;
    if (e >= 42) return;
    const int a = e < 36 ? e / 6 : e - 36, b = e < 36 ? e % 6 : 0;
    double acc = 0.0;
    for (int i = kf_ptr[k]; i < kf_ptr[k + 1]; ++i) {
        const int o = kf_obs[i];       // global observation id: points [0, n_pt_obs), then lines
        const bool pt = o < n_pt_obs;
        const int oo = pt ? o : o - n_pt_obs;
        const double* J = (pt ? Jp_pt : Jp_ls) + (size_t)oo * 6;
        const double ww = (pt ? w_pt : w_ls)[oo];
        if (e < 36) acc += J[a] * J[b] * ww;
        else acc += J[a] * (pt ? r_pt : r_ls)[oo] * ww;
    }
    if (e < 36) Hpp[(size_t)k * 36 + e] = acc;
    else gp[(size_t)k * 6 + a] = acc;
}

__global__ void __launch_bounds__(256)
k_weighted_error(const double* __restrict__ r_pt, const double* __restrict__ w_pt, int32_t n_pt,
                 const double* __restrict__ r_ls, const double* __restrict__ w_ls, int32_t n_ls,
                 double* __restrict__ err)
{
    __shared__ double red[256];
    double acc = 0.0;
    for (int o = threadIdx.x; o < n_pt; o += 256) acc += r_pt[o] * r_pt[o] * w_pt[o];
    for (int o = threadIdx.x; o < n_ls; o += 256) acc += r_ls[o] * r_ls[o] * w_ls[o];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *err = red[0];
}

namespace {
struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~size_t(255); return o; }
};
}  // namespace
}  // namespace plslam

using namespace plslam;

extern "C" int plslam_lba_assemble(plslam_ctx* ctx, int32_t nkf, int32_t npt, int32_t nls,
                                   const int32_t* pt_lm_loc, const int32_t* pt_kf_loc, int32_t n_pt_obs,
                                   const double* pt_J_pose, const double* pt_J_lm, const double* pt_r,
                                   const double* pt_w, const int32_t* ls_lm_loc, const int32_t* ls_kf_loc,
                                   int32_t n_ls_obs, const double* ls_J_pose, const double* ls_J_lm,
                                   const double* ls_r, const double* ls_w, double* g, double* H_pose,
                                   double* H_pt, double* H_ls, double* W_pt, double* W_ls, double* err)
{
    PLSLAM_REQUIRE(ctx && nkf >= 0 && npt >= 0 && nls >= 0 && n_pt_obs >= 0 && n_ls_obs >= 0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(n_pt_obs == 0 || (pt_lm_loc && pt_kf_loc && pt_J_pose && pt_J_lm && pt_r && pt_w && W_pt), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(n_ls_obs == 0 || (ls_lm_loc && ls_kf_loc && ls_J_pose && ls_J_lm && ls_r && ls_w && W_ls), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(g && err && (nkf == 0 || H_pose) && (npt == 0 || H_pt) && (nls == 0 || H_ls), PLSLAM_EINVAL);
    for (int32_t o = 0; o < n_pt_obs; ++o)
        PLSLAM_REQUIRE(pt_lm_loc[o] >= 0 && pt_lm_loc[o] < npt && pt_kf_loc[o] >= -1 && pt_kf_loc[o] < nkf, PLSLAM_EINVAL);
    for (int32_t o = 0; o < n_ls_obs; ++o)
        PLSLAM_REQUIRE(ls_lm_loc[o] >= 0 && ls_lm_loc[o] < nls && ls_kf_loc[o] >= -1 && ls_kf_loc[o] < nkf, PLSLAM_EINVAL);

    // ---- stable CSR lists (host, O(nobs)): observations per landmark and per keyframe, list order
    auto csr = [](const int32_t* key, int32_t n, int32_t nkeys, int32_t id0, std::vector<int32_t>& ptr,
                  std::vector<int32_t>& ids, bool append) {
        if (!append) ptr.assign((size_t)nkeys + 1, 0);
        std::vector<int32_t> cnt((size_t)nkeys, 0);
        for (int32_t o = 0; o < n; ++o) if (key[o] >= 0) ++cnt[key[o]];
        if (!append) {
            for (int32_t k = 0; k < nkeys; ++k) ptr[k + 1] = ptr[k] + cnt[k];
            ids.assign((size_t)ptr[nkeys], 0);
            std::vector<int32_t> pos(ptr.begin(), ptr.end() - 1);
            for (int32_t o = 0; o < n; ++o) if (key[o] >= 0) ids[pos[key[o]]++] = id0 + o;
        }
    };
    std::vector<int32_t> ptp, pti, lsp, lsi;
    csr(pt_lm_loc, n_pt_obs, npt, 0, ptp, pti, false);
    csr(ls_lm_loc, n_ls_obs, nls, 0, lsp, lsi, false);
    // keyframes: points first, then lines (the reference runs the point loop before the line loop)
    std::vector<int32_t> kfp((size_t)nkf + 1, 0), kfi;
    {
        std::vector<int32_t> cnt((size_t)nkf, 0);
        for (int32_t o = 0; o < n_pt_obs; ++o) if (pt_kf_loc[o] >= 0) ++cnt[pt_kf_loc[o]];
        for (int32_t o = 0; o < n_ls_obs; ++o) if (ls_kf_loc[o] >= 0) ++cnt[ls_kf_loc[o]];
        for (int32_t k = 0; k < nkf; ++k) kfp[k + 1] = kfp[k] + cnt[k];
        kfi.assign((size_t)kfp[nkf], 0);
        std::vector<int32_t> pos(kfp.begin(), kfp.end() - 1);
        for (int32_t o = 0; o < n_pt_obs; ++o) if (pt_kf_loc[o] >= 0) kfi[pos[pt_kf_loc[o]]++] = o;
        for (int32_t o = 0; o < n_ls_obs; ++o) if (ls_kf_loc[o] >= 0) kfi[pos[ls_kf_loc[o]]++] = n_pt_obs + o;
    }

    std::lock_guard<std::mutex> lk(ctx->mu);
    hipStream_t s = ctx->stream;
    Carve c;
    const size_t np = (size_t)n_pt_obs, nl = (size_t)n_ls_obs;
    const size_t oPJp = c.take(np * 48), oPJl = c.take(np * 24), oPr = c.take(np * 8), oPw = c.take(np * 8),
                 oPk = c.take(np * 4), oLJp = c.take(nl * 48), oLJl = c.take(nl * 48), oLr = c.take(nl * 8),
                 oLw = c.take(nl * 8), oLk = c.take(nl * 4), oPtp = c.take(ptp.size() * 4), oPti = c.take(pti.size() * 4),
                 oLsp = c.take(lsp.size() * 4), oLsi = c.take(lsi.size() * 4), oKfp = c.take(kfp.size() * 4),
                 oKfi = c.take(kfi.size() * 4 + 4);
    const size_t N = 6 * (size_t)nkf + 3 * (size_t)npt + 6 * (size_t)nls;
    Carve co;
    const size_t oG = co.take(N * 8 + 8), oHp = co.take((size_t)nkf * 288 + 8), oHpt = co.take((size_t)npt * 72 + 8),
                 oHls = co.take((size_t)nls * 288 + 8), oWp = co.take(np * 144 + 8), oWl = co.take(nl * 288 + 8),
                 oErr = co.take(8);
    int rc;
    if ((rc = ctx->in_a.reserve(c.off + 256))) return rc;
    if ((rc = ctx->out_a.reserve(co.off + 256))) return rc;
    char* di = ctx->in_a.as<char>();
    char* dout = ctx->out_a.as<char>();
    auto up = [&](size_t off, const void* src, size_t bytes) -> int {
        if (bytes) PLSLAM_HIP_CHECK(hipMemcpyAsync(di + off, src, bytes, hipMemcpyHostToDevice, s));
        return PLSLAM_OK;
    };
    if ((rc = up(oPJp, pt_J_pose, np * 48)) || (rc = up(oPJl, pt_J_lm, np * 24)) || (rc = up(oPr, pt_r, np * 8)) ||
        (rc = up(oPw, pt_w, np * 8)) || (rc = up(oPk, pt_kf_loc, np * 4)) || (rc = up(oLJp, ls_J_pose, nl * 48)) ||
        (rc = up(oLJl, ls_J_lm, nl * 48)) || (rc = up(oLr, ls_r, nl * 8)) || (rc = up(oLw, ls_w, nl * 8)) ||
        (rc = up(oLk, ls_kf_loc, nl * 4)) || (rc = up(oPtp, ptp.data(), ptp.size() * 4)) ||
        (rc = up(oPti, pti.data(), pti.size() * 4)) || (rc = up(oLsp, lsp.data(), lsp.size() * 4)) ||
        (rc = up(oLsi, lsi.data(), lsi.size() * 4)) || (rc = up(oKfp, kfp.data(), kfp.size() * 4)) ||
        (rc = up(oKfi, kfi.data(), kfi.size() * 4)))
        return rc;
    double* dG = (double*)(dout + oG);
    // g layout = the reference's X layout: [6*nkf poses | 3*npt points | 6*nls lines]
    if (npt)
        hipLaunchKernelGGL(k_landmark_blocks<3>, dim3((npt + 255) / 256), dim3(256), 0, s, (int32_t*)(di + oPtp),
                           (int32_t*)(di + oPti), npt, (double*)(di + oPJl), (double*)(di + oPr), (double*)(di + oPw),
                           (double*)(dout + oHpt), dG + 6 * (size_t)nkf);
    if (nls)
        hipLaunchKernelGGL(k_landmark_blocks<6>, dim3((nls + 255) / 256), dim3(256), 0, s, (int32_t*)(di + oLsp),
                           (int32_t*)(di + oLsi), nls, (double*)(di + oLJl), (double*)(di + oLr), (double*)(di + oLw),
                           (double*)(dout + oHls), dG + 6 * (size_t)nkf + 3 * (size_t)npt);
    if (n_pt_obs)
        hipLaunchKernelGGL(k_cross_blocks<3>, dim3((n_pt_obs + 255) / 256), dim3(256), 0, s, (int32_t*)(di + oPk),
                           n_pt_obs, (double*)(di + oPJp), (double*)(di + oPJl), (double*)(di + oPw), (double*)(dout + oWp));
    if (n_ls_obs)
        hipLaunchKernelGGL(k_cross_blocks<6>, dim3((n_ls_obs + 255) / 256), dim3(256), 0, s, (int32_t*)(di + oLk),
                           n_ls_obs, (double*)(di + oLJp), (double*)(di + oLJl), (double*)(di + oLw), (double*)(dout + oWl));
    if (nkf)
        hipLaunchKernelGGL(k_pose_blocks, dim3(nkf), dim3(64), 0, s, (int32_t*)(di + oKfp), (int32_t*)(di + oKfi), n_pt_obs,
                           (double*)(di + oPJp), (double*)(di + oPr), (double*)(di + oPw), (double*)(di + oLJp),
                           (double*)(di + oLr), (double*)(di + oLw), (double*)(dout + oHp), dG);
    hipLaunchKernelGGL(k_weighted_error, dim3(1), dim3(256), 0, s, (double*)(di + oPr), (double*)(di + oPw), n_pt_obs,
                       (double*)(di + oLr), (double*)(di + oLw), n_ls_obs, (double*)(dout + oErr));
    PLSLAM_HIP_CHECK(hipGetLastError());
    auto down = [&](void* dst, size_t off, size_t bytes) -> int {
        if (bytes) PLSLAM_HIP_CHECK(hipMemcpyAsync(dst, dout + off, bytes, hipMemcpyDeviceToHost, s));
        return PLSLAM_OK;
    };
    if ((rc = down(g, oG, N * 8)) || (rc = down(H_pose, oHp, (size_t)nkf * 288)) || (rc = down(H_pt, oHpt, (size_t)npt * 72)) ||
        (rc = down(H_ls, oHls, (size_t)nls * 288)) || (rc = down(W_pt, oWp, np * 144)) || (rc = down(W_ls, oWl, nl * 288)) ||
        (rc = down(err, oErr, 8)))
        return rc;
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return PLSLAM_OK;
}
