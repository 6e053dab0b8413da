// map2kf.hip -- the map <-> keyframe association drivers of the reference as fused entry points:
// MapHandler::matchMap2KFPoints (src/mapHandler.cpp:532-632) and matchMap2KFLines (:634-752), brute
// force path (fast_matching == false), without the map mutation (that bookkeeping stays with the
// caller).  Projection / visibility pre-filter, Q/T descriptor matrix construction, the StVO::match
// itself and the epipolar inlier gate all run on the MI355X; the host only turns the visibility
// mask into index lists (one small D2H), exactly the role of the reference's std::vector building.
#include <vector>

#include "common.hpp"

namespace plslam {

__global__ void __launch_bounds__(256)
k_gather_rows(const uint64_t* __restrict__ src, const int32_t* __restrict__ idx, int32_t n, int32_t words,
              uint64_t* __restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (int64_t)n * words) return;
    const int32_t row = (int32_t)(t / words), w = (int32_t)(t % words);
    dst[t] = src[(int64_t)idx[row] * words + w];
}

int launch_gather_rows(const void* src, const int32_t* idx, int32_t n, int32_t row_bytes, void* dst,
                       hipStream_t s)
{
    if (n <= 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(row_bytes % 8 == 0, PLSLAM_EINVAL);
    const int32_t words = row_bytes / 8;
    const int64_t total = (int64_t)n * words;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       static_cast<const uint64_t*>(src), idx, n, words, static_cast<uint64_t*>(dst));
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

namespace {
struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~size_t(255); return o; }
};

int map2kf_driver(plslam_ctx* ctx, int lines, const plslam_cam* K, const double* Twf, const double* LM,
                  const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map, const uint8_t* kf_desc,
                  const double* kf_feat, const int32_t* kf_idx, int32_t n_kf, float nnr, int mutual,
                  double max_epip, int32_t min_matches, int32_t* map_to_kf, int32_t* n_matches)
{
    PLSLAM_REQUIRE(ctx && K && Twf && n_map >= 0 && n_kf >= 0, PLSLAM_EINVAL);
    if (n_matches) *n_matches = 0;
    if (n_map == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(LM && med_desc && candidate && map_to_kf, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(n_kf == 0 || (kf_desc && kf_feat && kf_idx), PLSLAM_EINVAL);
    for (int32_t i = 0; i < n_map; ++i) map_to_kf[i] = -1;
    const int lw = lines ? 6 : 3, fw = lines ? 3 : 2;

    // T list first (host only): unmatched keyframe features, :563-569 / :668-674
    std::vector<int32_t> ti;
    for (int32_t i = 0; i < n_kf; ++i)
        if (kf_idx[i] == -1) ti.push_back(i);
    const int32_t nt = (int32_t)ti.size();
    if (nt == 0) return PLSLAM_OK;                                        // :571 / :676

    std::lock_guard<std::mutex> lk(ctx->mu);
    hipStream_t s = ctx->stream;
    // ---- stage the map and the keyframe on the device, project + visibility test ------------
    Carve c;
    const size_t oLM = c.take((size_t)n_map * lw * 8), oMD = c.take((size_t)n_map * 32),
                 oKD = c.take((size_t)n_kf * 32), oKF = c.take((size_t)n_kf * fw * 8), oVis = c.take((size_t)n_map),
                 oQi = c.take((size_t)n_map * 4), oTi = c.take((size_t)nt * 4), oQ = c.take((size_t)n_map * 32),
                 oT = c.take((size_t)nt * 32), oQL = c.take((size_t)n_map * lw * 8), oTF = c.take((size_t)nt * fw * 8),
                 oM = c.take((size_t)n_map * 4), oMask = c.take((size_t)n_map), oCnt = c.take(8);
    int rc;
    if ((rc = ctx->misc_a.reserve(c.off))) return rc;
    char* d = ctx->misc_a.as<char>();
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oLM, LM, (size_t)n_map * lw * 8, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oMD, med_desc, (size_t)n_map * 32, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oKD, kf_desc, (size_t)n_kf * 32, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oKF, kf_feat, (size_t)n_kf * fw * 8, hipMemcpyHostToDevice, s));
    if ((rc = launch_visible(*K, Twf, (double*)(d + oLM), n_map, lines, (uint8_t*)(d + oVis), s))) return rc;
    std::vector<uint8_t> vis((size_t)n_map);
    PLSLAM_HIP_CHECK(hipMemcpyAsync(vis.data(), d + oVis, (size_t)n_map, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));

    // ---- Q list: candidate landmarks that project inside the image, :545-558 / :647-663 ------
    std::vector<int32_t> qi;
    for (int32_t i = 0; i < n_map; ++i)
        if (candidate[i] && vis[i]) qi.push_back(i);
    const int32_t nq = (int32_t)qi.size();
    if (nq == 0 || nq <= min_matches) return PLSLAM_OK;                   // :571, :594-596 / :709-711

    // ---- build the Q / T matrices on the device, match, gate ----------------------------------
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oQi, qi.data(), (size_t)nq * 4, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oTi, ti.data(), (size_t)nt * 4, hipMemcpyHostToDevice, s));
    if ((rc = launch_gather_rows(d + oMD, (int32_t*)(d + oQi), nq, 32, d + oQ, s))) return rc;
    if ((rc = launch_gather_rows(d + oKD, (int32_t*)(d + oTi), nt, 32, d + oT, s))) return rc;
    if ((rc = launch_gather_rows(d + oLM, (int32_t*)(d + oQi), nq, lw * 8, d + oQL, s))) return rc;
    if ((rc = launch_gather_rows(d + oKF, (int32_t*)(d + oTi), nt, fw * 8, d + oTF, s))) return rc;
    plslam_match_problem p{};
    p.d1 = (uint8_t*)(d + oQ); p.n1 = nq; p.d2 = (uint8_t*)(d + oT); p.n2 = nt;
    p.nnr = nnr; p.mutual = mutual ? 1 : 0; p.matches_12 = (int32_t*)(d + oM); p.n_matches = nullptr;
    if ((rc = match_problems_on_ctx_stream(ctx, &p, 1))) return rc;       // :597 / :712
    rc = lines ? launch_line_gate(*K, Twf, (double*)(d + oQL), (int32_t*)(d + oM), nq, (double*)(d + oTF),
                                  max_epip, (uint8_t*)(d + oMask), (int32_t*)(d + oCnt), s)
               : launch_point_gate(*K, Twf, (double*)(d + oQL), (int32_t*)(d + oM), nq, (double*)(d + oTF),
                                   max_epip, (uint8_t*)(d + oMask), (int32_t*)(d + oCnt), s);
    if (rc) return rc;
    std::vector<int32_t> m12((size_t)nq);
    std::vector<uint8_t> mask((size_t)nq);
    int32_t cnt = 0;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(m12.data(), d + oM, (size_t)nq * 4, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(mask.data(), d + oMask, (size_t)nq, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(&cnt, d + oCnt, 4, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    for (int32_t a = 0; a < nq; ++a)
        if (mask[a]) map_to_kf[qi[a]] = ti[m12[a]];                       // :614-619 (the association)
    if (n_matches) *n_matches = cnt;
    return PLSLAM_OK;
}
}  // namespace
}  // namespace plslam

extern "C" {

int plslam_map2kf_match_points(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* Xw,
                               const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                               const uint8_t* kf_desc, const double* kf_pl, const int32_t* kf_idx,
                               int32_t n_kf, float nnr, int mutual, double max_epip, int32_t min_matches,
                               int32_t* map_to_kf, int32_t* n_matches)
{
    return plslam::map2kf_driver(ctx, 0, K, Twf, Xw, med_desc, candidate, n_map, kf_desc, kf_pl, kf_idx, n_kf,
                                 nnr, mutual, max_epip, min_matches, map_to_kf, n_matches);
}

int plslam_map2kf_match_lines(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* Lw,
                              const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                              const uint8_t* kf_desc, const double* kf_le, const int32_t* kf_idx,
                              int32_t n_kf, float nnr, int mutual, double max_epip, int32_t min_matches,
                              int32_t* map_to_kf, int32_t* n_matches)
{
    return plslam::map2kf_driver(ctx, 1, K, Twf, Lw, med_desc, candidate, n_map, kf_desc, kf_le, kf_idx, n_kf,
                                 nnr, mutual, max_epip, min_matches, map_to_kf, n_matches);
}

}  // extern "C"
