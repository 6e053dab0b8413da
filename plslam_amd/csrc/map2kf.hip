// map2kf.hip -- the map <-> keyframe association drivers of the reference as fused entry points:
// MapHandler::matchMap2KFPoints (src/mapHandler.cpp:532-632) and matchMap2KFLines (:634-752), with and
// without SlamConfig::fastMatching(), without the map mutation (that bookkeeping stays with the
// caller).  Projection / visibility pre-filter, Q/T descriptor matrix construction, the projection of
// the candidates into grid cells, StVO::matchGrid / StVO::match and the epipolar inlier gate all run on
// the MI355X; the host turns the visibility mask into index lists and fills the GridStructure (cell
// lists of the unmatched keyframe features, :580-584 / :683-699) -- the list building the reference's
// callers do with std::vector / std::list.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace plslam {

// match_grid.hip: one matchGrid problem on `s`, with its scratch (a large mutual problem gets its distances from a
// many-workgroup launch)
int grid_launch_single(const plslam_grid_problem& q, const GridDesc* d_desc, hipStream_t s, uint32_t* aux, bool n1_upper_bound,
                       const GridDesc* h_desc = nullptr);   // h_desc: the host's copy (kernels of a lone problem take it by value)
bool grid_dense_ok(int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs, int32_t n_centres);   // the one-workgroup kernel takes it
size_t grid_aux_words(int32_t n2);            // the words the two launches share, prefilled by grid_aux_fill in the upload image
void grid_aux_fill(void* host_image, int32_t n2);
// lba.hip: the visibility pre-filter AND the candidate flags, both on the device
int launch_visible_cand(const plslam_cam& K, const double* Twf16, const double* X, const uint8_t* cand, int32_t n, int lines,
                        uint8_t* vis, hipStream_t s);
// lba.hip: the epipolar gate with the row count on the device (*n_dev <= n_max) and, optionally, the association behind it
int launch_gate_n(int lines, const plslam_cam& K, const double* Twf16, const double* LM, const int32_t* m12, const int32_t* n_dev,
                  int32_t n_max, const double* feat, double th, uint8_t* mask, int32_t* count, const int32_t* idx, const int32_t* ti,
                  int32_t* map_to_kf, hipStream_t s, int32_t* publish_done = nullptr, const int32_t* publish_src = nullptr,
                  int32_t* publish_dst = nullptr, int32_t publish_n = 0);
// lba.hip: visibility x candidate flags -> stable list + length (+ -1 fill of the association table, + the row count into a
// matchGrid descriptor); Q rows + landmarks + window centres of the listed landmarks
size_t visible_compact_part_words(int32_t n);       // zeroed device words the kernel's workgroups chain their counts through
int launch_visible_compact(const plslam_cam& K, const double* Twf16, const double* X, const uint8_t* cand, int32_t n, int lines,
                           int32_t* idx, int32_t* n_out, int32_t* fill, GridDesc* desc, uint32_t* part, bool part_zeroed, hipStream_t s);
int launch_prepare_rows(const plslam_cam& K, const double* Twf16, const void* md, const double* lm, const int32_t* idx,
                        const int32_t* n_dev, int32_t n_max, int lines, double inv_w, double inv_h, void* Q, double* QL, int32_t* cells,
                        double* dir1, hipStream_t s);
// match_grid.hip: capacity of the windowed matcher's candidate store from the grid alone
int64_t grid_store_capacity_bound(int32_t n1, int32_t n_centres, const int32_t* cell_start, int32_t cols, int32_t rows,
                                  const int32_t window[4], int mutual);

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


// the drivers' counters, device -> the page-locked block the host reads (one lane per word): a LAUNCH behind the last kernel
// instead of a copy command -- on the timeline a small device-to-host copy starts ~13 us after the kernel in front of it, a kernel
// 2-3 us (profiles/r6_r_call_timeline_map2kf_points_fast.txt)
__global__ void k_publish_words(const int32_t* __restrict__ src, int32_t* __restrict__ dst, int n)
{
    if ((int)threadIdx.x < n) dst[threadIdx.x] = src[threadIdx.x];
}

// (env PLSLAM_MAP2KF_TABLE_IN_PLACE=0: the association table on the device and one copy at the call's end, as before round 6 -- for
// same-box comparisons)
static bool tab_in_place_enabled()
{
    static const bool on = [] { const char* e = getenv("PLSLAM_MAP2KF_TABLE_IN_PLACE"); return !(e && e[0] == '0'); }();
    return on;
}

namespace {
struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~size_t(255); return o; }
};

// getLineCoords of stvo-pl ([RECALL]; the callers' grid fill :693-697): Bresenham cells, the last x excluded
void line_cells(double x1, double y1, double x2, double y2, std::vector<int32_t>& xy)
{
    xy.clear();
    const bool steep = std::fabs(y2 - y1) > std::fabs(x2 - x1);
    if (steep) { std::swap(x1, y1); std::swap(x2, y2); }
    if (x1 > x2) { std::swap(x1, x2); std::swap(y1, y2); }
    const double dx = x2 - x1, dy = std::fabs(y2 - y1);
    double error = dx / 2.0;
    const int ystep = (y1 < y2) ? 1 : -1;
    int y = (int)y1;
    const int maxX = (int)x2;
    for (int x = (int)x1; x < maxX; x++) {
        xy.push_back(steep ? y : x);
        xy.push_back(steep ? x : y);
        error -= dy;
        if (error < 0) { y += ystep; error += dx; }
    }
}

// GridStructure fill in CSR form: `cells` lists (item, x, y) in push_back order
void csr_fill(const std::vector<int32_t>& item, const std::vector<int32_t>& cx, const std::vector<int32_t>& cy,
              int32_t cols, int32_t rows, std::vector<int32_t>& cs, std::vector<int32_t>& items)
{
    cs.assign((size_t)cols * rows + 1, 0);
    for (size_t k = 0; k < item.size(); ++k)
        if (cx[k] >= 0 && cx[k] < cols && cy[k] >= 0 && cy[k] < rows) ++cs[(size_t)cx[k] * rows + cy[k] + 1];
    for (size_t c = 0; c < (size_t)cols * rows; ++c) cs[c + 1] += cs[c];
    items.assign((size_t)cs.back() + 1, 0);
    std::vector<int32_t> fill(cs.begin(), cs.end() - 1);
    for (size_t k = 0; k < item.size(); ++k)
        if (cx[k] >= 0 && cx[k] < cols && cy[k] >= 0 && cy[k] < rows) items[(size_t)fill[(size_t)cx[k] * rows + cy[k]]++] = item[k];
}

inline int32_t cvtt_x86(double v)   // as the reference's x86 build converts (cvttsd2si): NaN / out of range -> INT_MIN
{
    return (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN;
}

// the GridStructure of the keyframe features feat_curr[sel ? sel[b] : b], b < nt, in CSR form (points: the feature's cell,
// :581-584; lines: the Bresenham cells of (spl, epl), :686-698) and, for lines, their directions
void fill_grid_tables(int lines, const double* feat_curr, const int32_t* sel, int32_t nt, const plslam_fast_matching* fm,
                      std::vector<int32_t>& cs, std::vector<int32_t>& items, std::vector<double>& dir2)
{
    const int32_t cols = fm->grid_cols, rows = fm->grid_rows;
    std::vector<int32_t> it, cx, cy, xy;
    if (!lines) {
        for (int32_t b = 0; b < nt; ++b) {
            const double* p = feat_curr + 2 * (size_t)(sel ? sel[b] : b);
            it.push_back(b);
            cx.push_back(cvtt_x86(p[0] * fm->inv_width));
            cy.push_back(cvtt_x86(p[1] * fm->inv_height));
        }
    } else {
        dir2.resize((size_t)nt * 2);
        for (int32_t b = 0; b < nt; ++b) {
            const double* sg = feat_curr + 4 * (size_t)(sel ? sel[b] : b);
            double vx = (sg[2] - sg[0]) * fm->inv_width, vy = (sg[3] - sg[1]) * fm->inv_height;
            const double magnitude = std::sqrt(vx * vx + vy * vy);
            dir2[2 * (size_t)b] = vx / magnitude;
            dir2[2 * (size_t)b + 1] = vy / magnitude;
            line_cells(sg[0] * fm->inv_width, sg[1] * fm->inv_height, sg[2] * fm->inv_width, sg[3] * fm->inv_height, xy);
            for (size_t k = 0; k + 1 < xy.size(); k += 2) {
                it.push_back(b);
                cx.push_back(xy[k]);
                cy.push_back(xy[k + 1]);
            }
        }
    }
    csr_fill(it, cx, cy, cols, rows, cs, items);
}

// StVO::matchGrid over projected 3D features, device-resident except the grid fill:
//   d_X3 (device): nq x 3 (points) / nq x 6 (lines) features, projected with T16 into cells * (sx, sy) by K16';
//   feat_curr (HOST): the keyframe features that fill the GridStructure -- item b = feat_curr[sel ? sel[b] : b],
//   2 doubles (pl) or 4 (spl, epl); d_Q / d_T (device): the descriptor matrices; d_m12: nq entries out (device memory, or
//   the device address of page-locked host memory: the caller then has the table without a copy).
// The host-built tables (cell_start, items, directions) and the problem descriptor travel in ONE page-locked image = one
// upload; the count comes back through page-locked memory written by the kernel.  Uses ctx->pin_misc, ctx->misc_b (tables)
// and ctx->misc_c (kernel scratch).  Synchronises the stream once, at the end.
int grid_path(plslam_ctx* ctx, int lines, const plslam_cam* K, const double* T16, const double* d_X3, int32_t nq, double sx,
              double sy, const uint8_t* d_Q, const double* feat_curr, const int32_t* sel, int32_t nt, const uint8_t* d_T,
              const plslam_fast_matching* fm, int mutual, int32_t* d_m12, int32_t* matches, const int32_t** deferred = nullptr)
{
    // deferred != nullptr: nothing is waited for here -- the caller has more work for the stream and synchronises itself; then
    // (*deferred)[0] is the count and (*deferred)[1] must be 0 ((*deferred)[0] >= 0): grid_path_check
    hipStream_t s = ctx->stream;
    StreamSyncOnError sg(s);
    int rc;
    const int nc = lines ? 2 : 1;
    const int32_t cols = fm->grid_cols, rows = fm->grid_rows;
    std::vector<int32_t> cs, items;
    std::vector<double> dir2;
    fill_grid_tables(lines, feat_curr, sel, nt, fm, cs, items, dir2);
    const int32_t n_items = cs.back();
    // image (host -> device in one copy): result words | cell_start | items | directions | descriptor; behind it, device only:
    // the projected cells and the query directions
    Carve cf;
    const size_t oSt = cf.take(16), oCs = cf.take(cs.size() * 4), oIt = cf.take((size_t)(n_items + 1) * 4),
                 oD2 = cf.take(lines ? (size_t)nt * 16 : 0), oDesc = cf.take(sizeof(GridDesc)), oAux = cf.take(grid_aux_words(nt) * 4);
    const size_t image = cf.off;
    const size_t oCen = cf.take((size_t)nq * nc * 8), oD1 = cf.take(lines ? (size_t)nq * 16 : 0);
    if ((rc = ctx->pin_misc.reserve(image))) return rc;
    if ((rc = ctx->misc_b.reserve(cf.off))) return rc;
    char* h = ctx->pin_misc.as<char>();
    char* f = ctx->misc_b.as<char>();
    // (the image of a problem the dense one-workgroup kernel takes is read ONCE, into LDS: it is read where it lies in page-locked
    // memory -- ctx option "zero_copy_kb", as plslam_match_grid -- and one copy command leaves the call; the projected cells and
    // the query directions, written by a kernel, stay in device memory: fi = the image's base as the kernels see it)
    const bool dense = grid_dense_ok(nq, nt, (int64_t)cols * rows, n_items, lines != 0, nc);
    const size_t zc_limit = (size_t)(ctx->zero_copy_kb < 0 ? -ctx->zero_copy_kb : ctx->zero_copy_kb) * 1024;
    const bool zero_copy = ctx->pin_misc.dev && zc_limit > 0 && image <= zc_limit && (dense || ctx->zero_copy_kb < 0);
    char* fi = zero_copy ? static_cast<char*>(ctx->pin_misc.dev) : f;
    memcpy(h + oCs, cs.data(), cs.size() * 4);
    memcpy(h + oIt, items.data(), (size_t)(n_items + 1) * 4);
    if (lines) memcpy(h + oD2, dir2.data(), (size_t)nt * 16);
    // capacity of the candidate store from the grid alone (fullest cell x cells of a window, at most every item, per window
    // centre; rows in blocks of 1024): the projected cells stay on the device, no round trip before the matcher is launched
    const int32_t win[4] = {fm->ws, fm->ws, fm->ws, fm->ws};
    const int64_t cap = grid_store_capacity_bound(nq, nc, cs.data(), cols, rows, win, mutual);
    PLSLAM_REQUIRE(cap < (int64_t(1) << 31) - 1, PLSLAM_ERANGE);
    if ((rc = ctx->misc_c.reserve(grid_scratch_words(nq, nt, (int64_t)cols * rows, (int32_t)cap) * 4 + 256))) return rc;
    // the count: written by the kernel into the page-locked image when the device can address it (no status word then: it
    // is bumped with an atomic; an overflow also shows as a count of -1)
    int32_t* res_host = (int32_t*)(h + oSt);
    int32_t* res_dev = ctx->pin_misc.dev ? (int32_t*)(static_cast<char*>(ctx->pin_misc.dev) + oSt) : nullptr;
    const bool in_place = res_dev != nullptr;
    res_host[0] = res_host[1] = 0;
    plslam_grid_problem q{};
    q.d1 = d_Q; q.d2 = d_T; q.centres1 = (int32_t*)(f + oCen);
    q.cell_start = (int32_t*)(fi + oCs); q.cell_items = (int32_t*)(fi + oIt);
    q.dir1 = lines ? (double*)(f + oD1) : nullptr; q.dir2 = lines ? (double*)(fi + oD2) : nullptr;
    q.n1 = nq; q.n2 = nt; q.n_centres = nc; q.grid_cols = cols; q.grid_rows = rows; q.n_items = n_items;
    for (int k = 0; k < 4; ++k) q.window[k] = fm->ws;
    q.sim_th = fm->line_sim_th; q.nnr = fm->nnr_grid; q.mutual = mutual ? 1 : 0;
    q.pair_capacity = (int32_t)cap;
    q.matches_12 = d_m12; q.n_matches = in_place ? res_dev : (int32_t*)(f + oSt);
    if ((rc = grid_prepare_one(q, ctx->misc_c.as<uint32_t>(), in_place ? nullptr : (int32_t*)(f + oSt) + 1, (GridDesc*)(h + oDesc))))
        return rc;
    grid_aux_fill(h + oAux, nt);
    if (!zero_copy) PLSLAM_HIP_CHECK(hipMemcpyAsync(f, h, image, hipMemcpyHostToDevice, s));      // (zeroes the device result words too)
    else if (!in_place) PLSLAM_HIP_CHECK(hipMemsetAsync(f + oSt, 0, 16, s));
    if ((rc = launch_project_cells(*K, T16, d_X3, nq, lines, sx, sy, (int32_t*)(f + oCen),
                                   lines ? (double*)(f + oD1) : nullptr, s)))
        return rc;
    if ((rc = grid_launch_single(q, (const GridDesc*)(fi + oDesc), s, (uint32_t*)(fi + oAux), false, (const GridDesc*)(h + oDesc)))) return rc;
    if (!in_place) PLSLAM_HIP_CHECK(hipMemcpyAsync(res_host, f + oSt, 8, hipMemcpyDeviceToHost, s));
    if (deferred) {
        sg.dismiss();
        *deferred = res_host;
        return PLSLAM_OK;
    }
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    sg.dismiss();
    PLSLAM_REQUIRE(res_host[1] == 0 && res_host[0] >= 0, PLSLAM_ERANGE);
    *matches = res_host[0];
    return PLSLAM_OK;
}

// An upper bound of what the windowed matcher can find for the KF<->KF LINE call, from the host's copy of the previous key frame's
// lines: a row can only match if one of its two window centres has a grid cell in its window.  The reference hands matchGrid the
// projected end points in PIXELS there (:392-393; the point call scales by the grid, :256), so on a 752 x 480 image nearly every
// window lies outside the 64 x 48 grid and the windowed pass finds next to nothing -- StVO::match runs behind it (:421-425) in
// practically every call.  When this bound is below min_matches that is KNOWN before anything is launched, and the driver
// enqueues both matchers and waits once.  Conservative: the host's projection may differ from the device's in the last bit, so a
// centre within two cells of the window's reach, or not finite, counts as "may have candidates".
int32_t kf2kf_line_grid_bound(const plslam_cam* K, const double* DT, const double* sPeP, int32_t n, const plslam_fast_matching* fm)
{
    const double reach = (double)fm->ws + 2.0;
    int32_t may = 0;
    for (int32_t i = 0; i < n; ++i) {
        bool any = false;
        for (int e = 0; e < 2 && !any; ++e) {
            const double* X = sPeP + 6 * (size_t)i + 3 * e;
            double P[3];
            for (int r = 0; r < 3; ++r) P[r] = DT[4 * r] * X[0] + DT[4 * r + 1] * X[1] + DT[4 * r + 2] * X[2] + DT[4 * r + 3];
            const double u = K->cx + K->fx * P[0] / P[2], v = K->cy + K->fy * P[1] / P[2];
            const bool out = u < -reach || u > (double)fm->grid_cols + reach || v < -reach || v > (double)fm->grid_rows + reach;
            any = !out;                            // (NaN compares false everywhere: "may")
        }
        may += any;
    }
    return may;
}

bool fast_ok(const plslam_fast_matching* fm)
{
    return fm->grid_cols >= 1 && fm->grid_rows >= 1 && fm->ws >= 0 && (int64_t)fm->grid_cols * fm->grid_rows < (int64_t(1) << 30);
}

// The map<->keyframe driver with fast_matching as ONE launch sequence and ONE synchronisation: the candidate list is built on
// the device (visibility [x candidate flags] -> stable compaction), so its length nq stays there -- the gathers, the projection,
// the gate and the association read it from device memory, matchGrid's descriptor is patched with it behind the upload (its
// launch geometry follows the upper bound n_map; k_match_grid / k_grid_candidates pick their form from the patched row count)
// -- and the one host decision of the reference loop that needs it, `|Q| > min && matches < min` (:594-598, :709-713), is taken
// AFTER the results are back: the brute-force matcher then runs on what is still resident (Q, T, matchGrid's table as the
// vector it is handed), followed by the gate -- a second, short launch sequence.  *redo = 1 only when the candidate store would
// have to be sized for every landmark of a huge map: the caller runs the step-by-step form.
// Everything the host knows beforehand travels in ONE upload: [the map, unless it is resident] | T rows | their features | ti |
// the grid of the unmatched keyframe features | matchGrid's descriptor | zeroed counters.  Caller holds ctx->mu.
int map2kf_fast_once(plslam_ctx* ctx, int lines, const plslam_cam* K, const double* Twf, const double* LM, const uint8_t* med_desc,
                     const uint8_t* candidate, int32_t n_map, const uint8_t* kf_desc, const double* kf_feat, const double* kf_seg,
                     const std::vector<int32_t>& ti, float nnr, int mutual, double max_epip, int32_t min_matches,
                     const plslam_fast_matching* fm, int32_t* map_to_kf, int32_t* n_matches, int32_t* used_match, bool map_dev, int* redo)
{
    *redo = 0;
    hipStream_t s = ctx->stream;
    StreamSyncOnError sg(s);
    int rc;
    const int32_t nt = (int32_t)ti.size();
    const int lw = lines ? 6 : 3, fw = lines ? 3 : 2, nc = lines ? 2 : 1;
    const int32_t cols = fm->grid_cols, rows = fm->grid_rows;
    std::vector<int32_t> cs, items;
    std::vector<double> dir2;
    fill_grid_tables(lines, lines ? kf_seg : kf_feat, ti.data(), nt, fm, cs, items, dir2);
    const int32_t n_items = cs.back();
    const int32_t win[4] = {fm->ws, fm->ws, fm->ws, fm->ws};
    const int64_t cap = grid_store_capacity_bound(n_map, nc, cs.data(), cols, rows, win, mutual);     // (rows: the upper bound)
    if (cap >= (int64_t(1) << 28)) {       // a store sized for every landmark of a huge map: the step-by-step form sizes it for the list
        *redo = 1;
        sg.dismiss();
        return PLSLAM_OK;
    }
    // ---- one image up
    Carve c;
    const size_t oLM = c.take(map_dev ? 0 : (size_t)n_map * lw * 8), oMD = c.take(map_dev ? 0 : (size_t)n_map * 32),
                 oCand = c.take(map_dev ? 0 : (size_t)n_map), oT = c.take((size_t)nt * 32), oTF = c.take((size_t)nt * fw * 8),
                 oTi = c.take((size_t)nt * 4), oCs = c.take(cs.size() * 4), oIt = c.take((size_t)(n_items + 1) * 4),
                 oD2 = c.take(lines ? (size_t)nt * 16 : 0), oDesc = c.take(sizeof(GridDesc)), oAux = c.take(grid_aux_words(nt) * 4),
                 oRes = c.take(16),                                      // gate count | nq | matchGrid's count | -
                 oPart = c.take(visible_compact_part_words(n_map) * 4);  // k_visible_compact's chain (zero)
    const size_t image = c.off;
    // ---- device only
    const size_t oMap = c.take((size_t)n_map * 4);                       // the association table: directly behind the counters' page
    const size_t oQi = c.take((size_t)n_map * 4), oQ = c.take((size_t)n_map * 32),
                 oQL = c.take((size_t)n_map * lw * 8), oCen = c.take((size_t)n_map * nc * 8),
                 oD1 = c.take(lines ? (size_t)n_map * 16 : 0), oM = c.take((size_t)n_map * 4), oMask = c.take((size_t)n_map);
    if ((rc = ctx->misc_a.reserve(c.off))) return rc;
    if ((rc = ctx->pin_in.reserve(image))) return rc;
    if ((rc = ctx->pin_out.reserve((oMap - oRes) + (size_t)n_map * 4))) return rc;      // the counters' pages + the table behind them
    if ((rc = ctx->misc_c.reserve(grid_scratch_words(n_map, nt, (int64_t)cols * rows, (int32_t)cap) * 4 + 256))) return rc;
    char* d = ctx->misc_a.as<char>();
    char* h = ctx->pin_in.as<char>();
    if (!map_dev) {
        memcpy(h + oLM, LM, (size_t)n_map * lw * 8);
        memcpy(h + oMD, med_desc, (size_t)n_map * 32);
        memcpy(h + oCand, candidate, (size_t)n_map);
    }
    const char* const d_LM = map_dev ? reinterpret_cast<const char*>(LM) : d + oLM;
    const char* const d_MD = map_dev ? reinterpret_cast<const char*>(med_desc) : d + oMD;
    const uint8_t* const d_cand = map_dev ? candidate : (const uint8_t*)(d + oCand);
    for (int32_t b = 0; b < nt; ++b) {                                   // the T matrix and its features, gathered here (:563-569)
        memcpy(h + oT + (size_t)b * 32, kf_desc + (size_t)ti[b] * 32, 32);
        memcpy(h + oTF + (size_t)b * fw * 8, kf_feat + (size_t)ti[b] * fw, (size_t)fw * 8);
    }
    memcpy(h + oTi, ti.data(), (size_t)nt * 4);
    memcpy(h + oCs, cs.data(), cs.size() * 4);
    memcpy(h + oIt, items.data(), (size_t)(n_items + 1) * 4);
    if (lines) memcpy(h + oD2, dir2.data(), (size_t)nt * 16);
    memset(h + oRes, 0, 16);
    memset(h + oPart, 0, visible_compact_part_words(n_map) * 4);
    grid_aux_fill(h + oAux, nt);
    int32_t* const res = (int32_t*)(d + oRes);                           // [0] gate count, [1] nq, [2] matchGrid's count
    // the association table: written where the host reads it (the page-locked block, mapped) when the device can address it --
    // its -1 fill and the few hundred entries the gate sets cross PCIe as posted writes; the counters follow by k_publish_words:
    // no copy command at the call's end
    char* ho = ctx->pin_out.as<char>();
    const bool tab_in_place = ctx->pin_out.dev != nullptr && tab_in_place_enabled();
    int32_t* const tab = tab_in_place ? (int32_t*)(static_cast<char*>(ctx->pin_out.dev) + (oMap - oRes)) : (int32_t*)(d + oMap);
    plslam_grid_problem q{};
    q.d1 = (const uint8_t*)(d + oQ); q.d2 = (const uint8_t*)(d + oT); q.centres1 = (int32_t*)(d + oCen);
    q.cell_start = (int32_t*)(d + oCs); q.cell_items = (int32_t*)(d + oIt);
    q.dir1 = lines ? (double*)(d + oD1) : nullptr; q.dir2 = lines ? (double*)(d + oD2) : nullptr;
    q.n1 = n_map; q.n2 = nt; q.n_centres = nc; q.grid_cols = cols; q.grid_rows = rows; q.n_items = n_items;
    for (int k = 0; k < 4; ++k) q.window[k] = fm->ws;
    q.sim_th = fm->line_sim_th; q.nnr = fm->nnr_grid; q.mutual = mutual ? 1 : 0;
    q.pair_capacity = (int32_t)cap;
    q.matches_12 = (int32_t*)(d + oM); q.n_matches = res + 2;
    if ((rc = grid_prepare_one(q, ctx->misc_c.as<uint32_t>(), nullptr, (GridDesc*)(h + oDesc)))) return rc;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d, h, image, hipMemcpyHostToDevice, s));
    // ---- the launch sequence (five launches: the small steps are fused -- a launch of a microsecond's work costs 4-5 us)
    if ((rc = launch_visible_compact(*K, Twf, (const double*)d_LM, d_cand, n_map, lines, (int32_t*)(d + oQi), res + 1,
                                     tab, (GridDesc*)(d + oDesc), (uint32_t*)(d + oPart), /* zeroed by the image above */ true, s)))
        return rc;
    if ((rc = launch_prepare_rows(*K, Twf, d_MD, (const double*)d_LM, (const int32_t*)(d + oQi), res + 1, n_map, lines, fm->inv_width,
                                  fm->inv_height, d + oQ, (double*)(d + oQL), (int32_t*)(d + oCen), lines ? (double*)(d + oD1) : nullptr, s)))
        return rc;
    if ((rc = grid_launch_single(q, (const GridDesc*)(d + oDesc), s, (uint32_t*)(d + oAux), true, (const GridDesc*)(h + oDesc)))) return rc;
    if ((rc = launch_gate_n(lines, *K, Twf, (const double*)(d + oQL), (const int32_t*)(d + oM), res + 1, n_map, (const double*)(d + oTF),
                            max_epip, (uint8_t*)(d + oMask), res, (const int32_t*)(d + oQi), (const int32_t*)(d + oTi), tab, s,
                            tab_in_place ? res + 3 : nullptr, res, (int32_t*)ctx->pin_out.dev, 3)))     // (res[3]: zero in the image)
        return rc;
    // ---- the counters (from the gate's last workgroup) and, when it is not there already, the table behind them down; one synchronisation
    if (!tab_in_place) {
        PLSLAM_HIP_CHECK(hipMemcpyAsync(ho, d + oRes, (oMap - oRes) + (size_t)n_map * 4, hipMemcpyDeviceToHost, s));
    }
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    sg.dismiss();
    const int32_t* r = reinterpret_cast<const int32_t*>(ho);
    PLSLAM_REQUIRE(r[2] >= 0, PLSLAM_ERANGE);                            // (matchGrid's candidate store: the capacity is an upper bound)
    if (r[1] > min_matches && r[2] < min_matches) {
        // :594-598 / :709-713 -- matchGrid found too little: StVO::match runs over the SAME Q and T with the vector matchGrid
        // filled (keep_prior), then the gate.  Everything it needs is still on the device -- Q, T, their features, the lists,
        // matchGrid's table -- and the list's length is known now: no re-staging, no second visibility pass.  The association
        // table and the gate's counter start over.
        StreamSyncOnError sg2(s);
        const int32_t nq = r[1];
        PLSLAM_HIP_CHECK(hipMemsetAsync(d + oMap, 0xFF, (size_t)n_map * 4, s));
        PLSLAM_HIP_CHECK(hipMemsetAsync(res, 0, 4, s));
        plslam_match_problem p{};
        p.d1 = (const uint8_t*)(d + oQ); p.n1 = nq; p.d2 = (const uint8_t*)(d + oT); p.n2 = nt;
        p.nnr = nnr; p.mutual = mutual ? 1 : 0; p.matches_12 = (int32_t*)(d + oM); p.n_matches = nullptr; p.keep_prior = 1;
        if ((rc = match_problems_on_ctx_stream(ctx, &p, 1))) return rc;
        if ((rc = launch_gate_n(lines, *K, Twf, (const double*)(d + oQL), (const int32_t*)(d + oM), res + 1, n_map,
                                (const double*)(d + oTF), max_epip, (uint8_t*)(d + oMask), res, (const int32_t*)(d + oQi),
                                (const int32_t*)(d + oTi), (int32_t*)(d + oMap), s)))
            return rc;
        PLSLAM_HIP_CHECK(hipMemcpyAsync(ho, d + oRes, (oMap - oRes) + (size_t)n_map * 4, hipMemcpyDeviceToHost, s));
        PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
        sg2.dismiss();
        if (used_match) *used_match = 1;
    }
    memcpy(map_to_kf, ho + (oMap - oRes), (size_t)n_map * 4);
    if (n_matches) *n_matches = r[0];
    return PLSLAM_OK;
}

// The map<->keyframe driver WITHOUT fast_matching (:594-598 / :709-713 with an empty matches_12: StVO::match over all of Q) as
// ONE launch sequence and ONE synchronisation.  As in map2kf_fast_once the candidate list is built on the device and its length
// nq stays there; the matcher is a two-launch column-split plan sized for the bound n_map whose kernels read the row count from
// device memory (SymDesc::n1_dev: workgroups behind the last row leave at once), the gate and the association read it too.  The
// one host decision that needs nq -- match() runs only if |Q| > min_matches -- is applied after the results are back (the
// matcher's table is then simply not used: every entry of map_to_kf stays -1, as when no matcher ran).  *done = 0: the plan
// cannot take a device-side row count under the context's options (or the problem is not mutual) -- asked FIRST
// (ctx_takes_device_row_count), so nothing is staged, uploaded or enqueued -- and the caller runs the step-by-step form.
// Caller holds ctx->mu.
int map2kf_bf_once(plslam_ctx* ctx, int lines, const plslam_cam* K, const double* Twf, const double* LM, const uint8_t* med_desc,
                   const uint8_t* candidate, int32_t n_map, const uint8_t* kf_desc, const double* kf_feat,
                   const std::vector<int32_t>& ti, float nnr, int mutual, double max_epip, int32_t min_matches,
                   int32_t* map_to_kf, int32_t* n_matches, int32_t* used_match, bool map_dev, int* done)
{
    *done = 0;
    if (!mutual || n_map > PLSLAM_MAX_TRAIN_ROWS) return PLSLAM_OK;       // (the plan is sized for the bound n_map, not for |Q|)
    if (!ctx_takes_device_row_count(ctx) || n_map <= 0 || ti.empty()) return PLSLAM_OK;   // (before anything is staged or enqueued)
    hipStream_t s = ctx->stream;
    StreamSyncOnError sg(s);
    int rc;
    const int32_t nt = (int32_t)ti.size();
    const int lw = lines ? 6 : 3, fw = lines ? 3 : 2;
    // ---- one image up: [the map, unless it is resident] | T rows | their features | ti | zeroed counters
    Carve c;
    const size_t oLM = c.take(map_dev ? 0 : (size_t)n_map * lw * 8), oMD = c.take(map_dev ? 0 : (size_t)n_map * 32),
                 oCand = c.take(map_dev ? 0 : (size_t)n_map), oT = c.take((size_t)nt * 32), oTF = c.take((size_t)nt * fw * 8),
                 oTi = c.take((size_t)nt * 4),
                 oRes = c.take(16),                                      // gate count | nq | - | -
                 oPart = c.take(visible_compact_part_words(n_map) * 4);  // k_visible_compact's chain (zero)
    const size_t image = c.off;
    // ---- device only
    const size_t oMap = c.take((size_t)n_map * 4);                       // the association table: directly behind the counters' page
    const size_t oQi = c.take((size_t)n_map * 4), oQ = c.take((size_t)n_map * 32), oQL = c.take((size_t)n_map * lw * 8),
                 oM = c.take((size_t)n_map * 4), oMask = c.take((size_t)n_map);
    if ((rc = ctx->misc_a.reserve(c.off))) return rc;
    if ((rc = ctx->pin_in.reserve(image))) return rc;
    if ((rc = ctx->pin_out.reserve((oMap - oRes) + (size_t)n_map * 4))) return rc;      // the counters' pages + the table behind them
    char* d = ctx->misc_a.as<char>();
    char* h = ctx->pin_in.as<char>();
    int32_t* const res = (int32_t*)(d + oRes);                           // [0] gate count, [1] nq
    // the matcher's plan first: it decides whether this form applies at all (nothing is enqueued before it says yes)
    plslam_match_problem p{};
    p.d1 = (const uint8_t*)(d + oQ); p.n1 = n_map; p.d2 = (const uint8_t*)(d + oT); p.n2 = nt;
    p.nnr = nnr; p.mutual = 1; p.matches_12 = (int32_t*)(d + oM); p.n_matches = nullptr; p.keep_prior = 0;
    if (!map_dev) {
        memcpy(h + oLM, LM, (size_t)n_map * lw * 8);
        memcpy(h + oMD, med_desc, (size_t)n_map * 32);
        memcpy(h + oCand, candidate, (size_t)n_map);
    }
    const char* const d_LM = map_dev ? reinterpret_cast<const char*>(LM) : d + oLM;
    const char* const d_MD = map_dev ? reinterpret_cast<const char*>(med_desc) : d + oMD;
    const uint8_t* const d_cand = map_dev ? candidate : (const uint8_t*)(d + oCand);
    for (int32_t b = 0; b < nt; ++b) {                                   // the T matrix and its features, gathered here (:563-569)
        memcpy(h + oT + (size_t)b * 32, kf_desc + (size_t)ti[b] * 32, 32);
        memcpy(h + oTF + (size_t)b * fw * 8, kf_feat + (size_t)ti[b] * fw, (size_t)fw * 8);
    }
    memcpy(h + oTi, ti.data(), (size_t)nt * 4);
    memset(h + oRes, 0, 16);
    memset(h + oPart, 0, visible_compact_part_words(n_map) * 4);
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d, h, image, hipMemcpyHostToDevice, s));
    // (the association table where the host reads it, the counters by k_publish_words: as map2kf_fast_once)
    char* ho = ctx->pin_out.as<char>();
    const bool tab_in_place = ctx->pin_out.dev != nullptr && tab_in_place_enabled();
    int32_t* const tab = tab_in_place ? (int32_t*)(static_cast<char*>(ctx->pin_out.dev) + (oMap - oRes)) : (int32_t*)(d + oMap);
    if ((rc = launch_visible_compact(*K, Twf, (const double*)d_LM, d_cand, n_map, lines, (int32_t*)(d + oQi), res + 1,
                                     tab, nullptr, (uint32_t*)(d + oPart), /* zeroed by the image above */ true, s)))
        return rc;
    if ((rc = launch_prepare_rows(*K, Twf, d_MD, (const double*)d_LM, (const int32_t*)(d + oQi), res + 1, n_map, lines, 0.0, 0.0,
                                  d + oQ, (double*)(d + oQL), nullptr, nullptr, s)))
        return rc;
    rc = match_problems_on_ctx_stream(ctx, &p, 1, res + 1);              // :597 / :712
    if (rc == PLSLAM_ENOTSUP) {
        // (cannot happen after the test at the top -- the plan's own refusals are the context's options --; if it ever does, what
        // is in flight writes scratch only: wait for it and let the caller take the step-by-step form)
        PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
        sg.dismiss();
        return PLSLAM_OK;
    }
    if (rc) return rc;
    if ((rc = launch_gate_n(lines, *K, Twf, (const double*)(d + oQL), (const int32_t*)(d + oM), res + 1, n_map, (const double*)(d + oTF),
                            max_epip, (uint8_t*)(d + oMask), res, (const int32_t*)(d + oQi), (const int32_t*)(d + oTi), tab, s,
                            tab_in_place ? res + 3 : nullptr, res, (int32_t*)ctx->pin_out.dev, 3)))     // (res[3]: zero in the image)
        return rc;
    // ---- the counters (from the gate's last workgroup) and, when it is not there already, the table behind them down; one synchronisation
    if (!tab_in_place) {
        PLSLAM_HIP_CHECK(hipMemcpyAsync(ho, d + oRes, (oMap - oRes) + (size_t)n_map * 4, hipMemcpyDeviceToHost, s));
    }
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    sg.dismiss();
    *done = 1;
    const int32_t* r = reinterpret_cast<const int32_t*>(ho);
    if (!(r[1] > min_matches)) return PLSLAM_OK;                          // match() would not have run: map_to_kf stays -1 everywhere
    memcpy(map_to_kf, ho + (oMap - oRes), (size_t)n_map * 4);
    if (n_matches) *n_matches = r[0];
    if (used_match) *used_match = 1;
    return PLSLAM_OK;
}

// MapHandler::matchKF2KFPoints / matchKF2KFLines, compute part (src/mapHandler.cpp:246-278 / :378-426)
int kf2kf_driver(plslam_ctx* ctx, int lines, const plslam_cam* K, const double* DT, const double* X_prev,
                 const uint8_t* desc_prev, int32_t n_prev, const double* feat_curr, const uint8_t* desc_curr,
                 int32_t n_curr, float nnr, int mutual, int32_t min_matches, const plslam_fast_matching* fm,
                 int32_t* matches_12, int32_t* n_matches, int32_t* used_match, bool rows_dev = false)
{
    // rows_dev: X_prev, desc_prev and desc_curr are DEVICE pointers (the keyframes' descriptors and the previous keyframe's 3D
    // features stay on the GPU between calls; 16-byte aligned rows): nothing is staged or uploaded but the grid of feat_curr
    PLSLAM_REQUIRE(ctx && K && DT && n_prev >= 0 && n_curr >= 0, PLSLAM_EINVAL);
    const bool fast = fm && fm->enabled;
    if (fast) PLSLAM_REQUIRE(fast_ok(fm), PLSLAM_EINVAL);
    if (n_matches) *n_matches = 0;
    if (used_match) *used_match = 0;
    if (n_prev == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(matches_12 != nullptr, PLSLAM_EINVAL);
    for (int32_t i = 0; i < n_prev; ++i) matches_12[i] = -1;
    if (n_curr == 0) return PLSLAM_OK;                                    // :243 / :368
    PLSLAM_REQUIRE(X_prev && desc_prev && feat_curr && desc_curr, PLSLAM_EINVAL);
    if (rows_dev)
        PLSLAM_REQUIRE((reinterpret_cast<uintptr_t>(desc_prev) & 15) == 0 && (reinterpret_cast<uintptr_t>(desc_curr) & 15) == 0 &&
                           (reinterpret_cast<uintptr_t>(X_prev) & 7) == 0,
                       PLSLAM_EINVAL);
    const bool bf_possible = n_curr > min_matches && n_prev > min_matches;
    if (!fast && !(bf_possible && 0 < min_matches)) return PLSLAM_OK;      // no matcher would run
    const int xw = lines ? 6 : 3;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    hipStream_t s = ctx->stream;
    StreamSyncOnError sg(s);
    // ONE page-locked image [X_prev | desc_prev | desc_curr] -> one upload (X only when the windowed matcher runs); the
    // match table comes back through page-locked memory the kernels write (no download on the common path)
    Carve c;
    const size_t oQ = c.take(rows_dev ? 0 : (size_t)n_prev * 32), oT = c.take(rows_dev ? 0 : (size_t)n_curr * 32),
                 oX = c.take(fast && !rows_dev ? (size_t)n_prev * xw * 8 : 0);
    const size_t image = c.off;
    const size_t oM = c.take((size_t)n_prev * 4), oCnt = c.take(16);
    int rc;
    if ((rc = ctx->misc_a.reserve(c.off))) return rc;
    if ((rc = ctx->pin_in.reserve(image))) return rc;
    if ((rc = ctx->pin_out.reserve((size_t)n_prev * 4 + 256))) return rc;
    char* d = ctx->misc_a.as<char>();
    char* h = ctx->pin_in.as<char>();
    if (!rows_dev) {
        memcpy(h + oQ, desc_prev, (size_t)n_prev * 32);
        memcpy(h + oT, desc_curr, (size_t)n_curr * 32);
        if (fast) memcpy(h + oX, X_prev, (size_t)n_prev * xw * 8);
        PLSLAM_HIP_CHECK(hipMemcpyAsync(d, h, image, hipMemcpyHostToDevice, s));
    }
    const uint8_t* const d_Q = rows_dev ? desc_prev : (const uint8_t*)(d + oQ);
    const uint8_t* const d_T = rows_dev ? desc_curr : (const uint8_t*)(d + oT);
    const double* const d_X = rows_dev ? X_prev : (const double*)(d + oX);
    int32_t* tab_host = ctx->pin_out.as<int32_t>();
    int32_t* tab_mapped = static_cast<int32_t*>(ctx->pin_out.dev);
    int32_t* tab_dev = (int32_t*)(d + oM);
    int32_t matches = 0;
    bool have = false, on_host = false;            // on_host: the current table is in tab_host (written by a kernel, synchronised)
    // (lines: when the windowed pass provably stays below min_matches -- kf2kf_line_grid_bound -- StVO::match is enqueued behind it
    // at once: one synchronisation for the call instead of two)
    const bool both_known = fast && lines && bf_possible && !rows_dev && tab_mapped &&
                            kf2kf_line_grid_bound(K, DT, X_prev, n_prev, fm) < min_matches;
    const int32_t* grid_res = nullptr;
    if (fast) {
        // points: pj_points = projection * inv (:256); lines: pj_lines = the projected PIXELS (:392-393, as upstream)
        if ((rc = grid_path(ctx, lines, K, DT, d_X, n_prev, lines ? 1.0 : fm->inv_width,
                            lines ? 1.0 : fm->inv_height, d_Q, feat_curr, nullptr, n_curr,
                            d_T, fm, mutual, tab_mapped ? tab_mapped : tab_dev, &matches, both_known ? &grid_res : nullptr)))
            return rc;
        have = true;
        on_host = tab_mapped != nullptr;           // (both_known: "will be" -- the stream's order puts match() behind the kernel that writes it)
    }
    bool count_entries = false;
    if (bf_possible && (both_known || matches < min_matches)) {            // :274-278 / :421-425
        plslam_match_problem p{};
        p.d1 = d_Q; p.n1 = n_prev; p.d2 = d_T; p.n2 = n_curr;
        p.nnr = nnr; p.mutual = mutual ? 1 : 0; p.n_matches = (int32_t*)(d + oCnt);
        p.keep_prior = have ? 1 : 0;             // the vector matchGrid filled is handed on (:271 -> :277, :418 -> :424)
        bool count_only = false;                 // the table is (stays) in page-locked memory: only the counter comes back
        if (p.keep_prior && on_host && tab_mapped) {
            // the finalize kernel reads the earlier entry of a rejected row and writes every row: a few hundred to a few
            // thousand 4-byte accesses to page-locked memory cost less than moving the table to the device and back (two
            // copy commands on the call's critical path)
            p.matches_12 = tab_mapped;
            count_only = true;
        } else if (p.keep_prior) {
            if (on_host) PLSLAM_HIP_CHECK(hipMemcpyAsync(tab_dev, tab_host, (size_t)n_prev * 4, hipMemcpyHostToDevice, s));
            p.matches_12 = tab_dev;
            on_host = false;
        } else {
            p.matches_12 = tab_mapped ? tab_mapped : tab_dev;
            on_host = tab_mapped != nullptr;
            count_entries = on_host;             // StVO::match on a fresh vector: the count is the number of entries
        }
        if ((rc = match_problems_on_ctx_stream(ctx, &p, 1))) return rc;
        if (!on_host) {
            PLSLAM_HIP_CHECK(hipMemcpyAsync(tab_host, tab_dev, (size_t)n_prev * 4, hipMemcpyDeviceToHost, s));
            PLSLAM_HIP_CHECK(hipMemcpyAsync(&matches, d + oCnt, 4, hipMemcpyDeviceToHost, s));
        } else if (count_only) {
            // (the count: behind the table in the page-locked block, by a one-lane launch -- a small copy command starts ~13 us after
            // the kernel in front of it, a kernel 2-3 us)
            hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(64), 0, s, (const int32_t*)(d + oCnt), tab_mapped + n_prev, 1);
            PLSLAM_HIP_CHECK(hipGetLastError());
        }
        PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
        if (grid_res) PLSLAM_REQUIRE(grid_res[1] == 0 && grid_res[0] >= 0 && grid_res[0] < min_matches, PLSLAM_ERANGE);   // (the bound held)
        if (count_only) matches = tab_host[n_prev];
        on_host = true;
        have = true;
        if (used_match) *used_match = 1;
    } else if (have && !on_host) {
        PLSLAM_HIP_CHECK(hipMemcpyAsync(tab_host, tab_dev, (size_t)n_prev * 4, hipMemcpyDeviceToHost, s));
        PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
        on_host = true;
    }
    sg.dismiss();
    if (!have) return PLSLAM_OK;
    memcpy(matches_12, tab_host, (size_t)n_prev * 4);
    if (count_entries) {
        matches = 0;
        for (int32_t i = 0; i < n_prev; ++i) matches += matches_12[i] >= 0;
    }
    if (n_matches) *n_matches = matches;
    return PLSLAM_OK;
}

int map2kf_driver(plslam_ctx* ctx, int lines, const plslam_cam* K, const double* Twf, const double* LM,
                  const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map, const uint8_t* kf_desc,
                  const double* kf_feat, const double* kf_seg, const int32_t* kf_idx, int32_t n_kf, float nnr,
                  int mutual, double max_epip, int32_t min_matches, const plslam_fast_matching* fm,
                  int32_t* map_to_kf, int32_t* n_matches, int32_t* used_match, bool map_dev = false)
{
    // map_dev: LM, med_desc and candidate are DEVICE pointers (the local map lives on the GPU across keyframes): nothing of
    // the map is staged or uploaded, the candidate flags are folded into the visibility kernel
    PLSLAM_REQUIRE(ctx && K && Twf && n_map >= 0 && n_kf >= 0, PLSLAM_EINVAL);
    const bool fast = fm && fm->enabled;
    if (fast) {
        PLSLAM_REQUIRE(fast_ok(fm), PLSLAM_EINVAL);
        PLSLAM_REQUIRE(!lines || kf_seg || n_kf == 0, PLSLAM_EINVAL);
    }
    if (n_matches) *n_matches = 0;
    if (used_match) *used_match = 0;
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
    DeviceGuard dg_(ctx->device);    // every entry point runs on the context's device, whatever the calling thread's current one
    if (fast) {
        // one launch sequence, one synchronisation (+ a second, short one on what is resident when the brute-force matcher
        // has to replace matchGrid's table); it asks for the step-by-step form below only for a huge map
        int redo = 0;
        const int rc1 = map2kf_fast_once(ctx, lines, K, Twf, LM, med_desc, candidate, n_map, kf_desc, kf_feat, kf_seg, ti, nnr, mutual,
                                         max_epip, min_matches, fm, map_to_kf, n_matches, used_match, map_dev, &redo);
        if (rc1 || !redo) return rc1;
    } else {
        // no matcher would run at all (:594 / :709 with no table from matchGrid: matches = 0 < min_matches is the condition)
        if (!(0 < min_matches)) return PLSLAM_OK;
        int done = 0;
        const int rc1 = map2kf_bf_once(ctx, lines, K, Twf, LM, med_desc, candidate, n_map, kf_desc, kf_feat, ti, nnr, mutual, max_epip,
                                       min_matches, map_to_kf, n_matches, used_match, map_dev, &done);
        if (rc1 || done) return rc1;
    }
    hipStream_t s = ctx->stream;
    StreamSyncOnError sg(s);
    // ---- stage the map and the keyframe on the device (ONE page-locked image, one upload), project + visibility test ----
    Carve c;
    const size_t oLM = c.take(map_dev ? 0 : (size_t)n_map * lw * 8), oMD = c.take(map_dev ? 0 : (size_t)n_map * 32),
                 oKD = c.take((size_t)n_kf * 32), oKF = c.take((size_t)n_kf * fw * 8);
    const size_t image1 = c.off;
    const size_t oQi = c.take((size_t)n_map * 4), oTi = c.take((size_t)nt * 4);          // second image: the two lists
    const size_t image2 = c.off - oQi;
    const size_t oVis = c.take((size_t)n_map), oQ = c.take((size_t)n_map * 32),
                 oT = c.take((size_t)nt * 32), oQL = c.take((size_t)n_map * lw * 8), oTF = c.take((size_t)nt * fw * 8);
    const size_t oM = c.take((size_t)n_map * 4), oMask = c.take((size_t)n_map), oCnt = c.take(8);   // results: one download
    const size_t results = c.off - oM;
    int rc;
    if ((rc = ctx->misc_a.reserve(c.off))) return rc;
    if ((rc = ctx->pin_in.reserve(std::max(image1, image2)))) return rc;
    if ((rc = ctx->pin_out.reserve(std::max(results, (size_t)n_map)))) return rc;
    char* d = ctx->misc_a.as<char>();
    char* h = ctx->pin_in.as<char>();
    char* ho = ctx->pin_out.as<char>();
    if (!map_dev) {
        memcpy(h + oLM, LM, (size_t)n_map * lw * 8);
        memcpy(h + oMD, med_desc, (size_t)n_map * 32);
    }
    const char* const d_LM = map_dev ? reinterpret_cast<const char*>(LM) : d + oLM;          // the map on the device
    const char* const d_MD = map_dev ? reinterpret_cast<const char*>(med_desc) : d + oMD;
    memcpy(h + oKD, kf_desc, (size_t)n_kf * 32);
    memcpy(h + oKF, kf_feat, (size_t)n_kf * fw * 8);
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d, h, image1, hipMemcpyHostToDevice, s));
    // the visibility flags come back through page-locked memory the kernel writes (no download command)
    uint8_t* vis_mapped = static_cast<uint8_t*>(mapped_device_pointer(ho));
    rc = map_dev ? launch_visible_cand(*K, Twf, (const double*)d_LM, candidate, n_map, lines, vis_mapped ? vis_mapped : (uint8_t*)(d + oVis), s)
                 : launch_visible(*K, Twf, (const double*)d_LM, n_map, lines, vis_mapped ? vis_mapped : (uint8_t*)(d + oVis), s);
    if (rc) return rc;
    if (!vis_mapped) PLSLAM_HIP_CHECK(hipMemcpyAsync(ho, d + oVis, (size_t)n_map, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    const uint8_t* vis = reinterpret_cast<const uint8_t*>(ho);

    // ---- Q list: candidate landmarks that project inside the image, :545-558 / :647-663 ------
    std::vector<int32_t> qi;
    for (int32_t i = 0; i < n_map; ++i)
        if ((map_dev || candidate[i]) && vis[i]) qi.push_back(i);        // (map_dev: the kernel folded the flags in)
    const int32_t nq = (int32_t)qi.size();
    if (nq == 0) { sg.dismiss(); return PLSLAM_OK; }                      // :571 / :676

    // ---- build the Q / T matrices on the device, match, gate ----------------------------------
    memcpy(h, qi.data(), (size_t)nq * 4);                                  // (image 1 is on the device: the buffer is free)
    memcpy(h + (oTi - oQi), ti.data(), (size_t)nt * 4);
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oQi, h, image2, hipMemcpyHostToDevice, s));
    if ((rc = launch_gather_rows(d_MD, (int32_t*)(d + oQi), nq, 32, d + oQ, s))) return rc;
    if ((rc = launch_gather_rows(d + oKD, (int32_t*)(d + oTi), nt, 32, d + oT, s))) return rc;
    if ((rc = launch_gather_rows(d_LM, (int32_t*)(d + oQi), nq, lw * 8, d + oQL, s))) return rc;
    if ((rc = launch_gather_rows(d + oKF, (int32_t*)(d + oTi), nt, fw * 8, d + oTF, s))) return rc;
    int32_t matches = 0;
    bool have_m12 = false;                       // matches_12.size() != 0: a matcher ran
    if (fast) {                                  // :578-592 / :681-707
        // the grid of the unmatched keyframe features: points :581-584, lines :686-698 (kf_seg = spl, epl)
        if ((rc = grid_path(ctx, lines, K, Twf, (const double*)(d + oQL), nq, fm->inv_width, fm->inv_height,
                            (const uint8_t*)(d + oQ), lines ? kf_seg : kf_feat, ti.data(), nt, (const uint8_t*)(d + oT),
                            fm, mutual, (int32_t*)(d + oM), &matches)))
            return rc;
        have_m12 = true;
    }
    if (nq > min_matches && matches < min_matches) {                     // :594-598 / :709-713
        plslam_match_problem p{};
        p.d1 = (uint8_t*)(d + oQ); p.n1 = nq; p.d2 = (uint8_t*)(d + oT); p.n2 = nt;
        p.nnr = nnr; p.mutual = mutual ? 1 : 0; p.matches_12 = (int32_t*)(d + oM); p.n_matches = nullptr;
        p.keep_prior = have_m12 ? 1 : 0;         // the vector matchGrid filled is handed on (:591 -> :597, :706 -> :712)
        if ((rc = match_problems_on_ctx_stream(ctx, &p, 1))) return rc;   // :597 / :712
        have_m12 = true;
        if (used_match) *used_match = 1;
    }
    if (!have_m12) { PLSLAM_HIP_CHECK(hipStreamSynchronize(s)); sg.dismiss(); return PLSLAM_OK; }
    rc = lines ? launch_line_gate(*K, Twf, (double*)(d + oQL), (int32_t*)(d + oM), nq, (double*)(d + oTF),
                                  max_epip, (uint8_t*)(d + oMask), (int32_t*)(d + oCnt), s)
               : launch_point_gate(*K, Twf, (double*)(d + oQL), (int32_t*)(d + oM), nq, (double*)(d + oTF),
                                   max_epip, (uint8_t*)(d + oMask), (int32_t*)(d + oCnt), s);
    if (rc) return rc;
    // table, mask and count lie behind one another: one download
    PLSLAM_HIP_CHECK(hipMemcpyAsync(ho, d + oM, results, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    sg.dismiss();
    const int32_t* m12 = reinterpret_cast<const int32_t*>(ho);
    const uint8_t* mask = reinterpret_cast<const uint8_t*>(ho + (oMask - oM));
    for (int32_t a = 0; a < nq; ++a)
        if (mask[a]) map_to_kf[qi[a]] = ti[m12[a]];                       // :614-619 (the association)
    if (n_matches) *n_matches = *reinterpret_cast<const int32_t*>(ho + (oCnt - oM));
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
    return plslam::map2kf_driver(ctx, 0, K, Twf, Xw, med_desc, candidate, n_map, kf_desc, kf_pl, nullptr, kf_idx, n_kf,
                                 nnr, mutual, max_epip, min_matches, nullptr, map_to_kf, n_matches, nullptr);
}

int plslam_map2kf_match_lines(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* Lw,
                              const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                              const uint8_t* kf_desc, const double* kf_le, const int32_t* kf_idx,
                              int32_t n_kf, float nnr, int mutual, double max_epip, int32_t min_matches,
                              int32_t* map_to_kf, int32_t* n_matches)
{
    return plslam::map2kf_driver(ctx, 1, K, Twf, Lw, med_desc, candidate, n_map, kf_desc, kf_le, nullptr, kf_idx, n_kf,
                                 nnr, mutual, max_epip, min_matches, nullptr, map_to_kf, n_matches, nullptr);
}

int plslam_map2kf_match_points_fast(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* Xw,
                                    const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                                    const uint8_t* kf_desc, const double* kf_pl, const int32_t* kf_idx, int32_t n_kf,
                                    float nnr, int mutual, double max_epip, int32_t min_matches,
                                    const plslam_fast_matching* fm, int32_t* map_to_kf, int32_t* n_matches,
                                    int32_t* used_match)
{
    return plslam::map2kf_driver(ctx, 0, K, Twf, Xw, med_desc, candidate, n_map, kf_desc, kf_pl, nullptr, kf_idx, n_kf,
                                 nnr, mutual, max_epip, min_matches, fm, map_to_kf, n_matches, used_match);
}

int plslam_map2kf_match_lines_fast(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* Lw,
                                   const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                                   const uint8_t* kf_desc, const double* kf_le, const double* kf_seg,
                                   const int32_t* kf_idx, int32_t n_kf, float nnr, int mutual, double max_epip,
                                   int32_t min_matches, const plslam_fast_matching* fm, int32_t* map_to_kf,
                                   int32_t* n_matches, int32_t* used_match)
{
    return plslam::map2kf_driver(ctx, 1, K, Twf, Lw, med_desc, candidate, n_map, kf_desc, kf_le, kf_seg, kf_idx, n_kf,
                                 nnr, mutual, max_epip, min_matches, fm, map_to_kf, n_matches, used_match);
}

int plslam_map2kf_match_points_dev(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* d_Xw,
                                   const uint8_t* d_med_desc, const uint8_t* d_candidate, int32_t n_map,
                                   const uint8_t* kf_desc, const double* kf_pl, const int32_t* kf_idx, int32_t n_kf,
                                   float nnr, int mutual, double max_epip, int32_t min_matches,
                                   const plslam_fast_matching* fm, int32_t* map_to_kf, int32_t* n_matches,
                                   int32_t* used_match)
{
    PLSLAM_REQUIRE(n_map == 0 || (((uintptr_t)d_Xw & 7) == 0 && ((uintptr_t)d_med_desc & 7) == 0), PLSLAM_EINVAL);
    return plslam::map2kf_driver(ctx, 0, K, Twf, d_Xw, d_med_desc, d_candidate, n_map, kf_desc, kf_pl, nullptr, kf_idx, n_kf,
                                 nnr, mutual, max_epip, min_matches, fm, map_to_kf, n_matches, used_match, true);
}

int plslam_map2kf_match_lines_dev(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* d_Lw,
                                  const uint8_t* d_med_desc, const uint8_t* d_candidate, int32_t n_map,
                                  const uint8_t* kf_desc, const double* kf_le, const double* kf_seg,
                                  const int32_t* kf_idx, int32_t n_kf, float nnr, int mutual, double max_epip,
                                  int32_t min_matches, const plslam_fast_matching* fm, int32_t* map_to_kf,
                                  int32_t* n_matches, int32_t* used_match)
{
    PLSLAM_REQUIRE(n_map == 0 || (((uintptr_t)d_Lw & 7) == 0 && ((uintptr_t)d_med_desc & 7) == 0), PLSLAM_EINVAL);
    return plslam::map2kf_driver(ctx, 1, K, Twf, d_Lw, d_med_desc, d_candidate, n_map, kf_desc, kf_le, kf_seg, kf_idx, n_kf,
                                 nnr, mutual, max_epip, min_matches, fm, map_to_kf, n_matches, used_match, true);
}

int plslam_kf2kf_match_points(plslam_ctx* ctx, const plslam_cam* K, const double* DT, const double* P_prev,
                              const uint8_t* desc_prev, int32_t n_prev, const double* pl_curr, const uint8_t* desc_curr,
                              int32_t n_curr, float nnr, int mutual, int32_t min_matches, const plslam_fast_matching* fm,
                              int32_t* matches_12, int32_t* n_matches, int32_t* used_match)
{
    return plslam::kf2kf_driver(ctx, 0, K, DT, P_prev, desc_prev, n_prev, pl_curr, desc_curr, n_curr, nnr, mutual,
                                min_matches, fm, matches_12, n_matches, used_match);
}

int plslam_kf2kf_match_lines(plslam_ctx* ctx, const plslam_cam* K, const double* DT, const double* sPeP_prev,
                             const uint8_t* desc_prev, int32_t n_prev, const double* seg_curr, const uint8_t* desc_curr,
                             int32_t n_curr, float nnr, int mutual, int32_t min_matches, const plslam_fast_matching* fm,
                             int32_t* matches_12, int32_t* n_matches, int32_t* used_match)
{
    return plslam::kf2kf_driver(ctx, 1, K, DT, sPeP_prev, desc_prev, n_prev, seg_curr, desc_curr, n_curr, nnr, mutual,
                                min_matches, fm, matches_12, n_matches, used_match);
}

int plslam_kf2kf_match_points_dev(plslam_ctx* ctx, const plslam_cam* K, const double* DT, const double* d_P_prev,
                                  const uint8_t* d_desc_prev, int32_t n_prev, const double* pl_curr, const uint8_t* d_desc_curr,
                                  int32_t n_curr, float nnr, int mutual, int32_t min_matches, const plslam_fast_matching* fm,
                                  int32_t* matches_12, int32_t* n_matches, int32_t* used_match)
{
    return plslam::kf2kf_driver(ctx, 0, K, DT, d_P_prev, d_desc_prev, n_prev, pl_curr, d_desc_curr, n_curr, nnr, mutual,
                                min_matches, fm, matches_12, n_matches, used_match, true);
}

int plslam_kf2kf_match_lines_dev(plslam_ctx* ctx, const plslam_cam* K, const double* DT, const double* d_sPeP_prev,
                                 const uint8_t* d_desc_prev, int32_t n_prev, const double* seg_curr, const uint8_t* d_desc_curr,
                                 int32_t n_curr, float nnr, int mutual, int32_t min_matches, const plslam_fast_matching* fm,
                                 int32_t* matches_12, int32_t* n_matches, int32_t* used_match)
{
    return plslam::kf2kf_driver(ctx, 1, K, DT, d_sPeP_prev, d_desc_prev, n_prev, seg_curr, d_desc_curr, n_curr, nnr, mutual,
                                min_matches, fm, matches_12, n_matches, used_match, true);
}

}  // extern "C"
