/*
 * plslam_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the PL-SLAM stereo point+line matching hot path and
 * of the local-BA residual/Jacobian row build.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this.  The product library
 * (plslam_amd/csrc -> libplslam_hip.so) never includes, links or falls back to it.
 *
 * PARITY STATUS ("parity unpinned" for the matcher semantics):
 *   - Hamming-256 distance: PINNED against the reference's own in-tree code
 *     (3rdparty/line_descriptor/src/bitops_custom.hpp:83-96 compiled from where it
 *     lies into oracle/_ref/, and the SWAR form of 3rdparty/DBoW2/src/DBoW2/FORB.cpp:78-101).
 *   - The two nearest DISTANCES of every query: PINNED against the only kNN code that exists in the
 *     reference tree, the exact multi-index-hashing search BinaryDescriptorMatcher::knnMatch
 *     (3rdparty/line_descriptor/src/binary_descriptor_matcher.cpp:258-335, compiled from where it lies
 *     into oracle/_ref/ against the cv:: stand-in oracle/ref_shim/; outputs committed as
 *     tests/golden/ref_knn_golden.npz).  Its choice among equally distant rows is not repeatable
 *     between runs, so indices are compared only where distance alone decides.
 *   - kNN-2 TIE order, ratio test, mutual check: the arithmetic lives in OpenCV 3.x
 *     features2d (cv::BFMatcher::knnMatch) and in the un-vendored stvo-pl
 *     (matching.cpp: match()/matchNNR()) -- neither is under /root/reference, neither
 *     has a pinned version, and the reference holds no tests/golden vectors for it.
 *     Restated here from the published algorithm; anchored on the reference's call
 *     sites src/mapHandler.cpp:277,424,597,712,3223,3249 and the contract visible at
 *     src/mapHandler.cpp:280-283 (matches_12[i1] = i2 or -1).  => parity UNPINNED.
 *   - LBA rows: literal restatement of in-tree code src/mapHandler.cpp:1358-1540
 *     (first pass) and :1587-1772 (iteration pass); the external helpers it calls
 *     (cam->projection, inverse_se3, robustWeightCauchy from stvo-pl) are restated
 *     from the published stvo-pl sources and cross-checked by finite differences.
 */
#ifndef PLSLAM_ORACLE_H
#define PLSLAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLO_DESC_BYTES 32

/* ---- distance --------------------------------------------------------------------- */
/* popcount(a XOR b) over 32 bytes; 4x u32 builtin popcount per 16 B exactly as
 * 3rdparty/line_descriptor/src/bitops_custom.hpp:83-96 */
int plo_hamming256(const uint8_t* a, const uint8_t* b);
/* same value via the SWAR bit-trick on 4x u64 of 3rdparty/DBoW2/src/DBoW2/FORB.cpp:78-101 */
int plo_hamming256_swar(const uint8_t* a, const uint8_t* b);
/* byte-LUT form (lookup[] table, bitops_custom.hpp:58-76,93-94 tail loop) */
int plo_hamming256_lut(const uint8_t* a, const uint8_t* b);

/* ---- kNN-2 (cv::BFMatcher(NORM_HAMMING,false).knnMatch(q,t,matches,2)) ------------ */
/* idx/dist are nq*2; absent neighbours are idx=-1, dist=INT32_MAX.
 * Scan j ascending, K=2 slots, insert on strict '<' => ties keep the lowest trainIdx. */
void plo_knn2(const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt,
              int32_t* idx, int32_t* dist);

/* ---- stvo-pl matchNNR / match ------------------------------------------------------ */
/* ratio test in fp32: accept iff (float)d0 < (float)d1 * nnr. nt<2 => no match (defined
 * divergence from upstream UB). returns #matches; m12 has nq entries. */
int32_t plo_match_nnr(const uint8_t* q, int32_t nq, const uint8_t* t, int32_t nt,
                      float nnr, int32_t* m12);
/* mutual!=0: run both directions and keep i1->i2 iff m21[i2]==i1 (best_lr_matches). */
int32_t plo_match(const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2,
                  float nnr, int mutual, int32_t* m12);
/* the same on a matches_12 that already holds n1 entries: rows the ratio test rejects KEEP theirs ([RECALL] stvo-pl
 * matchNNR: `matches_12.resize(desc1.rows, -1)`), the mutual loop then runs over every entry >= 0 and the returned count
 * is (rows accepted) - (entries cleared) -- the fall-back of src/mapHandler.cpp:274-278, :421-425, :594-598, :709-713 */
int32_t plo_match_prior(const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2,
                        float nnr, int mutual, int32_t* m12);
/* B independent problems described by row offsets (B+1 entries each). */
void plo_match_batched(const uint8_t* d1, const int32_t* off1, const uint8_t* d2,
                       const int32_t* off2, int32_t B, float nnr, int mutual,
                       int32_t* m12, int32_t* n_matches);
/* same, sharded over nthreads pthreads (cpu_baseline "all cores" leg) */
void plo_match_batched_mt(const uint8_t* d1, const int32_t* off1, const uint8_t* d2,
                          const int32_t* off2, int32_t B, float nnr, int mutual,
                          int32_t* m12, int32_t* n_matches, int nthreads);

/* ---- representative descriptor (src/mapFeatures.cpp:51-93, :121-163) ---------------- */
/* returns the index of the observation whose median Hamming distance to all
 * observations (itself included, d=0) is smallest; first minimum wins. */
int32_t plo_median_desc(const uint8_t* descs, int32_t n);
/* the same for every landmark of a map: lists concatenated, landmark l = rows off[l]..off[l+1]-1.
 * med_idx[l] = plo_median_desc of its list (-1 for an empty list), med_desc (optional) = that row. */
void plo_median_desc_batched(const uint8_t* descs, const int32_t* off, int32_t n_lm, int32_t* med_idx,
                             uint8_t* med_desc);

/* ---- SE(3) helpers (stvo-pl auxiliar.cpp, [RECALL]) -------------------------------- */
void plo_inverse_se3(const double T[16], double Tinv[16]);      /* [R^T, -R^T t] */
void plo_expmap_se3(const double x[6], double T[16]);           /* x = [t, w]    */
void plo_logmap_se3(const double T[16], double x[6]);

typedef struct { double fx, fy, cx, cy, b; int32_t width, height; } plo_cam;

/* ---- LBA rows ---------------------------------------------------------------------- */
/* Point rows: src/mapHandler.cpp:1358-1431 (first pass) == :1587-1666 (iteration pass;
 * same algebra, pose/landmark source differs and is the caller's business).
 * T_kf_w: nkf*16 row-major KF->world; Xw: npt*3; obs_uv: nobs*2;
 * lm_loc[nobs] indexes Xw, kf_slot[nobs] indexes T_kf_w.
 * out: J_pose nobs*6, J_lm nobs*3, r nobs, w nobs. */
void plo_lba_point_rows(const plo_cam* K, double homog_th, const double* T_kf_w,
                        const double* Xw, const double* obs_uv, const int32_t* lm_loc,
                        const int32_t* kf_slot, int32_t nobs, double* J_pose, double* J_lm,
                        double* r, double* w);
/* Line rows: first pass src/mapHandler.cpp:1436-1516.  compat_iter_pass!=0 reproduces the
 * iteration-pass quirks of :1668-1748 (P and Q both read Lw3[3*lm_loc..], i.e. the
 * landmark array is addressed with stride 3, and th is the literal 1e-7).
 * Lw: nls*6 (first pass) -- with compat_iter_pass the same buffer is read at stride 3. */
void plo_lba_line_rows(const plo_cam* K, double homog_th, int compat_iter_pass,
                       const double* T_kf_w, const double* Lw, const double* l_obs,
                       const int32_t* lm_loc, const int32_t* kf_slot, int32_t nobs,
                       double* J_pose, double* J_lm, double* r, double* w);

/* Dense accumulation of H (N*N row-major), g (N), err exactly as :1410-1429 / :1519-1538.
 * kf_loc[nobs] = local KF slot in X or -1 (not optimised). N = 6*nkf_opt + 3*npt + 6*nls. */
void plo_lba_accumulate_points(int32_t nkf_opt, int32_t npt, int32_t nls, const int32_t* lm_loc,
                               const int32_t* kf_loc, int32_t nobs, const double* J_pose,
                               const double* J_lm, const double* r, const double* w,
                               double* H, double* g, double* err);
void plo_lba_accumulate_lines(int32_t nkf_opt, int32_t npt, int32_t nls, const int32_t* lm_loc,
                              const int32_t* kf_loc, int32_t nobs, const double* J_pose,
                              const double* J_lm, const double* r, const double* w,
                              double* H, double* g, double* err);

/* ---- pose-only GN system of the loop-closure relative pose ---------------------------------------------
 * MapHandler::computeRelativePoseGN iteration body, src/mapHandler.cpp:3324-3424 (== computeRelativePoseRobustGN
 * :3588-3689): literal restatement; H (36, row-major) = H_p + H_l, g = g_p + g_l, *e = e_p + e_l (not yet divided by
 * N_l + N_p); n_obs[0] = N_p, n_obs[1] = N_l. */
void plo_pose_gn_accumulate(const plo_cam* K, double homog_th, const double T_inc[16], const double* P,
                            const double* pl_obs, const uint8_t* pt_inlier, int32_t npt, const double* sPeP,
                            const double* le_obs, const uint8_t* ls_inlier, int32_t nls, double* H, double* g, double* e,
                            int32_t* n_obs);

/* ---- map<->KF geometric gates (inlier masks) ---------------------------------------- */
/* src/mapHandler.cpp:605-613: mask[i]=1 iff m12[i]>=0 and ||proj(Twf*X_i) - pl[m12[i]]|| < th.
 * returns #inliers. Twf row-major 4x4; Xw nq*3; pl nt*2. */
int32_t plo_map2kf_point_gate(const plo_cam* K, const double Twf[16], const double* Xw,
                              const int32_t* m12, int32_t nq, const double* pl, double max_epip,
                              uint8_t* mask);
/* src/mapHandler.cpp:720-729: signed test, both endpoints. Lw nq*6; le nt*3. */
int32_t plo_map2kf_line_gate(const plo_cam* K, const double Twf[16], const double* Lw,
                             const int32_t* m12, int32_t nq, const double* le, double max_epip,
                             uint8_t* mask);
/* candidate pre-filter src/mapHandler.cpp:549-551 (points) / :650-655 (lines): 1 iff the
 * landmark projects strictly inside the image with positive depth. */
void plo_map_point_visible(const plo_cam* K, const double Twf[16], const double* Xw, int32_t n,
                           uint8_t* vis);
void plo_map_line_visible(const plo_cam* K, const double Twf[16], const double* Lw, int32_t n,
                          uint8_t* vis);

/* ---- map <-> keyframe drivers (BF path, fast_matching == false) -------------------------- */
/* MapHandler::matchMap2KFPoints, src/mapHandler.cpp:532-632 without the map mutation:
 *  Q = med_desc rows of candidate[i] landmarks that project inside the image (:545-558);
 *  T = kf_desc rows whose kf_idx == -1 (:563-569); if either is empty -> 0 (:571);
 *  match() only if |Q| > min_matches (:594-597, matchGrid absent => matches == 0 before);
 *  accept i1->i2 iff || proj(Twf X) - pl || < max_epip (:610-613), else --matches (:628).
 * map_to_kf[n_map] receives the ORIGINAL kf feature index or -1.  Returns #accepted. */
int32_t plo_map2kf_match_points(const plo_cam* K, const double Twf[16], const double* Xw,
                                const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                                const uint8_t* kf_desc, const double* kf_pl, const int32_t* kf_idx,
                                int32_t n_kf, float nnr, int mutual, double max_epip,
                                int32_t min_matches, int32_t* map_to_kf);
/* MapHandler::matchMap2KFLines, src/mapHandler.cpp:634-752 (both endpoints visible :654-655,
 * signed gate :727-729, min_matches = SlamConfig::minLineMatches :709-712). */
int32_t plo_map2kf_match_lines(const plo_cam* K, const double Twf[16], const double* Lw,
                               const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                               const uint8_t* kf_desc, const double* kf_le, const int32_t* kf_idx,
                               int32_t n_kf, float nnr, int mutual, double max_epip,
                               int32_t min_matches, int32_t* map_to_kf);

/* ---- the same drivers with SlamConfig::fastMatching() (the shipped configurations) ----------
 * src/mapHandler.cpp:578-592 (points) / :681-707 (lines): project the candidates into the keyframe
 * (pj_points / pj_lines in grid units, truncated to int by make_pair<int,int>), fill a GridStructure with
 * the unmatched keyframe features (points: their cell; lines: their Bresenham cells + directions), window
 * of matching_f2f_ws cells, matchGrid (ratio Config::minRatio12P() for BOTH kinds); then, as in the plain
 * drivers, StVO::match replaces the result when `|Q| > min_matches && matches < min_matches` (:594-598,
 * :709-713; lines use minRatio12L there).  *used_match (may be NULL) tells whether that happened.
 * kf_seg: n_kf x 4 = (spl, epl) pixel end points of the keyframe lines (only lines, only when enabled). */
typedef struct {
    int32_t enabled;              /* SlamConfig::fastMatching() */
    int32_t grid_cols, grid_rows; /* GRID_COLS, GRID_ROWS */
    int32_t ws;                   /* SlamConfig::matchingF2FWs() */
    double inv_width, inv_height; /* StereoFrame::inv_width / inv_height = GRID_COLS / width, GRID_ROWS / height */
    double nnr_grid;              /* Config::minRatio12P() */
    double line_sim_th;           /* Config::lineSimTh() */
} plo_fast_matching;
int32_t plo_map2kf_match_points_fast(const plo_cam* K, const double Twf[16], const double* Xw,
                                     const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                                     const uint8_t* kf_desc, const double* kf_pl, const int32_t* kf_idx,
                                     int32_t n_kf, float nnr, int mutual, double max_epip, int32_t min_matches,
                                     const plo_fast_matching* fm, int32_t* map_to_kf, int32_t* used_match);
int32_t plo_map2kf_match_lines_fast(const plo_cam* K, const double Twf[16], const double* Lw,
                                    const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                                    const uint8_t* kf_desc, const double* kf_le, const double* kf_seg,
                                    const int32_t* kf_idx, int32_t n_kf, float nnr, int mutual, double max_epip,
                                    int32_t min_matches, const plo_fast_matching* fm, int32_t* map_to_kf,
                                    int32_t* used_match);

/* ---- keyframe <-> keyframe drivers: MapHandler::matchKF2KFPoints / matchKF2KFLines ----------------
 * src/mapHandler.cpp:234-363 / :365-530, compute part (:246-278 / :378-426); creating MapPoints / MapLines from the
 * table (:280-363) stays with the caller.
 *   if fastMatching: project the previous keyframe's stereo features with DT (:254-256 / :384-394), fill the grid with
 *       the current keyframe's features (:259-263 / :397-411), window of matchingF2FWs cells, matchGrid;
 *       points: pj_points = projection * inv_width / inv_height; lines: pj_lines = the projections in PIXELS, not
 *       multiplied by inv_width (:392-393, as written upstream), both truncated to int by make_pair<int,int>
 *       (a projection that is not a finite int32 becomes INT_MIN, what x86's cvttsd2si returns);
 *   if n_curr > min_matches && n_prev > min_matches && matches < min_matches: matches = match(prev, curr, nnr)
 *       (:274-278 / :421-425).
 * m12: n_prev entries (all -1 when no matcher ran).  Returns matches.  seg_curr: n_curr x 4 (spl, epl). */
int32_t plo_kf2kf_match_points(const plo_cam* K, const double DT[16], const double* P_prev, const uint8_t* desc_prev,
                               int32_t n_prev, const double* pl_curr, const uint8_t* desc_curr, int32_t n_curr,
                               float nnr, int mutual, int32_t min_matches, const plo_fast_matching* fm, int32_t* m12,
                               int32_t* used_match);
int32_t plo_kf2kf_match_lines(const plo_cam* K, const double DT[16], const double* sPeP_prev,
                              const uint8_t* desc_prev, int32_t n_prev, const double* seg_curr,
                              const uint8_t* desc_curr, int32_t n_curr, float nnr, int mutual, int32_t min_matches,
                              const plo_fast_matching* fm, int32_t* m12, int32_t* used_match);

/* ---- stereo L<->R gates inside StVO::StereoFrame (stvo-pl stereoFrame.cpp, [RECALL]) -------
 * Thresholds are the reference's own config keys: max_dist_epip (config/config/config_kitti.yaml:25), min_disp (:26),
 * stereo_overlap_th (:31), line_horiz_th (:34), ls_min_disp_ratio (:36).  matches_12 is the table StVO::match /
 * matchGrid produced for (desc_l, desc_r).
 * Points (matchStereoPoints): keep i1 -> i2 iff  std::abs(pt_l.y - pt_r.y) <= max_dist_epip  (float arithmetic on
 * cv::KeyPoint::pt, compared against the double threshold)  and  disp = pt_l.x - pt_r.x >= min_disp  (float
 * subtraction, then double).  kp: n x 2 float32 (pt.x, pt.y).  stereo_12[i1] = i2 or -1, disp[i1] (0 when dropped).
 * Lines (matchStereoLines): end points as doubles; overlap = lineSegmentOverlapStereo(sp_l.y, ep_l.y, sp_r.y, ep_r.y);
 * the right end points are moved along the right line to the rows of the left end points (the second one already
 * reads the overwritten first one, as upstream); disp_s / disp_e = differences of x, both set to -1 when
 * min/max < ls_min_disp_ratio (filterLineSegmentDisparity); keep iff both >= min_disp, |sp_l.y - ep_l.y| >
 * line_horiz_th, |sp_r'.y - ep_r'.y| > line_horiz_th, overlap > stereo_overlap_th.
 * seg: n x 4 float32 (startPointX, startPointY, endPointX, endPointY of the cv::line_descriptor::KeyLine).
 * disp_se: n_l x 2 (disp_s, disp_e; 0, 0 when dropped).  Both return the number of stereo features. */
int32_t plo_stereo_point_gate(const int32_t* m12, int32_t n_l, const float* kp_l, const float* kp_r, int32_t n_r,
                              double max_dist_epip, double min_disp, int32_t* stereo_12, double* disp);
int32_t plo_stereo_line_gate(const int32_t* m12, int32_t n_l, const float* seg_l, const float* seg_r, int32_t n_r,
                             double min_disp, double line_horiz_th, double stereo_overlap_th,
                             double ls_min_disp_ratio, int32_t* stereo_12, double* disp_se);
double plo_line_segment_overlap_stereo(double spl_obs, double epl_obs, double spl_proj, double epl_proj,
                                       double line_horiz_th);

/* ---- stvo-pl matchGrid: the windowed ("fast_matching") matcher ------------------------------
 * Call sites in the reference: src/mapHandler.cpp:271 (points, KF<->KF), :418 (lines), :591 (map points
 * <-> KF), :706 (map lines <-> KF); the grid is filled by the callers at :258-264, :395-411, :580-584,
 * :683-699 and the window comes from SlamConfig::matchingF2FWs() (:266-269).  The function itself lives
 * in the un-vendored stvo-pl (matching.cpp, gridStructure.cpp; no pinned version) => [RECALL], parity
 * UNPINNED.  Restated literally from the published source:
 *   for i1 = 0 .. n1-1 (in order):
 *     candidates = set union over the query's window centres c of grid.get(c.x, c.y, w)
 *         get(): x_ in [max(0, x - w.width.first), min(cols, x + w.width.second + 1)),
 *                y_ in [max(0, y - w.height.first), min(rows, y + w.height.second + 1)): all of grid[x_][y_]
 *     for i2 in candidates:   skip if i2 < 0 || i2 >= n2
 *         lines only: skip if |dot(dir1[i1], dir2[i2])| < sim_th          (Config::lineSimTh())
 *         d = hamming(desc1[i1], desc2[i2])
 *         if mutual (Config::bestLRMatches()):  if d < distances[i2] { distances[i2] = d; matches_21[i2] = i1 }
 *                                               else skip this candidate
 *         if d < best_d { best_d2 = best_d; best_d = d; best_idx = i2 } else if d < best_d2 { best_d2 = d }
 *     if best_d < best_d2 * nnr (int * double, best_d2 = INT_MAX when absent): matches_12[i1] = best_idx
 *   if mutual: drop i1 -> i2 when matches_21[i2] != i1.
 * Points have one centre (points1[i1], a point_2d = pair<int,int>), lines two (start and end point of
 * lines1[i1]).  Upstream iterates a std::unordered_set<int>: the visiting order -- and with it WHICH of
 * several equally distant best candidates becomes best_idx -- is implementation-defined.  This
 * restatement (and the device path) visits candidates in ascending index order: lowest i2 wins a tie,
 * the same rule as the brute-force matcher.  Nothing else depends on the order.
 * Grid in CSR form: cell (x, y), 0 <= x < cols, 0 <= y < rows, has id x*rows + y and owns
 * cell_items[cell_start[id] .. cell_start[id+1]-1] (the list's push_back order).
 * centres: n1 * n_centres * 2 int32 (x, y).  dir1 (n1*2) / dir2 (n2*2) may be NULL (points).
 * w = {width.first, width.second, height.first, height.second}.  Returns #matches. */
int32_t plo_match_grid(const int32_t* centres, int32_t n_centres, const uint8_t* d1, int32_t n1,
                       const int32_t* cell_start, const int32_t* cell_items, int32_t cols, int32_t rows,
                       const uint8_t* d2, int32_t n2, const double* dir1, const double* dir2,
                       double sim_th, const int32_t w[4], double nnr, int mutual, int32_t* m12);
/* GridStructure fill as the reference's callers do it: grid.at(x, y).push_back(idx) for idx ascending;
 * at() with an out-of-range cell appends to a dummy list (the item is never returned by get()).
 * xy: n*2 int32 cells.  cell_start: cols*rows+1, cell_items: n (entries past cell_start[cols*rows] unused). */
void plo_grid_fill_points(const int32_t* xy, int32_t n, int32_t cols, int32_t rows, int32_t* cell_start,
                          int32_t* cell_items);
/* getLineCoords (stvo-pl gridStructure.cpp, [RECALL]): Bresenham cells of the segment (x1,y1)-(x2,y2) given
 * in (real-valued) grid units; writes up to `cap` (x, y) pairs, returns the number of cells. */
int32_t plo_get_line_coords(double x1, double y1, double x2, double y2, int32_t* out_xy, int32_t cap);
/* normalize(std::pair<double,double>&) of stvo-pl ([RECALL]): v /= sqrt(v.v); a zero vector becomes NaN,
 * and |dot| < th is then false, i.e. the direction test passes. */
void plo_normalize2(double v[2]);

/* ---- LBD float descriptor of a line: BinaryDescriptor::computeLBD ------------------------------------------
 * 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:1026-1372, literal restatement (fp32, the source's
 * sequential summation order; built without FMA contraction): the line support region of NUM_OF_BANDS (9) bands x
 * widthOfBand rows, row sums of the gradient projected on the line direction / its normal split by sign (:1140-1187),
 * the global Gaussian weight per row gaussCoefG_ (:1188-1196), band sums with the local weights gaussCoefL_ of the row's
 * own band and its two neighbours (:1201-1239), mean / std per band (:1253-1277), the two-part normalisation
 * (:1279-1312), the 0.4 clamp (:1318-1325) and the re-normalisation (:1327-1338).  Weight tables as the constructor
 * builds them (:146-176, integer divisions included).
 * PINNED: oracle/_ref holds the reference's own computeLBD / constructor / binaryConversion compiled from where the
 * file lies (oracle/ref_wrap_lbd.cpp); tests/test_oracle_pin.py::test_lbd_compute_pinned_to_reference_code requires
 * every output bit to agree (cos / sin of the direction are the FLOAT overloads, as the reference binds them).
 *   dx, dy      gradient images of ONE octave, int16, `width` x `height`, row stride = width (dxImg_vector[octave])
 *   lines       n x plo_lbd_line: the OctaveSingleLine fields computeLBD reads
 *   lbd         n x 72 float32 (the `descriptor` vector of each line) */
typedef struct {
    int32_t num_pixels;          /* numOfPixels: length of the support region */
    float sx, sy, ex, ey;        /* sPointInOctaveX/Y, ePointInOctaveX/Y */
    float direction;             /* angle of the line */
} plo_lbd_line;
void plo_lbd_compute(const int16_t* dx, const int16_t* dy, int32_t width, int32_t height, const plo_lbd_line* lines,
                     int32_t n, int32_t width_of_band, float* lbd);
void plo_lbd_gauss_tables(int32_t width_of_band, double* coef_l /* 3 w */, double* coef_g /* 9 w */);

/* ---- LBD float -> binary line descriptor ---------------------------------------------------
 * 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp: the 32 band pairs of
 * combinations[32][2] (:74-107), binaryConversion (:401-412; bit i set iff f1[i] > f2[i]) and the
 * row fill loop of computeImpl (:653-668).  lbd: n x 72 f32, desc: n x 32 u8.
 * Pinned: tests/test_oracle_pin.py parses the pair table out of the reference source text. */
#define PLO_LBD_FLOATS 72
extern const int plo_lbd_pairs[32][2];
uint8_t plo_lbd_binary_conversion(const float* f1, const float* f2);
void plo_lbd_binarise(const float* lbd, int32_t n, uint8_t* desc);

#ifdef __cplusplus
}
#endif
#endif
