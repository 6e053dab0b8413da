/*
 * plslam_hip.h -- C ABI of the MI355X (gfx950) implementation of PL-SLAM's stereo
 * point+line matching front end and local-BA row build.
 *
 * This is the drop-in boundary: every entry point is extern "C", takes plain pointers and
 * sizes, never throws, and returns PLSLAM_OK (0) or a negative PLSLAM_E* code.  Each one
 * cites the reference interface it replaces (paths relative to the pl-slam repository).
 *
 * Descriptor matrices are N x 32 uint8, row-major, contiguous -- the layout of the
 * reference's cv::Mat descriptor blocks (ORB: DBoW2 FORB::L = 32,
 * 3rdparty/DBoW2/include/DBoW2/FORB.h:31; LBD: cv::Mat(n,32,CV_8UC1),
 * 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:639) and of the matrices the
 * map-level drivers assemble with Mat::push_back (src/mapHandler.cpp:555,567,660,672).
 * Rows must start on 4-byte boundaries (any cv::Mat / hipMalloc / torch allocation does).
 *
 * Pointer conventions: entry points WITHOUT a _dev suffix take HOST pointers and do their
 * own H2D/D2H on the context's stream; *_dev entry points and match plans take DEVICE
 * pointers and a hipStream_t (passed as void*) and are asynchronous with respect to the host.
 * A NULL stream means "the context's own (non-blocking) stream", NOT HIP's legacy default stream:
 * callers that need ordering against their own work must pass a real stream handle.
 *
 * Threading: the reference calls StVO::match() concurrently from the VO thread, the local
 * mapping thread and the loop-closure thread (app/plslam_dataset.cpp:127,
 * src/mapHandler.cpp:1103, :1164 -> :3223).  Host-pointer entry points serialise on the
 * context; use one context per thread for concurrency.  Plans are immutable after
 * creation; plslam_match_plan_run may be issued from any one thread at a time per plan.
 *
 * There is NO CPU fallback: without a HIP device plslam_ctx_create fails with
 * PLSLAM_ENODEV and nothing else can be called.
 */
#ifndef PLSLAM_HIP_H
#define PLSLAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: plslam_match_problem grew (keep_prior, reserved: 56 bytes), plslam_lba_plan_iterate's flags became a bit mask, options
 * "mfma_form" 3/4 and "exact_second"; 3 (round 4): "mfma_form" 5 (the default), "post_fuse", plslam_match_plan_key_state;
 * 4 (round 5): plslam_match_plan_set_wire16, the Schur step, plslam_lba_plan_host_state; 5 (round 6): plslam_lba_plan_get_landmarks, plslam_lba_plan_iterate_schur / _apply_step, plslam_lba_point_rows_dev_n / _line_rows_dev_n, plslam_match_plan_step_gather / _gather_sync, plslam_rccl_use / _rccl_available.
 * Clients compare plslam_abi_version() with the value they were compiled against. */
#define PLSLAM_ABI_VERSION 5
#define PLSLAM_DESC_BYTES 32
/* largest train set of one directed scan: the composite (distance,index) key keeps 23
 * index bits beside the 9 distance bits */
#define PLSLAM_MAX_TRAIN_ROWS (1 << 23)

enum {
    PLSLAM_OK = 0,
    PLSLAM_EINVAL = -1,  /* bad argument (NULL pointer, negative size, misaligned rows)   */
    PLSLAM_ENODEV = -2,  /* no usable HIP device / wrong architecture                     */
    PLSLAM_EHIP = -3,    /* a HIP runtime call failed; see plslam_last_error()            */
    PLSLAM_ENOMEM = -4,  /* device or host allocation failed                              */
    PLSLAM_ERANGE = -5,  /* size beyond a documented limit (PLSLAM_MAX_TRAIN_ROWS)        */
    PLSLAM_ENOTSUP = -6  /* optional component unavailable (e.g. RCCL not loadable)       */
};

/* kernel variants of the directed Hamming scan (plslam_ctx_set_option "scan_variant") */
enum {
    PLSLAM_SCAN_AUTO = 0,
    PLSLAM_SCAN_LANE_PER_QUERY = 1, /* query in VGPRs, train rows streamed as SGPRs        */
    PLSLAM_SCAN_WAVE_PER_QUERY = 2, /* train tile in LDS, query in SGPRs, wave best-2 reduce */
    PLSLAM_SCAN_SYMMETRIC = 3,      /* mutual problems: one distance serves both directions */
    PLSLAM_SCAN_MFMA = 4            /* symmetric scan with the distances from the matrix cores:
                                       d = (256 - <s(a),s(b)>)/2 over +-1 fp4 codes, exact; non-mutual
                                       problems take its directed form                               */
};

typedef struct plslam_ctx plslam_ctx;
typedef struct plslam_match_plan plslam_match_plan;

/* Pinhole stereo camera: the fields of stvo-pl's PinholeStereoCamera that the reference
 * reads through cam->projection / getFx / getFy / getWidth / getHeight / getB
 * (src/mapHandler.cpp:255,550-551,1374,1384-1385). */
typedef struct plslam_cam {
    double fx, fy, cx, cy, b;
    int32_t width, height;
} plslam_cam;

const char* plslam_strerror(int code);
/* text of the last failure on the calling thread (HIP error string, file:line) */
const char* plslam_last_error(void);
int plslam_abi_version(void);

/* ---- context ------------------------------------------------------------------------ */
/* Binds to HIP device `device_ordinal` (must be gfx950), creates the context stream and
 * scratch pools.  Replaces nothing in the reference (it has no device state). */
int plslam_ctx_create(int device_ordinal, plslam_ctx** out);
void plslam_ctx_destroy(plslam_ctx* ctx);
/* options: "scan_variant" (PLSLAM_SCAN_*), "scan_block" (queries per workgroup of the directed
 * scan: 256|512|1024), "sym_rows" (rows of d1 per lane in the symmetric scan: 0 = auto (default) | 1 | 4),
 * "group_cap" (workgroups of one problem co-scheduled on one XCD: 0 = auto (default) | 1..64),
 * "mfma_form" (bookkeeping of the matrix-core scan: 0 = auto (default; = 5) | 1 = best-2 push per tile (K1e) |
 * 2 = group minima in the row direction + second best by recomputation (K1f) | 3 = two directed scans per mutual problem
 * (K1g) | 4 = group minima in both directions, class-major layouts (K1h) | 5 = K1h with the two M-tiles of a wave
 * software-pipelined against each other (K1i); identical match tables),
 * "post_fuse" (K1h / K1i plans: the stages behind the scan -- merge of the column partials, ratio test + mutual check, stereo
 * gates -- as ONE kernel with a workgroup per problem and the merged column keys in LDS: 0 = auto (default; = never: measured
 * slower than the separate kernels on MI355X although it moves 0.8 GB less per 4096-pair step) | 1 = never | 2 = whenever the
 * plan is eligible (every problem mutual, at most 4096 columns and 4096 rows); identical tables),
 * "exact_second" (K1h: 0 (default) = the index of a row's SECOND neighbour in the internal key tables is exact only where
 * it is an output (plslam_knn2_hamming256) and the column keys are completed lazily by the finalize stage | 1 = every key
 * of plslam_match_plan_dump is exact; match tables are identical either way),
 * "post_workgroups" (0 = default: one workgroup per block-table entry; n > 0: in a split run (plslam_match_plan_run_split) the
 * stages behind the scan -- K1h's merge of the column partials, the finalize kernel -- run as at most n workgroups that walk
 * their block tables, i.e. they hold a bounded number of workgroup slots beside the next scan; measured neutral to +1 %),
 * "split_post" (column-split plans of mutual problems -- a few LARGE problems, e.g. one local map against one frame: 0 = auto
 * (default): everything behind the scan is ONE kernel, which merges the column partials and decides the matches from the
 * column side -- two launches per run | 1 = never: merge kernel + finalize kernel; identical tables and counts),
 * "post_xcd" (how the finalize kernel's workgroups take the block table: 2 = default: the table is dealt to the eight XCDs problem
 * by problem, so the row blocks of one problem gather the problem's column keys through ONE L2 while consecutive problems sit on
 * different XCDs -- by the counters 0.36 instead of 1.0 GB fetched per 4096-pair step, step time equal or 1 % better | 1 = every
 * XCD takes a contiguous eighth of the table: the same bytes, measured 8 % SLOWER for the stage | 0 = table order: a problem's
 * six row blocks sit on six XCDs and each fetches the whole column table; identical tables),
 * "split_target", "split_min_tiles" (column-split plans: workgroups per CU the split aims at, 0 = 3; tiles of 32 columns per
 * range at least, 0 = 4),
 * "graph" (plslam_match_plan_run as ONE replayed HIP graph -- the run's launches captured on the caller's stream at its first
 * use: 0 = plans of fewer waves than the chip has SIMDs | 1 = never (default: measured SLOWER on ROCm 7 -- C3's three-kernel
 * run 26.9 us per back-to-back run against 22.3 us with plain launches) | 2 = always; never while profiling),
 * "fuse" (K1f: one workgroup per problem that also merges the column results and applies the ratio test + mutual
 * check, i.e. one kernel per plan run: 0 = auto (currently: never -- measured no faster) | 1 = never | 2 = always),
 * "grid_dense" (plslam_match_grid, process-wide: 1 (default) = a lone problem of at most 256 x 256 rows runs on ONE workgroup with
 * everything in LDS (the SLAM loop's 200 x 200 line problems) | 0 = the general kernels; identical tables),
 * "zero_copy_kb" (plslam_match_grid: n > 0 (default 64) = an upload image of at most n KB is read by the dense kernel where it
 * lies in page-locked host memory -- no copy command in front of the kernel: 57 -> 51 us per 200 x 200 call | -n = for the general
 * kernels too (measured neutral to slower: they read their inputs many times) | 0 = always copy; identical tables) */
int plslam_ctx_set_option(plslam_ctx* ctx, const char* key, int value);
int plslam_ctx_get_option(plslam_ctx* ctx, const char* key, int* value);
/* device facts for reports: CU count, max clock (kHz), LDS bytes per workgroup */
int plslam_ctx_device_info(plslam_ctx* ctx, int32_t* cu_count, int32_t* clock_khz,
                           int32_t* lds_bytes, char* name, int32_t name_len);

/* ---- K1: brute-force Hamming kNN-2 ---------------------------------------------------- */
/* Replaces cv::BFMatcher::create(NORM_HAMMING,false)->knnMatch(q, t, matches, 2) as called by
 * stvo-pl's matchNNR (matching.cpp; reached from src/mapHandler.cpp:277,424,597,712,3223,3249).
 * idx, dist: nq*2 int32.  Order is lexicographic (distance, trainIdx): ties keep the lowest
 * train index, the second neighbour may share the first one's distance.  Absent neighbours
 * (nt < 2) are idx = -1, dist = INT32_MAX. */
int plslam_knn2_hamming256(plslam_ctx* ctx, const uint8_t* q, int32_t nq, const uint8_t* t,
                           int32_t nt, int32_t* idx, int32_t* dist);

/* ---- K1+K2: StVO::match ----------------------------------------------------------------- */
/* Replaces  int StVO::match(const cv::Mat& desc1, const cv::Mat& desc2, float nnr,
 *                           std::vector<int>& matches_12)           (stvo-pl matching.h,
 * included at src/mapHandler.cpp:28; call sites :277,:424,:597,:712,:3223,:3249).
 * matches_12 has n1 entries: matches_12[i1] = i2 or -1 (contract used at :280-283).
 * Ratio test in fp32: accept iff (float)d0 < (float)d1 * nnr.  mutual != 0 is the
 * reference's Config::bestLRMatches() (config/config/config_kitti.yaml:17): also run
 * desc2 -> desc1 and keep i1 -> i2 only if matches_21[i2] == i1.  n2 < 2 => no matches
 * (upstream indexes matches_[idx][1] out of range there; this ABI defines the case).
 * *n_matches (may be NULL) receives the return value of StVO::match. */
int plslam_match(plslam_ctx* ctx, const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2,
                 float nnr, int mutual, int32_t* matches_12, int32_t* n_matches);

/* StVO::match on a matches_12 that already holds n1 entries (in: the earlier table, out: the result); see
 * plslam_match_problem.keep_prior.  The drop-in header calls this when the caller's vector is not empty. */
int plslam_match_prior(plslam_ctx* ctx, const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2,
                       float nnr, int mutual, int32_t* matches_12, int32_t* n_matches);

/* B independent match() problems in one launch.  off1/off2 have B+1 row offsets into d1/d2;
 * matches_12 has off1[B] entries (problem b writes rows off1[b]..off1[b+1]); n_matches has B
 * entries (may be NULL).  This is the per-frame work of StereoFrame::extractStereoFeatures +
 * StereoFrameHandler::f2fTracking (stvo-pl; app/plslam_dataset.cpp:127) for a batch of frames. */
int plslam_match_batched(plslam_ctx* ctx, const uint8_t* d1, const int32_t* off1,
                         const uint8_t* d2, const int32_t* off2, int32_t B, float nnr,
                         int mutual, int32_t* matches_12, int32_t* n_matches);

/* ---- device-resident match plans (the throughput path) --------------------------------- */
/* One StVO::match problem with DEVICE pointers.  matches_12: n1 int32 (device).
 * n_matches: device pointer to one int32, or NULL. */
typedef struct plslam_match_problem {
    const uint8_t* d1;
    const uint8_t* d2;
    int32_t n1, n2;
    float nnr;
    int32_t mutual;
    int32_t* matches_12;
    int32_t* n_matches;
    /* 0: every row is written (-1 where nothing is accepted) -- StVO::match on a fresh vector.
     * 1: matches_12 already holds n1 entries and rows the ratio test rejects KEEP theirs; the consistency loop then
     *    runs over every entry >= 0 and n_matches = (rows accepted) - (entries cleared).  This is StVO::match on the
     *    vector matchGrid filled, src/mapHandler.cpp:271+277, :418+424, :591+597, :706+712 ([RECALL] stvo-pl matchNNR
     *    opens with matches_12.resize(desc1.rows, -1)). */
    int32_t keep_prior;
    int32_t reserved;
} plslam_match_problem;

typedef struct plslam_plan_info {
    int64_t distance_evals;    /* 256-bit distances evaluated per run (directed, as executed) */
    int64_t directed_evals;    /* distances the reference would evaluate (n1*n2 per direction) */
    int64_t algorithmic_bytes; /* sum over directed scans of 32*(Q+T) + 16*Q      */
    int32_t n_scans;           /* directed scans                                             */
    int32_t scan_blocks;       /* workgroups of the scan kernel                               */
    int32_t scan_variant;      /* PLSLAM_SCAN_* actually used                                 */
    int32_t scan_block_threads;
} plslam_plan_info;

/* Builds the immutable launch tables for `nprob` problems (host array of structs holding
 * device pointers) and uploads them.  Runs no kernel. */
int plslam_match_plan_create(plslam_ctx* ctx, const plslam_match_problem* probs, int32_t nprob,
                             plslam_match_plan** out);
/* Enqueues scan + finalize on `stream` (hipStream_t; NULL = context stream).  Asynchronous. */
int plslam_match_plan_run(plslam_match_plan* plan, void* stream);
/* The same with the scan kernel(s) on `scan_stream` and the stages behind them (merge of the column partials, finalize,
 * stereo gates, count scatter) on `post_stream`: a stream of runs -- of several plans or of this one -- then overlaps
 * each run's last stages (HBM-bound) with the next run's scan (instruction-issue-bound).  Ordering is the plan's
 * business: the stages wait for their scan, and the plan's next scan (split or not) waits for them.  Results are
 * complete when `post_stream` has drained.  A plan created with option "fuse" = 2 has no stage behind its scan (the scan
 * kernel writes the tables itself): it is not split, all of it runs on `scan_stream` and its results are complete when
 * THAT stream has drained. */
int plslam_match_plan_run_split(plslam_match_plan* plan, void* scan_stream, void* post_stream);
/* With profiling on, every run brackets each kernel with HIP events on the launch stream. */
int plslam_match_plan_set_profiling(plslam_match_plan* plan, int enable);
/* Synchronises the recorded events and returns accumulated kernel milliseconds since the
 * last call (then resets): the scan kernel(s) alone, then everything after them (merge of the
 * symmetric scan's column partials + finalize), and the number of profiled runs. */
int plslam_match_plan_elapsed(plslam_match_plan* plan, double* scan_ms, double* finalize_ms,
                              int64_t* runs);
int plslam_match_plan_info(plslam_match_plan* plan, plslam_plan_info* info);
/* Diagnostics (no reference counterpart): after a device-wide synchronise, copies the plan's intermediate key
 * table (per problem: n1 then n2 rows of two composite keys (distance << 23 | index), the state between the scan
 * and the finalize kernel) and the symmetric scan's per-workgroup column partials to the host.  Either buffer may be
 * NULL / too small: the sizes are always returned.  The determinism check (tools/determinism_check.py and its GPU
 * test) compares these words between repeated runs -- ~10^4 x more sensitive than the match tables, where only a
 * difference that flips a ratio test shows. */
int plslam_match_plan_dump(plslam_match_plan* plan, void* keys_out, size_t keys_cap, void* part_out,
                           size_t part_cap, size_t* keys_bytes, size_t* part_bytes);
/* What the key table of plslam_match_plan_dump holds for THIS plan (a bit mask; 0 = every word is an exact
 * (distance << 23 | index) key):
 *   PLSLAM_KEYS_ROW_SECOND_INDEX_INEXACT   the INDEX of a row's second key is the first column of its (group, class), not
 *                                          necessarily the second neighbour's (its distance is exact) -- K1h / K1i without
 *                                          "exact_second";
 *   PLSLAM_KEYS_COLUMN_SECOND_LAZY         a column's second key is (an upper bound of the second-best distance, index all
 *                                          ones): the stage behind the scan completes it where a match decision needs it;
 *   PLSLAM_KEYS_COLUMNS_NOT_IN_MEMORY      the merged column keys never reach memory (the fused stage behind the scan keeps
 *                                          them in LDS: option "post_fuse"); the table's column rows are unspecified.
 * Option "exact_second" = 1 at plan creation clears all three. */
#define PLSLAM_KEYS_ROW_SECOND_INDEX_INEXACT 1
#define PLSLAM_KEYS_COLUMN_SECOND_LAZY 2
#define PLSLAM_KEYS_COLUMNS_NOT_IN_MEMORY 4
int plslam_match_plan_key_state(plslam_match_plan* plan, int32_t* flags);
void plslam_match_plan_destroy(plslam_match_plan* plan);

/* ---- K14: StVO::matchGrid, the windowed ("fast_matching") matcher ------------------------------ */
/* Replaces  int StVO::matchGrid(const std::vector<point_2d>& points1, const cv::Mat& desc1,
 *                               const GridStructure& grid, const cv::Mat& desc2, const GridWindow& w,
 *                               std::vector<int>& matches_12)                       (points) and
 *           int StVO::matchGrid(const std::vector<line_2d>& lines1, const cv::Mat& desc1,
 *                               const GridStructure& grid, const cv::Mat& desc2,
 *                               const std::vector<std::pair<double,double>>& directions2,
 *                               const GridWindow& w, std::vector<int>& matches_12)  (lines)
 * of stvo-pl matching.h (un-vendored, [RECALL]); call sites src/mapHandler.cpp:271 (KF<->KF points), :418
 * (KF<->KF lines), :591 (map points <-> KF), :706 (map lines <-> KF).  This is the matcher the shipped
 * configurations use first (fast_matching: true); the callers fall back to StVO::match when it finds fewer than
 * min_*_matches (:274-278, :421-426, :594-598, :709-713).
 *   centres1   n1 x n_centres x 2 int32: the grid cells (x, y) whose windows supply row i1's candidates --
 *              n_centres = 1: points1[i1] (a point_2d = pair<int,int>); n_centres = 2: start and end point of
 *              lines1[i1].  Any int32 value is legal (GridStructure::get clamps the window to the grid).
 *   grid       the GridStructure the caller filled (:258-264, :395-411) in CSR form: cell (x, y), 0 <= x <
 *              grid_cols, 0 <= y < grid_rows, has id x*grid_rows + y and owns cell_items[cell_start[id] ..
 *              cell_start[id+1]-1]; items outside [0, n2) are ignored as upstream does.  GRID_COLS x GRID_ROWS
 *              is 64 x 48 upstream.
 *   window     {width.first, width.second, height.first, height.second} of the GridWindow (>= 0): cells
 *              x - window[0] .. x + window[1], y - window[2] .. y + window[3].
 *   dir1/dir2  line overload only (else NULL): unit directions of the query lines (what upstream computes from
 *              lines1[i1] with normalize()) and `directions2`; a candidate is skipped when |dot| < sim_th
 *              (Config::lineSimTh()).  A NaN direction (zero-length segment) never skips, as upstream.
 *   nnr        Config::minRatio12P(), used by BOTH overloads upstream; the test is `best_d < best_d2 * nnr` in
 *              fp64 with best_d2 = INT_MAX when there is no second candidate (a lone candidate is accepted).
 *   mutual     Config::bestLRMatches(): upstream's sequential `if (d < distances[i2]) {...} else continue;`
 *              (a candidate only counts for row i1 if it strictly beats every EARLIER row's distance to the
 *              same i2) followed by the matches_21[i2] == i1 check; reproduced exactly, order-free.
 *   matches_12 n1 entries (i2 or -1); *n_matches (may be NULL) the return value.  (With nnr > 1 -- no shipped
 *              configuration -- upstream also counts every row that had candidates none of which counted:
 *              `INT_MAX < INT_MAX * nnr`; reproduced, so there n_matches can exceed the entries >= 0.)
 * One deliberate definition: among several candidates at the same best distance upstream keeps the one its
 * std::unordered_set<int> happens to visit first (implementation-defined); here the lowest i2 wins, the
 * brute-force matcher's rule.  Limits: n1 < 2^22, n2 <= PLSLAM_MAX_TRAIN_ROWS. */
#define PLSLAM_MAX_GRID_ROWS (1 << 22)
int plslam_match_grid(plslam_ctx* ctx, const int32_t* centres1, int32_t n_centres, const uint8_t* d1,
                      int32_t n1, const int32_t* cell_start, const int32_t* cell_items, int32_t grid_cols,
                      int32_t grid_rows, const uint8_t* d2, int32_t n2, const double* dir1,
                      const double* dir2, double sim_th, const int32_t window[4], double nnr, int mutual,
                      int32_t* matches_12, int32_t* n_matches);

/* Device-resident form: a batch of matchGrid problems in ONE kernel launch (one workgroup per problem).
 * Every pointer of a problem is a DEVICE pointer; d1 / d2 must be 16-byte aligned.  pair_capacity sizes the
 * kernel's candidate store (mutual problems only; 0 otherwise): rows are handled in blocks of 1024 and a block
 * needs 1024 x (the number of grid items -- duplicates and out-of-range items included -- inside the windows of
 * its fullest row) entries of 4 bytes.  That size is always sufficient.  A problem whose candidates do not fit
 * matches nothing, gets n_matches = -1 and is counted by plslam_grid_plan_overflows.  (Problems of at most 2048 x 2048
 * rows whose tables fit the LDS keep their candidates there and use the store only for the excess: they may succeed
 * with less.) */
typedef struct plslam_grid_problem {
    const uint8_t* d1;
    const uint8_t* d2;
    const int32_t* centres1;
    const int32_t* cell_start;
    const int32_t* cell_items;
    const double* dir1;
    const double* dir2;
    int32_t n1, n2, n_centres, grid_cols, grid_rows;
    int32_t n_items; /* entries of cell_items: cell_start[grid_cols*grid_rows] must not exceed it (else: overflow) */
    int32_t window[4];
    double sim_th, nnr;
    int32_t mutual;
    int32_t pair_capacity;
    int32_t* matches_12;
    int32_t* n_matches; /* device pointer to one int32, or NULL */
} plslam_grid_problem;
typedef struct plslam_grid_plan plslam_grid_plan;
/* uploads the problem table and allocates the scratch (column lists, keys); runs no kernel */
int plslam_grid_plan_create(plslam_ctx* ctx, const plslam_grid_problem* probs, int32_t nprob,
                            plslam_grid_plan** out);
/* enqueues the kernel on `stream` (hipStream_t; NULL = the context's stream); asynchronous */
int plslam_grid_plan_run(plslam_grid_plan* plan, void* stream);
/* synchronises `stream` and returns the number of pair-list overflows since the last call */
int plslam_grid_plan_overflows(plslam_grid_plan* plan, void* stream, int32_t* n_overflows);
void plslam_grid_plan_destroy(plslam_grid_plan* plan);
/* plslam_grid_problem.pair_capacity from HOST copies of a problem's window centres and cell_start (pure host code, no
 * device needed).  exact: the documented size (rows in blocks of 1024, 1024 x the item count of a block's fullest row);
 * bound: an upper bound of it from the grid alone -- fullest cell x cells of a window (at most every item) x n_centres
 * per row -- for callers whose centres only exist on the device.  Either is sufficient.  0 for non-mutual problems. */
int64_t plslam_grid_pair_capacity(const int32_t* centres1, int32_t n1, int32_t n_centres, const int32_t* cell_start,
                                  int32_t grid_cols, int32_t grid_rows, const int32_t window[4], int mutual);
int64_t plslam_grid_pair_capacity_bound(int32_t n1, int32_t n_centres, const int32_t* cell_start, int32_t grid_cols,
                                        int32_t grid_rows, const int32_t window[4], int mutual);

/* ---- host-to-host pipeline: descriptors born on the host, tables wanted on the host -------------------------------- */
/* plslam_match_batched is strictly serial (H2D -> kernels -> D2H) and uploads every problem's rows separately.  A
 * pipeline is created ONCE for a batch shape: the host hands over one ARENA of descriptor rows per batch (any layout;
 * problems name byte offsets into it, so prev<->curr and L<->R problems share the rows of an image instead of carrying
 * copies: 109 kB per C2 stereo pair instead of 218 kB) and receives one int32 output table.  `depth` (2..8) batches are
 * in flight: the upload of batch k+1 and the download of batch k-1 run on their own HIP streams under the kernels of
 * batch k.  submit() returns as soon as the work is enqueued (it first waits for the batch that used the slot `depth`
 * submits ago); the output of a submit is complete after the submit that re-uses its slot or after wait().  Tables in
 * page-locked memory are written by the GPU itself behind the finalize (a copy-engine download would queue between two
 * uploads and serialise the pipeline).  Host
 * buffers should come from plslam_pinned_alloc (pageable memory works but the runtime then stages every copy itself). */
typedef struct plslam_arena_problem {
    int64_t d1_off, d2_off;      /* byte offsets of the two descriptor sets inside the arena (4-byte aligned)  */
    int32_t n1, n2;
    float nnr;
    int32_t mutual;
    int64_t out_off;             /* first entry of this problem's matches_12 inside the output table (int32 units) */
} plslam_arena_problem;
typedef struct plslam_match_pipeline plslam_match_pipeline;
int plslam_match_pipeline_create(plslam_ctx* ctx, size_t arena_bytes, const plslam_arena_problem* probs, int32_t nprob,
                                 size_t out_entries, int32_t depth, plslam_match_pipeline** out);
/* out_host / counts_host: page-locked AND 16-byte aligned buffers are written by a kernel straight from the compute
 * stream; anything else (pageable memory, an offset slice of a pinned buffer) takes a copy-engine download. */
int plslam_match_pipeline_submit(plslam_match_pipeline* pipe, const void* arena_host, int32_t* out_host,
                                 int32_t* counts_host /* nprob entries or NULL */);
int plslam_match_pipeline_wait(plslam_match_pipeline* pipe);
void plslam_match_pipeline_destroy(plslam_match_pipeline* pipe);
void* plslam_pinned_alloc(size_t bytes);      /* hipHostMalloc; NULL on failure */
void plslam_pinned_free(void* p);

/* ---- K15/K16: the stereo L<->R gates of StVO::StereoFrame ---------------------------------------- */
/* stvo-pl stereoFrame.cpp, matchStereoPoints / matchStereoLines ([RECALL]; the un-vendored dependency): what turns
 * the match table of (pdesc_l, pdesc_r) / (ldesc_l, ldesc_r) -- from StVO::match or StVO::matchGrid with the window
 * {matching_s_ws, 0, 0, 0} -- into the frame's stereo features.  Thresholds are the reference's config keys
 * max_dist_epip, min_disp, line_horiz_th, stereo_overlap_th, ls_min_disp_ratio
 * (config/config/config_kitti.yaml:25,26,34,31,36).
 *   points: kp = n x 2 float32 (cv::KeyPoint::pt.x, .y).  Kept iff |pt_l.y - pt_r.y| <= max_dist_epip (float
 *           subtraction) and disp = pt_l.x - pt_r.x >= min_disp.  disp: n_l doubles (0 where dropped).
 *   lines:  seg = n x 4 float32 (KeyLine startPointX, startPointY, endPointX, endPointY).  The right end points are
 *           moved along the right line to the rows of the left end points, disp_s / disp_e are the differences of x
 *           (both -1 when min/max < ls_min_disp_ratio); kept iff both >= min_disp, neither segment is horizontal
 *           (|dy| > line_horiz_th) and lineSegmentOverlapStereo of the y ranges > stereo_overlap_th.
 *           disp_se: n_l x 2 doubles.
 * stereo_12[i1] = i2 or -1; *n_stereo (may be NULL) = number kept. */
int plslam_stereo_point_gate(plslam_ctx* ctx, const int32_t* matches_12, int32_t n_l, const float* kp_l,
                             const float* kp_r, int32_t n_r, double max_dist_epip, double min_disp,
                             int32_t* stereo_12, double* disp, int32_t* n_stereo);
int plslam_stereo_line_gate(plslam_ctx* ctx, const int32_t* matches_12, int32_t n_l, const float* seg_l,
                            const float* seg_r, int32_t n_r, double min_disp, double line_horiz_th,
                            double stereo_overlap_th, double ls_min_disp_ratio, int32_t* stereo_12,
                            double* disp_se, int32_t* n_stereo);

/* Device-pointer forms: every pointer is a device pointer (kp / seg rows 8 / 16-byte aligned), the kernel is enqueued on
 * `stream` (NULL = the context's stream) and NOT synchronised; *n_stereo (device, may be NULL) is zeroed on the stream
 * first.  matches_12 is normally the device table a match plan or plslam_match_dev wrote. */
int plslam_stereo_point_gate_dev(plslam_ctx* ctx, const int32_t* matches_12, int32_t n_l, const float* kp_l,
                                 const float* kp_r, int32_t n_r, double max_dist_epip, double min_disp,
                                 int32_t* stereo_12, double* disp, int32_t* n_stereo, void* stream);
int plslam_stereo_line_gate_dev(plslam_ctx* ctx, const int32_t* matches_12, int32_t n_l, const float* seg_l,
                                const float* seg_r, int32_t n_r, double min_disp, double line_horiz_th,
                                double stereo_overlap_th, double ls_min_disp_ratio, int32_t* stereo_12,
                                double* disp_se, int32_t* n_stereo, void* stream);

/* The gate stage of a match plan: StereoFrame::matchStereoPoints / matchStereoLines for a whole batch.  Each gate
 * problem names the L<->R match table of one frame (usually the matches_12 of one of the plan's problems) and that
 * frame's key points / segments; after plslam_match_plan_add_stereo_gates, every plslam_match_plan_run ends with ONE
 * more launch that turns all those tables into stereo associations (stereo_12, disparities, counts).  All pointers are
 * device pointers and must stay valid for the life of the plan.  n_stereo: either NULL in every problem, or the
 * problems' counters form one contiguous int32 array (gates[i].n_stereo == gates[0].n_stereo + i).  Calling it again
 * replaces the stage; ngates = 0 removes it. */
typedef struct plslam_stereo_gate_problem {
    const int32_t* matches_12;   /* n_l entries: right index or -1                                    */
    const float* f_l;            /* n_l x 2 (points: pt.x, pt.y) or n_l x 4 (lines: sx, sy, ex, ey)  */
    const float* f_r;            /* n_r x 2 / n_r x 4                                                */
    int32_t n_l, n_r;
    int32_t lines;               /* 0 = points, 1 = lines                                            */
    int32_t pad;
    double max_dist_epip;        /* points  (config_kitti.yaml:25)                                   */
    double min_disp;             /* both    (:26)                                                    */
    double line_horiz_th;        /* lines   (:34)                                                    */
    double stereo_overlap_th;    /* lines   (:31)                                                    */
    double ls_min_disp_ratio;    /* lines   (:36)                                                    */
    int32_t* stereo_12;          /* out: n_l entries                                                 */
    double* disp;                /* out: n_l (points) or 2 n_l (lines: disp_s, disp_e) doubles       */
    int32_t* n_stereo;           /* out: number kept; may be NULL                                    */
} plslam_stereo_gate_problem;
/* (Calling it again replaces the stage: the call then waits for the device, so that no run still in flight reads tables
 * that are being replaced.) */
int plslam_match_plan_add_stereo_gates(plslam_match_plan* plan, const plslam_stereo_gate_problem* gates,
                                       int32_t ngates);

/* A 16-bit mirror of a plan's match tables (round 5): the wire format of the N > 1 table gather (the per-frame
 * tables the reference's loop would hand on, app/plslam_dataset.cpp:111-135).  Every problem whose matches_12 lies in
 * [table32, table32 + n_entries) also stores its entries, as int16, at the same index of table16 (device pointers) -- by the
 * kernel that decides them, so that no narrowing pass over the table is needed before the gather.  Requires n2 <= 32768 for
 * those problems and a plan whose tables are written by the finalize kernel (PLSLAM_ENOTSUP for fused and column-split
 * plans).  table16 = NULL removes the mirror.  Like plslam_match_plan_add_stereo_gates the call waits for the device. */
int plslam_match_plan_set_wire16(plslam_match_plan* plan, const int32_t* table32, int16_t* table16, size_t n_entries);

/* ---- K3/K4: local-BA residual + Jacobian rows ------------------------------------------- */
/* Point rows: the per-observation body of MapHandler::levMarquardtOptimizationLBA,
 * src/mapHandler.cpp:1358-1407 (first pass) == :1587-1642 (iteration pass; the caller
 * supplies poses/landmarks from X).  fp64, no FMA contraction, reference operation order.
 *   T_kf_w  nkf*16  row-major KF->world poses (map_keyframes[kf]->T_kf_w, :1370)
 *   Xw      npt*3   landmarks (:1368)          obs_uv nobs*2 observations (:1375)
 *   lm_loc[o] -> row of Xw, kf_slot[o] -> pose in T_kf_w
 * out: J_pose nobs*6 (:1392-1398, tangent order [t, w]), J_lm nobs*3 (:1401-1404),
 *      r nobs (||p_err||, :1377), w nobs (robustWeightCauchy, :1407). */
int plslam_lba_point_rows(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                          const double* T_kf_w, int32_t nkf, const double* Xw, int32_t npt,
                          const double* obs_uv, const int32_t* lm_loc, const int32_t* kf_slot,
                          int32_t nobs, double* J_pose, double* J_lm, double* r, double* w);
/* Line rows: src/mapHandler.cpp:1436-1516.  compat_iter_pass != 0 reproduces the iteration
 * pass :1668-1748 as written: both endpoints read Lw[3*lm_loc .. +3] (:1678-1679) and the
 * threshold is the literal 1e-7 (:1698).  Lw holds n_lw doubles (6 per line landmark).
 * out: J_pose nobs*6 (:1509), J_lm nobs*6 (:1486,:1506,:1512-1513), r (:1460), w (:1516). */
int plslam_lba_line_rows(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                         int compat_iter_pass, const double* T_kf_w, int32_t nkf,
                         const double* Lw, int32_t n_lw, const double* l_obs,
                         const int32_t* lm_loc, const int32_t* kf_slot, int32_t nobs,
                         double* J_pose, double* J_lm, double* r, double* w);
/* device-pointer forms (all arrays on the device; asynchronous on `stream`) */
int plslam_lba_point_rows_dev(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                              const double* T_kf_w, const double* Xw, const double* obs_uv,
                              const int32_t* lm_loc, const int32_t* kf_slot, int32_t nobs,
                              double* J_pose, double* J_lm, double* r, double* w, void* stream);
int plslam_lba_line_rows_dev(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                             int compat_iter_pass, const double* T_kf_w, const double* Lw,
                             const double* l_obs, const int32_t* lm_loc, const int32_t* kf_slot,
                             int32_t nobs, double* J_pose, double* J_lm, double* r, double* w,
                             void* stream);
/* The same with the length of T_kf_w stated (ABI v5, round 6): n_pose_slots = the number of 4 x 4 matrices behind T_kf_w (> max
 * kf_slot).  The kernels then keep the first 32 of them in LDS instead of gathering them from global memory row by row: K3 at C3
 * sizes 0.345 -> 0.276 ms per 12.8 M rows.  n_pose_slots = 0 is the call above.  Identical rows. */
int plslam_lba_point_rows_dev_n(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                                const double* T_kf_w, int32_t n_pose_slots, const double* Xw, const double* obs_uv,
                                const int32_t* lm_loc, const int32_t* kf_slot, int32_t nobs,
                                double* J_pose, double* J_lm, double* r, double* w, void* stream);
int plslam_lba_line_rows_dev_n(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                               int compat_iter_pass, const double* T_kf_w, int32_t n_pose_slots, const double* Lw,
                               const double* l_obs, const int32_t* lm_loc, const int32_t* kf_slot,
                               int32_t nobs, double* J_pose, double* J_lm, double* r, double* w,
                               void* stream);

/* ---- K7-K10: normal equations of the local BA in block form ------------------------------- */
/* Replaces the accumulation into the dense H / g of levMarquardtOptimizationLBA
 * (src/mapHandler.cpp:1410-1429 points, :1519-1538 lines) by its block structure:
 *   g       N = 6*nkf + 3*npt + 6*nls      (the layout of the reference's X / g)
 *   H_pose  nkf * 6x6   H.block(idx,idx,6,6)          H_pt  npt * 3x3   H.block(jdx,jdx,3,3)
 *   H_ls    nls * 6x6   H.block(jdx,jdx,6,6)
 *   W_pt    n_pt_obs * 3x6   the observation's Haux = H.block(jdx,idx,3,6) contribution (:1424-1426)
 *   W_ls    n_ls_obs * 6x6   (:1533-1535); zero when kf_loc == -1 (keyframe not optimised, :1413)
 *   err     sum r^2 w (:1416,1423,1525,1532)
 * Inputs are the rows of plslam_lba_*_rows plus the Vector6i columns lm_loc (1) and kf_loc (4)
 * (:1259,1263).  Every block entry is the sequential sum over the observations in list order
 * (points, then lines), i.e. the dense accumulation's own order; all blocks row-major. */
int plslam_lba_assemble(plslam_ctx* ctx, int32_t nkf, int32_t npt, int32_t nls,
                        const int32_t* pt_lm_loc, const int32_t* pt_kf_loc, int32_t n_pt_obs,
                        const double* pt_J_pose, const double* pt_J_lm, const double* pt_r,
                        const double* pt_w, const int32_t* ls_lm_loc, const int32_t* ls_kf_loc,
                        int32_t n_ls_obs, const double* ls_J_pose, const double* ls_J_lm,
                        const double* ls_r, const double* ls_w, double* g, double* H_pose,
                        double* H_pt, double* H_ls, double* W_pt, double* W_ls, double* err);

/* LBA plan: levMarquardtOptimizationLBA rebuilds rows + H/g once (:1358-1540) and then up to
 * max_iters_lba = 15 times (:1587-1772) while only X changes.  The plan uploads what is constant --
 * the Vector6i columns lm_loc (1) / kf_loc (4), the pose slot of each observation, the observations
 * themselves -- once; iterate() uploads poses (n_pose_slots*16, e.g. expmap_se3 of X, :1600-1603) and
 * landmarks (X.block(6Nkf..), :1597, :1678), runs K3/K4 and K7-K10 on the device and returns the
 * block-form normal equations (layout as plslam_lba_assemble).
 * Pose slots are per observation and separate for points and lines on purpose: in the iteration pass the reference
 * gives an optimised key frame's POINT observations the current estimate expmap_se3(X.block(6*kf_loc..)) (:1600-1601)
 * but its LINE observations the stored map_keyframes[kf]->T_kf_w (:1680) -- a caller that wants the reference's
 * numbers keeps one slot per key frame with the stored pose for ls_pose_slot and a second set with the current
 * estimates for pt_pose_slot, and passes compat_iter_pass = 1 (stride-3 end points :1677-1678, literal 1e-7 :1698).
 * All three quirks are checked against the reference's own source text by the CPU test suite. */
typedef struct plslam_lba_plan plslam_lba_plan;
int plslam_lba_plan_create(plslam_ctx* ctx, const plslam_cam* K, double homog_th, int32_t n_pose_slots,
                           int32_t nkf, int32_t npt, int32_t nls, const int32_t* pt_lm_loc,
                           const int32_t* pt_pose_slot, const int32_t* pt_kf_loc, const double* pt_obs_uv,
                           int32_t n_pt_obs, const int32_t* ls_lm_loc, const int32_t* ls_pose_slot,
                           const int32_t* ls_kf_loc, const double* ls_l_obs, int32_t n_ls_obs,
                           plslam_lba_plan** out);
/* compat_flags (a bit mask; 1 keeps its round-1 meaning "compat_iter_pass"):
 *   PLSLAM_LBA_COMPAT_ITER_PASS  the iteration pass's line quirks (see above)
 *   PLSLAM_LBA_COMPAT_GBA        levMarquardtOptimizationGBA writes the pose x line cross blocks TRANSPOSED
 *                                (src/mapHandler.cpp:2341-2352 against :1531-1532; H stays symmetric but the block is
 *                                J_l J_p^T where J_p J_l^T belongs): W_ls comes out the way the reference's GBA fills it */
#define PLSLAM_LBA_COMPAT_ITER_PASS 1
#define PLSLAM_LBA_COMPAT_GBA 2
int plslam_lba_plan_iterate(plslam_lba_plan* plan, const double* T_kf_w, const double* Xw, const double* Lw,
                            int compat_flags, double* g, double* H_pose, double* H_pt, double* H_ls,
                            double* W_pt, double* W_ls, double* err);
/* The same iteration with the blocks LEFT ON THE DEVICE: X goes up (0.34 MB at C3), only *err (and g, N doubles, unless
 * NULL) comes down -- not the 11.5 MB of blocks that made the host-to-host iteration 2.3 ms for 0.1 ms of kernels.
 * plslam_lba_plan_device_blocks names the device arrays (valid until the next iterate / destroy; ordered on `stream`, the
 * context's stream) for a device-side Schur complement / solver; plslam_lba_plan_blocks downloads any of them afterwards
 * (NULL = skip). */
int plslam_lba_plan_iterate_dev(plslam_lba_plan* plan, const double* T_kf_w, const double* Xw, const double* Lw,
                                int compat_flags, double* g, double* err);
typedef struct plslam_lba_blocks {
    const double *g, *H_pose, *H_pt, *H_ls, *W_pt, *W_ls, *err;   /* device pointers, layouts as plslam_lba_plan_iterate */
    void* stream;                                                  /* the HIP stream the blocks were computed on     */
} plslam_lba_blocks;
int plslam_lba_plan_device_blocks(plslam_lba_plan* plan, plslam_lba_blocks* out);
int plslam_lba_plan_blocks(plslam_lba_plan* plan, double* g, double* H_pose, double* H_pt, double* H_ls, double* W_pt,
                           double* W_ls, double* err);
/* The optimisation state itself on the device (round 4): after one plslam_lba_plan_iterate / _iterate_dev has uploaded it,
 * plslam_lba_plan_device_state names the device copies of T_kf_w (n_pose_slots x 16), Xw (npt x 3) and Lw (nls x 6) -- a
 * device-side solver applies its update to them in place (the reference's X <- X + delta, T <- T inv(exp(delta)),
 * src/mapHandler.cpp:1563-1568), on `stream` -- and plslam_lba_plan_iterate_resident runs the next iteration on them: no
 * upload, three launches, *err down (8 bytes).  Blocks as after plslam_lba_plan_iterate_dev. */
typedef struct plslam_lba_state {
    double *T_kf_w, *Xw, *Lw;                 /* device pointers */
    int32_t n_pose_slots, npt, nls;
    void* stream;                              /* the HIP stream the plan's iterations run on */
} plslam_lba_state;
int plslam_lba_plan_device_state(plslam_lba_plan* plan, plslam_lba_state* out);
int plslam_lba_plan_iterate_resident(plslam_lba_plan* plan, int compat_flags, double* err);
/* The plan's page-locked host images by name (round 5).  Every plslam_lba_plan_iterate / _iterate_dev copies the caller's
 * T_kf_w / Xw / Lw into one page-locked image (one upload) and g out of another (one download).  A host solver that keeps its
 * state IN that image -- X_aux of src/mapHandler.cpp:1231-1330 written there, X(i) += DX(i) applied there -- and reads g
 * there passes these very pointers to the iterate calls, which then skip both staging copies (0.34 MB each way at C3).  The
 * images belong to the plan (valid until plslam_lba_plan_destroy); the caller may write T / Xw / Lw and read g between calls
 * only (every plan call returns with the stream synchronised).  g is rewritten by every _iterate_dev with g != NULL;
 * pointers of any other origin keep working as before, array by array. */
typedef struct plslam_lba_host_state {
    double *T_kf_w, *Xw, *Lw;                 /* page-locked: n_pose_slots x 16, npt x 3, nls x 6 */
    double* g;                                 /* page-locked: n doubles (6 nkf + 3 npt + 6 nls) */
    int32_t n_pose_slots, npt, nls;
    int64_t n;
} plslam_lba_host_state;
int plslam_lba_plan_host_state(plslam_lba_plan* plan, plslam_lba_host_state* out);
/* The Schur step on the resident blocks (round 5) -- the solve of src/mapHandler.cpp:1544-1575 (and :1779-1800 in the loop)
 * with everything but a 6 nkf x 6 nkf system staying on the device.  The reference damps H(i,i) += lambda * H(i,i) and solves
 * the whole N x N system with a sparse LDLT; the landmark blocks of H are independent, so the same step is
 *     S dp = b,  S = Hpp' - sum_j Wj^T Vj'^-1 Wj,  b = gp - sum_j Wj^T Vj'^-1 gj,  dxj = Vj'^-1 (gj - Wj dp)
 * (Hpp', Vj': the damped pose / landmark blocks; Wj: the cross blocks of landmark j's observations).  All three calls work on
 * the blocks of the LAST plslam_lba_plan_iterate / _iterate_dev / _iterate_resident of the plan (run WITHOUT
 * PLSLAM_LBA_COMPAT_GBA: the transposed pose x line cross blocks of the reference's GBA are refused, PLSLAM_EINVAL).
 *   plslam_lba_plan_diag_max   *hmax = max |H(i,i)| over all N diagonal entries (the reference's "lambda *= Hmax", :1544-1550)
 *   plslam_lba_plan_schur      S (6 nkf x 6 nkf doubles, row-major, both triangles) and b (6 nkf) for the damping `lambda`;
 *                              *n_singular (may be NULL) = landmarks whose damped block is not positive definite (a zero
 *                              diagonal entry: no observation constrains that coordinate) -- they contribute nothing and get
 *                              a zero step.  Deterministic: fixed-shape sums, no atomics.
 *   plslam_lba_plan_backsub    the landmark steps for the pose step dpose (6 nkf doubles, the solution of S dp = b that the
 *                              host's dense LDLT found): dX_pt (npt x 3) / dX_ls (nls x 6), either may be NULL; apply != 0
 *                              also adds them to the resident Xw / Lw in place (X(i) += DX(i), :1570-1575).  Needs the
 *                              plslam_lba_plan_schur of the same blocks before it (its landmark inverses are reused).
 *   plslam_lba_plan_set_poses  uploads the poses alone (n_pose_slots x 16 doubles): the caller applies dp to them
 *                              (T <- T inv(exp(dp)), :1560-1566: the SE(3) maps stay on the host) -- after which
 *                              plslam_lba_plan_iterate_resident runs the next iteration with nothing else crossing PCIe. */
int plslam_lba_plan_diag_max(plslam_lba_plan* plan, double* hmax);
int plslam_lba_plan_schur(plslam_lba_plan* plan, double lambda, double* S, double* b, int32_t* n_singular);
int plslam_lba_plan_backsub(plslam_lba_plan* plan, const double* dpose, int apply, double* dX_pt, double* dX_ls);
int plslam_lba_plan_set_poses(plslam_lba_plan* plan, const double* T_kf_w);
/* An LM iteration of src/mapHandler.cpp:1583-1812 in TWO calls and two synchronisations (ABI v5, round 6):
 *   plslam_lba_plan_iterate_schur = plslam_lba_plan_iterate_resident + plslam_lba_plan_schur for a lambda known beforehand
 *                              (every iteration but the first pass, whose lambda needs plslam_lba_plan_diag_max): err, S, b and
 *                              the count of singular landmark blocks come back behind ONE synchronisation;
 *   plslam_lba_plan_apply_step = plslam_lba_plan_backsub(dpose, apply) + plslam_lba_plan_set_poses(T_kf_w; NULL = leave the
 *                              pose slots, a rejected step) + sum_i DX(i)^2 over the LANDMARK steps (dx_sumsq, may be NULL): with
 *                              |dpose|^2 added on the host this is the ||DX|| of the loop's last test (:1808), and 8 bytes
 *                              cross PCIe instead of the steps.  The sum is taken in a fixed order: the same bits on every run. */
int plslam_lba_plan_iterate_schur(plslam_lba_plan* plan, int compat_flags, double lambda, double* err, double* S, double* b,
                                  int32_t* n_singular);
int plslam_lba_plan_apply_step(plslam_lba_plan* plan, const double* dpose, const double* T_kf_w, int apply, double* dx_sumsq);
/* The resident landmarks, device -> host (ABI v5): Xw (npt x 3) / Lw (nls x 6), either may be NULL.  What
 * plslam_lba_plan_backsub(apply) updated in place comes back for the reference's write-back (src/mapHandler.cpp:1822-1852:
 * point3D / line3D <- X, inlier = false where ||X - old|| > 0.01).  The plan's page-locked images (plslam_lba_plan_host_state)
 * are refreshed by the same copy: after backsub(apply) the image holds the landmarks of BEFORE the step until this call (or the
 * caller) rewrites it -- an iterate / iterate_dev handed the image's pointers in between would upload the old ones. */
int plslam_lba_plan_get_landmarks(plslam_lba_plan* plan, double* Xw, double* Lw);
/* rows of the last iterate() (any pointer may be NULL), e.g. for the write-back logic of :1822-1855 */
int plslam_lba_plan_rows(plslam_lba_plan* plan, double* pt_J_pose, double* pt_J_lm, double* pt_r, double* pt_w,
                         double* ls_J_pose, double* ls_J_lm, double* ls_r, double* ls_w);
void plslam_lba_plan_destroy(plslam_lba_plan* plan);

/* ---- K17: pose-only Gauss-Newton system of the loop-closure relative pose ------------------------------- */
/* The iteration body of MapHandler::computeRelativePoseGN (src/mapHandler.cpp:3324-3424) and of
 * computeRelativePoseRobustGN (:3588-3689; identical loops): per inlier point / line the reprojection error, the 6-vector
 * Jacobian wrt the pose increment (note: fx / max(homogTh, z^2) scales BOTH image coordinates, and the line rows use
 * l_obs(0..1) as multipliers -- unlike the local-BA rows) and the Cauchy weight, accumulated into
 *   H = H_p + H_l (6 x 6 row-major), g = g_p + g_l, *e = e_p + e_l (BEFORE the division by N_l + N_p of :3421),
 *   n_obs[0] = N_p, n_obs[1] = N_l (may be NULL).
 * T_inc: 16 doubles row-major; P npt x 3 (lc_points[i]->P), pl_obs npt x 2, pt_inlier npt bytes; sPeP nls x 6
 * (lc_lines[i]->sP, ->eP), le_obs nls x 3, ls_inlier nls bytes.  The solve (ColPivHouseholderQR of a 6 x 6, :3426-3428)
 * and the update T_inc = T_inc * inverse_se3(expmap_se3(x_inc)) stay with the caller. */
int plslam_pose_gn_accumulate(plslam_ctx* ctx, const plslam_cam* K, double homog_th, const double* T_inc,
                              const double* P, const double* pl_obs, const uint8_t* pt_inlier, int32_t npt,
                              const double* sPeP, const double* le_obs, const uint8_t* ls_inlier, int32_t nls,
                              double* H, double* g, double* e, int32_t* n_obs);

/* ---- K5/K6: map <-> keyframe geometric gates (the inlier masks) -------------------------- */
/* Points: src/mapHandler.cpp:601-613.  mask[i] = 1 iff matches_12[i] >= 0 and
 * || proj(Twf * Xw[i]) - pl[matches_12[i]] ||_2 < max_epip.  Twf: 16 doubles row-major.
 * Xw nq*3 (the matched map points, query order), pl nt*2.  *n_inliers = #mask (may be NULL). */
int plslam_map2kf_point_gate(plslam_ctx* ctx, const plslam_cam* K, const double* Twf,
                             const double* Xw, const int32_t* matches_12, int32_t nq,
                             const double* pl, int32_t nt, double max_epip, uint8_t* mask,
                             int32_t* n_inliers);
/* Lines: src/mapHandler.cpp:716-729 (signed test on both endpoints, no abs()).
 * Lw nq*6, le nt*3 (normalised 2D line equations). */
int plslam_map2kf_line_gate(plslam_ctx* ctx, const plslam_cam* K, const double* Twf,
                            const double* Lw, const int32_t* matches_12, int32_t nq,
                            const double* le, int32_t nt, double max_epip, uint8_t* mask,
                            int32_t* n_inliers);
/* Candidate pre-filter src/mapHandler.cpp:549-551 / :650-655: vis[i] = 1 iff the landmark
 * projects strictly inside the image with positive depth (both endpoints for lines). */
int plslam_map_point_visible(plslam_ctx* ctx, const plslam_cam* K, const double* Twf,
                             const double* Xw, int32_t n, uint8_t* vis);
int plslam_map_line_visible(plslam_ctx* ctx, const plslam_cam* K, const double* Twf,
                            const double* Lw, int32_t n, uint8_t* vis);

/* ---- the map <-> keyframe association drivers, fused ---------------------------------------- */
/* Replaces the compute of MapHandler::matchMap2KFPoints (src/mapHandler.cpp:532-632), brute-force
 * path (fast_matching == false); the map mutation (:614-626) stays with the caller.
 *   candidate[i] != 0  <=>  the reference's  pt != nullptr && pt->local && pt->kf_obs_list.back() != kf2_idx (:547)
 *   Xw n_map*3 = pt->point3D, med_desc n_map*32 = pt->med_desc (:555)
 *   kf_desc n_kf*32 = pdesc_l, kf_pl n_kf*2 = stereo_pt[i]->pl, kf_idx[i] = stereo_pt[i]->idx (-1 = unmatched, :565)
 *   min_matches = SlamConfig::minPointMatches() (:594-596), max_epip = SlamConfig::maxKFEpipP() (:613)
 * On the device: projection + inside-image test (:549-551), Q/T matrix construction (:555, :567),
 * StVO::match (:597), epipolar gate (:610-613).  map_to_kf[n_map] receives, per map landmark, the
 * ORIGINAL keyframe feature index it is associated with, or -1; *n_matches the return value. */
int plslam_map2kf_match_points(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* Xw,
                               const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                               const uint8_t* kf_desc, const double* kf_pl, const int32_t* kf_idx,
                               int32_t n_kf, float nnr, int mutual, double max_epip, int32_t min_matches,
                               int32_t* map_to_kf, int32_t* n_matches);
/* MapHandler::matchMap2KFLines (src/mapHandler.cpp:634-752): Lw n_map*6 = ls->line3D, kf_le n_kf*3 =
 * stereo_ls[i]->le; both endpoints must project inside (:654-655); signed gate (:727-729);
 * min_matches = SlamConfig::minLineMatches() (:709-711). */
int plslam_map2kf_match_lines(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* Lw,
                              const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                              const uint8_t* kf_desc, const double* kf_le, const int32_t* kf_idx,
                              int32_t n_kf, float nnr, int mutual, double max_epip, int32_t min_matches,
                              int32_t* map_to_kf, int32_t* n_matches);

/* The same drivers with SlamConfig::fastMatching() (`fast_matching: true`, the shipped configurations):
 * src/mapHandler.cpp:578-592 (points) / :681-707 (lines).  The candidates' projections become grid cells
 * (pj_points / pj_lines: pixels * inv_width / inv_height truncated to int) on the device, the unmatched
 * keyframe features fill the GridStructure (points: their cell, :581-584; lines: the Bresenham cells of
 * (spl, epl) and their directions, :686-698), StVO::matchGrid runs with a window of `ws` cells and the
 * ratio nnr_grid (Config::minRatio12P() for BOTH kinds); then, exactly as in the plain drivers,
 * StVO::match(nnr) replaces its result when |Q| > min_matches && matches < min_matches (:594-598,
 * :709-713).  *used_match (may be NULL) reports whether that happened.  fm == NULL or !fm->enabled is the
 * plain driver.  kf_seg: n_kf x 4 doubles = stereo_ls[i]->spl, ->epl (lines only). */
typedef struct plslam_fast_matching {
    int32_t enabled;              /* SlamConfig::fastMatching() */
    int32_t grid_cols, grid_rows; /* GRID_COLS (64), GRID_ROWS (48) */
    int32_t ws;                   /* SlamConfig::matchingF2FWs() */
    double inv_width, inv_height; /* StereoFrame::inv_width / inv_height (GRID_COLS / width, GRID_ROWS / height) */
    double nnr_grid;              /* Config::minRatio12P() */
    double line_sim_th;           /* Config::lineSimTh() */
} plslam_fast_matching;
int plslam_map2kf_match_points_fast(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* Xw,
                                    const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                                    const uint8_t* kf_desc, const double* kf_pl, const int32_t* kf_idx,
                                    int32_t n_kf, float nnr, int mutual, double max_epip, int32_t min_matches,
                                    const plslam_fast_matching* fm, int32_t* map_to_kf, int32_t* n_matches,
                                    int32_t* used_match);
int plslam_map2kf_match_lines_fast(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* Lw,
                                   const uint8_t* med_desc, const uint8_t* candidate, int32_t n_map,
                                   const uint8_t* kf_desc, const double* kf_le, const double* kf_seg,
                                   const int32_t* kf_idx, int32_t n_kf, float nnr, int mutual, double max_epip,
                                   int32_t min_matches, const plslam_fast_matching* fm, int32_t* map_to_kf,
                                   int32_t* n_matches, int32_t* used_match);
/* The same drivers with the MAP SIDE DEVICE-RESIDENT: d_Xw / d_Lw, d_med_desc and d_candidate are device pointers (8-byte
 * aligned) -- a local map lives on the GPU across keyframes (the representative descriptors come from
 * plslam_median_desc_batched_dev, the landmarks from the LBA plan); the host-pointer forms above stage and upload 0.56 MB of
 * it per call at C3 sizes (10 000 landmarks), which is a quarter of their time.  The keyframe side (kf_*), the results and the
 * semantics are those of the _fast forms (fm == NULL or !fm->enabled: the plain driver).  kf_seg: lines only. */
int plslam_map2kf_match_points_dev(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* d_Xw,
                                   const uint8_t* d_med_desc, const uint8_t* d_candidate, int32_t n_map,
                                   const uint8_t* kf_desc, const double* kf_pl, const int32_t* kf_idx, int32_t n_kf,
                                   float nnr, int mutual, double max_epip, int32_t min_matches,
                                   const plslam_fast_matching* fm, int32_t* map_to_kf, int32_t* n_matches,
                                   int32_t* used_match);
int plslam_map2kf_match_lines_dev(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* d_Lw,
                                  const uint8_t* d_med_desc, const uint8_t* d_candidate, int32_t n_map,
                                  const uint8_t* kf_desc, const double* kf_le, const double* kf_seg,
                                  const int32_t* kf_idx, int32_t n_kf, float nnr, int mutual, double max_epip,
                                  int32_t min_matches, const plslam_fast_matching* fm, int32_t* map_to_kf,
                                  int32_t* n_matches, int32_t* used_match);


/* ---- the keyframe <-> keyframe association drivers ------------------------------------------------ */
/* Replace the compute of MapHandler::matchKF2KFPoints (src/mapHandler.cpp:234-363; :246-278) and matchKF2KFLines
 * (:365-530; :378-426); creating the MapPoints / MapLines from matches_12 (:280-363, :428-530) stays with the caller.
 *   fm enabled: the previous keyframe's stereo features are projected with DT (16 doubles row-major; :254-256,
 *     :384-394) on the device, the current keyframe's features fill the GridStructure (:259-263 / :397-411), window of
 *     matching_f2f_ws cells, StVO::matchGrid.  Points: pj_points = projection * inv_width / inv_height.  Lines: pj_lines
 *     are the projected PIXELS -- upstream does not multiply them by inv_width (:392-393) -- truncated to int like
 *     everything that goes into a point_2d; a projection that is not a finite int32 becomes INT_MIN (x86 cvttsd2si).
 *   then StVO::match(prev, curr, nnr) iff n_curr > min_matches && n_prev > min_matches && matches < min_matches
 *     (:274-278, :421-425).
 * P_prev n_prev x 3 (stereo_pt[i]->P) / sPeP_prev n_prev x 6 (stereo_ls[i]->sP, ->eP); pl_curr n_curr x 2 /
 * seg_curr n_curr x 4 (spl, epl).  matches_12: n_prev entries (all -1 when no matcher ran). */
int plslam_kf2kf_match_points(plslam_ctx* ctx, const plslam_cam* K, const double* DT, const double* P_prev,
                              const uint8_t* desc_prev, int32_t n_prev, const double* pl_curr,
                              const uint8_t* desc_curr, int32_t n_curr, float nnr, int mutual, int32_t min_matches,
                              const plslam_fast_matching* fm, int32_t* matches_12, int32_t* n_matches,
                              int32_t* used_match);
int plslam_kf2kf_match_lines(plslam_ctx* ctx, const plslam_cam* K, const double* DT, const double* sPeP_prev,
                             const uint8_t* desc_prev, int32_t n_prev, const double* seg_curr,
                             const uint8_t* desc_curr, int32_t n_curr, float nnr, int mutual, int32_t min_matches,
                             const plslam_fast_matching* fm, int32_t* matches_12, int32_t* n_matches,
                             int32_t* used_match);
/* The same with the ROWS ON THE DEVICE: d_P_prev / d_sPeP_prev (n_prev x 3 | 6 doubles), d_desc_prev and d_desc_curr (x 32
 * bytes, 16-byte aligned) are device pointers -- a keyframe's descriptors and 3D features stay on the GPU from one call to
 * the next (they are the rows the stereo and frame-to-frame plans already hold); nothing is staged or uploaded but the grid
 * the host builds from pl_curr / seg_curr (HOST pointers, as above).  Results and semantics are those of the forms above. */
int plslam_kf2kf_match_points_dev(plslam_ctx* ctx, const plslam_cam* K, const double* DT, const double* d_P_prev,
                                  const uint8_t* d_desc_prev, int32_t n_prev, const double* pl_curr,
                                  const uint8_t* d_desc_curr, int32_t n_curr, float nnr, int mutual,
                                  int32_t min_matches, const plslam_fast_matching* fm, int32_t* matches_12,
                                  int32_t* n_matches, int32_t* used_match);
int plslam_kf2kf_match_lines_dev(plslam_ctx* ctx, const plslam_cam* K, const double* DT, const double* d_sPeP_prev,
                                 const uint8_t* d_desc_prev, int32_t n_prev, const double* seg_curr,
                                 const uint8_t* d_desc_curr, int32_t n_curr, float nnr, int mutual,
                                 int32_t min_matches, const plslam_fast_matching* fm, int32_t* matches_12,
                                 int32_t* n_matches, int32_t* used_match);

/* ---- representative ("median") descriptor of every landmark, batched ------------------------ */
/* Replaces the descriptor part of MapPoint::updateAverageDescDir (src/mapFeatures.cpp:51-84) and of
 * MapLine::updateAverageDescDir (:121-157), which the reference runs per landmark whenever an
 * observation is added (:47, :118): per landmark, the observation whose row of the pairwise
 * Hamming matrix has the smallest element int(1+0.5*(n-1)) after sorting (:77), first row winning
 * ties (strict '<' :78).  All landmarks in one call, observation lists concatenated:
 *   desc_lists  total x 32 uint8  landmark l owns rows offsets[l] .. offsets[l+1]-1 (desc_list order)
 *   offsets     n_lm+1 int32, offsets[0] = 0, non-decreasing; every list shorter than 2^23 rows
 *   med_idx     n_lm int32: winner's position within its own list (0 for a single observation --
 *               the constructors :28-38 -- and -1 for an empty list, which the reference never has)
 *   med_desc    n_lm x 32 uint8 = desc_list[med_idx] (zeros for an empty list); may be NULL.  These
 *               rows are the `med_desc` argument of plslam_map2kf_match_points/_lines.
 * The direction average (:86-91) accumulates into an uninitialised vector upstream and is not
 * reproduced.  The _dev form takes device pointers (4-byte aligned) plus the total row count and
 * enqueues on `stream` (NULL = the context's stream) without synchronising. */
int plslam_median_desc_batched(plslam_ctx* ctx, const uint8_t* desc_lists, const int32_t* offsets,
                               int32_t n_lm, int32_t* med_idx, uint8_t* med_desc);
int plslam_median_desc_batched_dev(plslam_ctx* ctx, const uint8_t* desc_lists, const int32_t* offsets,
                                   int32_t n_lm, int32_t total, int32_t* med_idx, uint8_t* med_desc,
                                   void* stream);

/* ---- K18: the LBD float descriptor of a line (producer of the 72 floats K11 binarises) -------------------- */
/* Replaces BinaryDescriptor::computeLBD, 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:1026-1372, for the
 * lines of ONE octave: gradient images dx / dy (int16, width x height, row stride = width: dxImg_vector[octave] /
 * dyImg_vector[octave], :1095-1101) and, per line, the OctaveSingleLine fields the function reads (numOfPixels,
 * sPointInOctaveX/Y, ePointInOctaveX/Y, direction).  width_of_band = params.widthOfBand_ (7; 1..7 supported),
 * NUM_OF_BANDS = 9.  lbd_f32: n x 72 float32 = the lines' `descriptor` vectors -- the input of
 * plslam_lbd_binarise.  fp32 with the source's sequential summation order and no FMA; the weight tables and
 * cos / sin(direction) are evaluated on the host with libm as upstream does.
 * _dev form: dx_img / dy_img / lbd_f32 are DEVICE pointers (the line records stay host data: they are the line
 * detector's small output); enqueues on `stream` (NULL = the context's stream). */
typedef struct plslam_lbd_line {
    int32_t num_pixels;
    float sx, sy, ex, ey;
    float direction;
} plslam_lbd_line;
int plslam_lbd_compute(plslam_ctx* ctx, const int16_t* dx_img, const int16_t* dy_img, int32_t width, int32_t height,
                       const plslam_lbd_line* lines, int32_t n, int32_t width_of_band, float* lbd_f32);
int plslam_lbd_compute_dev(plslam_ctx* ctx, const int16_t* dx_img, const int16_t* dy_img, int32_t width,
                           int32_t height, const plslam_lbd_line* lines_host, int32_t n, int32_t width_of_band,
                           float* lbd_f32, void* stream);

/* ---- LBD float -> 256-bit binary line descriptor (producer of the matcher's LBD rows) ----- */
/* Replaces the "fill current row with binary descriptor" loop of BinaryDescriptor::computeImpl,
 * 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:653-668, with
 * binaryConversion (:401-412: bit i of a byte set iff f1[i] > f2[i]) over the 32 band pairs of
 * combinations[32][2] (:74-107); NUM_OF_BANDS 9 (:57) x 8 floats = 72 floats per line.
 *   lbd_f32  n x 72 float32, row-major contiguous (the ScaleLines' `descriptor` vectors, packed)
 *   desc_u8  n x 32 uint8: the cv::Mat(n,32,CV_8UC1) rows StVO::match consumes for lines
 * Comparisons with a NaN operand leave the bit clear, as in the reference.  The _dev form takes
 * device pointers (both 16-byte aligned) and enqueues on `stream`
 * (NULL = the context's stream) without synchronising. */
#define PLSLAM_LBD_FLOATS 72
int plslam_lbd_binarise(plslam_ctx* ctx, const float* lbd_f32, int32_t n, uint8_t* desc_u8);
int plslam_lbd_binarise_dev(plslam_ctx* ctx, const float* lbd_f32, int32_t n, uint8_t* desc_u8,
                            void* stream);

/* ---- multi-GPU: gather of per-frame match tables over RCCL/xGMI -------------------------- */
/* No reference counterpart (the reference is single-process).  `comm` is an ncclComm_t the
 * host created (one rank per GPU); `local` is this rank's n_local int32 match-table entries
 * (device); `gathered` (device, root only) receives nranks*n_local entries in rank order.
 * Implemented as one ncclGroup of point-to-point sends to root so each peer uses its own
 * xGMI link.  RCCL is loaded lazily (dlopen); returns PLSLAM_ENOTSUP if it cannot be. */
int plslam_gather_match_tables(plslam_ctx* ctx, void* comm, int nranks, int rank, int root,
                               const int32_t* local, int64_t n_local, int32_t* gathered,
                               void* stream);

/* Which librccl the host's communicators come from (call before the first gather; NULL = the default search: librccl.so,
 * librccl.so.1, /opt/rocm/lib/librccl.so).  A communicator belongs to ONE loaded copy of the library -- PyTorch ships its own
 * beside ROCm's -- and the send / receive entry points must be that copy's. */
int plslam_rccl_use(const char* path);
/* 1 when librccl's group / send / recv entry points could be loaded (the first call loads them), 0 otherwise: asked before the
 * first plslam_match_plan_step_gather, so that a missing library is a set-up failure and the caller can keep its own gather. */
int plslam_rccl_available(void);

/* One step of the N > 1 path in one call (ABI v5): the plan's scan on scan_stream, everything behind it on post_stream
 * (plslam_match_plan_run_split), then the gather of the finished table to `root` -- one ncclGroup of point-to-point transfers,
 * each peer on its own xGMI link -- and, on the root with the int16 wire format, the widening to int32; all enqueued, nothing
 * waited for on the host.  The next step of the SAME plan orders itself behind this gather (both streams) before anything
 * rewrites the table; plslam_match_plan_gather_sync waits for it on the host.
 *   wire_bytes 4: `send` = the plan's int32 match table itself (n_entries entries), `recv` (root) = nranks * n_entries int32 in
 *                 rank order -- the gathered tables of the C ABI; `wide` unused
 *   wire_bytes 2: `send` = the int16 mirror the finalize kernel writes (plslam_match_plan_set_wire16), `recv` (root) = nranks *
 *                 n_entries int16, `wide` (root; may be NULL) = nranks * n_entries int32, written behind the gather
 *   comm_stream   the stream the collective (and the widening) is enqueued on; NULL = post_stream
 * nranks == 1: no communicator needed (the root's own slice is copied / widened). */
typedef struct plslam_gather_step {
    void* comm;                  /* ncclComm_t of this rank */
    int32_t nranks, rank, root;
    int32_t wire_bytes;
    const void* send;
    int64_t n_entries;
    void* recv;
    int32_t* wide;
    void *scan_stream, *post_stream, *comm_stream;
} plslam_gather_step;
int plslam_match_plan_step_gather(plslam_match_plan* plan, const plslam_gather_step* step);
int plslam_match_plan_gather_sync(plslam_match_plan* plan);

#ifdef __cplusplus
}
#endif
#endif /* PLSLAM_HIP_H */
