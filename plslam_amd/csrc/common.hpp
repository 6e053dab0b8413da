// common.hpp -- internal declarations shared by the HIP translation units of libplslam_hip.so.
// Nothing here is part of the ABI (see include/plslam_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <mutex>
#include <vector>

#include "plslam_hip.h"

namespace plslam {

void set_last_error(const char* fmt, ...);

#define PLSLAM_HIP_CHECK(expr)                                                              \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            ::plslam::set_last_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,           \
                                     hipGetErrorString(e_));                                \
            return e_ == hipErrorOutOfMemory ? PLSLAM_ENOMEM : PLSLAM_EHIP;                 \
        }                                                                                   \
    } while (0)

#define PLSLAM_REQUIRE(cond, code)                                                          \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            ::plslam::set_last_error("%s:%d: requirement failed: %s", __FILE__, __LINE__,    \
                                     #cond);                                                \
            return (code);                                                                  \
        }                                                                                   \
    } while (0)

// grow-only device scratch buffer owned by a context
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return static_cast<T*>(p); }
};
// pinned host memory (hipHostMalloc), grow-only: staging for the latency path of the host-pointer entry
// points -- copies from/to pinned memory are plain DMA enqueues, copies from pageable memory are not
struct HostBuf {
    void* p = nullptr;
    void* dev = nullptr;     // the device address of the block (mapped page-locked memory), nullptr if the device cannot address it
    size_t cap = 0;
    int reserve(size_t bytes);
    void release();
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// ring of page-locked slots for small per-call host tables a kernel reads in place (lbd_float.hip: the line records)
struct LineRing {
    static constexpr int SLOTS = 4;
    HostBuf rec[SLOTS];
    hipEvent_t done[SLOTS] = {nullptr, nullptr, nullptr, nullptr};   // recorded behind the kernel that read the slot
    bool busy[SLOTS] = {false, false, false, false};
    int next = 0;
    void release()
    {
        for (int k = 0; k < SLOTS; ++k) {
            if (done[k]) (void)hipEventDestroy(done[k]);
            done[k] = nullptr;
            busy[k] = false;
            rec[k].release();
        }
    }
};

// the device address of page-locked, mapped host memory (hipHostMalloc / hipHostRegister), or nullptr
inline void* mapped_device_pointer(void* host)
{
    hipPointerAttribute_t at;
    if (!host || hipPointerGetAttributes(&at, host) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return at.type == hipMemoryTypeHost ? at.devicePointer : nullptr;
}

struct DeviceGuard {  // every entry point runs on the context's device
    int prev = -1;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
        else prev = -1;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

// Host-pointer entry points enqueue copies from stack / local objects and work on the context's shared scratch buffers:
// whatever way they return (PLSLAM_HIP_CHECK, `return rc`), nothing may still be in flight when the locals die and
// ctx->mu is released.  On the success path the stream is already idle and the extra synchronise costs a microsecond.
struct StreamSyncOnError {
    hipStream_t s;
    bool armed = true;
    explicit StreamSyncOnError(hipStream_t st) : s(st) {}
    void dismiss() { armed = false; }
    ~StreamSyncOnError()
    {
        if (armed) (void)hipStreamSynchronize(s);
    }
};

// carve several arrays out of one scratch buffer (256-byte aligned slices)
struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~size_t(255); return o; }
};

}  // namespace plslam

struct plslam_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop;
    int scan_variant = PLSLAM_SCAN_AUTO;
    int scan_block = 0;  // 0 = variant default
    int group_cap = 0;   // blocks of one problem kept together on one XCD; 0 = auto (capi.hip, `stripe`)
    int sym_rows = 0;    // rows of d1 per lane in the symmetric scan: 0 = auto, 1, 4
    int mfma_form = 0;   // matrix-core scan: 0 = auto (= 5), 1 = exact push per tile (K1e), 2 = grouped rows (K1f), 3 = directed pairs (K1g), 4 = grouped both ways (K1h), 5 = K1h with the M-tiles pipelined against each other (K1i)
    int col_split = 0;   // K1f, few large problems: 0 = auto (cut the columns into ranges when the plan cannot fill the chip), 1 = never, 2 = always
    int exact_second = 0; // K1h: 1 = the index of every second-best row key is exact (0: only where it is an output -- knnMatch)
    int graph = 1;             // plslam_match_plan_run as a replayed HIP graph: 0 = latency plans, 1 = never (default until measured), 2 = always
    int scan_tail = 0;         // (experiment builds -DPLSLAM_MI_TAIL=1) 1: K1i's last workgroup of a problem merges the column partials, no merge launch
    int post_workgroups = 0;   // > 0: the stages behind a scan (merge of K1h's partials, finalize) run as at most this many workgroups walking their block tables
    int split_target = 0, split_min_tiles = 0;   // column split: workgroups per CU aimed at (0 = 3), tiles per column range at least (0 = 4)
    int split_post = 0;  // column-split K1f plans: 0 = auto (merge + ratio + mutual behind the scan in ONE kernel: two launches per run), 1 = never
    int post_xcd = 2;    // finalize: 0 = table order, 1 = an XCD takes contiguous entries of the block table (a problem's row blocks share one L2), 2 = the table dealt to the XCDs problem by problem (default)
    int zero_copy_kb = 64;  // (round 6) host-pointer calls of ONE small problem (plslam_match_grid): an upload image of at most this many KB is read by the kernel where it lies in page-locked host memory -- no H2D copy command in front of it -- when the dense one-workgroup kernel takes the problem (it reads every word once); negative: for every problem whose image fits; 0 = always copy
    int post_fuse = 0;   // K1h / K1i plans: merge + finalize + gates behind the scan as ONE kernel: 0 = auto (throughput plans), 1 = never, 2 = whenever eligible
    int fuse = 0;        // K1f: 0 = auto (one workgroup per problem incl. merge + finalize when the plan is large), 1 = never, 2 = always
    std::mutex mu;       // serialises the host-pointer entry points
    plslam::DevBuf in_a, in_b, out_a, out_b, misc_a, misc_b, misc_c;
    plslam::HostBuf pin_in, pin_out;                // pinned staging of small host-pointer calls
    plslam::HostBuf pin_misc;                       // host-built tables of the map-level drivers (map2kf.hip)
    plslam::LineRing lbd_ring;                      // line records of the last plslam_lbd_compute* calls
    struct plslam_match_plan* host_plan = nullptr;  // reused by the host-pointer match entry points
};

namespace plslam {

// Pointers read from launch tables are GENERIC to the compiler, and a generic access is a FLAT instruction (it counts on
// lgkmcnt as well as vmcnt, and cannot take a scalar base).  Kernels spell the address space out at the point of use:
// g_(p)[i] is a global_load / global_store.  (tests/test_abi.py: no flat_* instruction in any kernel's ISA.)
#if defined(__HIPCC__)
#define PLSLAM_AS1 __attribute__((address_space(1)))
#define g_(p) ((PLSLAM_AS1 __typeof__(*(p))*)(p))          /* keeps the pointee's typedef (alignment attributes included) */
typedef uint32_t gvec2_t __attribute__((ext_vector_type(2)));   // native vectors: HIP's uint2 / uint4 structs cannot be copied
typedef uint32_t gvec4_t __attribute__((ext_vector_type(4)));   // out of / into another address space
template <class T> __device__ __forceinline__ int atomic_add_global(T* p, int v)
{
    return __hip_atomic_fetch_add((PLSLAM_AS1 int*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#endif

// --- Hamming scan (hamming.hip) ------------------------------------------------------------
// composite key: (distance << 23) | trainIdx; 0xFFFFFFFF = "no neighbour"
constexpr uint32_t KEY_IDX_BITS = 23;
constexpr uint32_t KEY_IDX_MASK = (1u << KEY_IDX_BITS) - 1u;
constexpr uint32_t KEY_NONE = 0xFFFFFFFFu;

struct ScanDesc {       // one directed scan: every query row against every train row
    const uint8_t* q;   // nq x 32
    const uint8_t* t;   // nt x 32
    uint32_t* keys;     // nq x 2 composite keys (best, second best)
    int32_t nq, nt;
};

// ---- K1h (hamming_mfma_h.hip): the layout of b over (tile, class): class-major inside groups of 16 tiles of 32 rows ------
// full groups: 512 rows each; the ragged rest (n2 mod 512 rows) is one more group of S = ceil(rest / 32) tiles with S rows per
// class.  Column slot of the partial table = 32 tile + class.
struct MhLayout {
    int n2, nfull, ntiles, rag_s;       // nfull: tiles of full groups (a multiple of 16); rag_s: tiles (= rows per class) of the ragged group
    __host__ __device__ explicit MhLayout(int n2_) : n2(n2_)
    {
        const int full = n2_ / 512, rest = n2_ - full * 512;
        nfull = full * 16;
        rag_s = (rest + 31) / 32;
        ntiles = nfull + rag_s;
    }
    __host__ __device__ int stride(int t) const { return t < nfull ? 16 : rag_s; }
    __host__ __device__ int row_of(int t, int cls) const { return (t >> 4) * 512 + stride(t) * cls + (t & 15); }   // may be >= n2
    __host__ __device__ int slot_of(int j) const
    {
        const int G = j >> 9, off = j & 511;
        const int s = G * 16 < nfull ? 16 : rag_s;
        // off / s for off < 512, 1 <= s <= 16 without a division: floor(off * ceil(2^16 / s) / 2^16), exact while off < 2^16 / s
        constexpr uint32_t recip[17] = {0, 65536, 32768, 21846, 16384, 13108, 10923, 9363, 8192, 7282, 6554, 5958, 5462, 5042,
                                        4682, 4370, 4096};
        const int cls = (int)(((uint32_t)off * recip[s]) >> 16), tt = off - cls * s;
        return 32 * (G * 16 + tt) + cls;
    }
    __host__ __device__ int slots_padded() const { return (32 * ntiles + 255) & ~255; }     // slots per row of the partial table
};

struct ProblemDesc {    // one StVO::match problem = scan12 (+ scan21 when mutual)
    const uint32_t* keys12;
    const uint32_t* keys21;  // nullptr when !mutual
    int32_t* matches_12;
    int32_t* n_matches;      // may be nullptr
    int32_t n1, n2;
    float nnr;
    int32_t mutual;
    // column-split problems (K1f, capi.hip): the row results arrive as `nsplit` tables [nsplit][n1][2] with column indices
    // relative to ranges of `cstep` columns; the finalize kernel merges them on the fly (and stores the merged pair to
    // keys12_out for diagnostics).  nsplit <= 1: keys12 is final.
    const uint32_t* split_tmp;
    uint32_t* keys12_out;
    int32_t nsplit, cstep;
    // plslam_match_problem.keep_prior: rows the ratio test rejects keep what matches_12 holds (stvo-pl's resize())
    int32_t keep_prior;
    // K1h plans: keys21[j] = (best row, best row OUTSIDE the best row's aligned group of 16 rows of d1): the exact second best
    // is recomputed here, and only for the columns a row actually points at (15 XOR + popcount distances from d1 / d2)
    int32_t lazy21;
    const uint8_t* d1;
    const uint8_t* d2;
    // index of the stereo-gate problem that consumes this table (plslam_match_plan_add_stereo_gates), or -1: the finalize
    // kernel applies the gate to a row's match the moment it is decided -- no second launch, no second pass over the table
    int32_t gate, pad2;
    // K1h / K1i plans with the fused stage behind the scan (k_post_fused): the problem's column partials -- [row block][slot]
    // words, SymDesc::part21 -- or nullptr
    const uint32_t* part21;
    // plslam_match_plan_set_wire16: the int16 mirror of matches_12 (the gather's wire format), or nullptr
    int16_t* matches_16;
};

struct BlockDesc {      // one workgroup's slice of a scan / problem
    int32_t item;       // scan or problem index
    int32_t row0;       // first query row of this workgroup
};

// launches (all asynchronous on `s`)
int launch_scan(const plslam_ctx* ctx, int variant, int block_threads, const ScanDesc* d_scans,
                const BlockDesc* d_blocks, int nblocks, int32_t* d_zero, int nzero, hipStream_t s);
int launch_finalize(const ProblemDesc* d_probs, const BlockDesc* d_blocks, int nblocks,
                    const plslam_stereo_gate_problem* d_gates, hipStream_t s, int grid_cap = 0, bool xcd_chunks = false, int dealt_row = 0);
// K2' (hamming.hip): merge of K1h's / K1i's column partials + finalize + gates, one workgroup per problem; lds_bytes = 8 x the
// largest n2 of the plan
constexpr int POST_FUSED_MAX_N2 = 4096;
constexpr int POST_FUSED_MAX_ROW_BLOCKS = 16;
int launch_post_fused(const ProblemDesc* d_probs, int nprob, const plslam_stereo_gate_problem* d_gates, size_t lds_bytes,
                      hipStream_t s);
int launch_scatter_counts(const int32_t* d_src, int32_t* const* d_dst, int32_t n, hipStream_t s);
int launch_unpack_keys(const uint32_t* d_keys, int32_t n, int32_t* d_idx, int32_t* d_dist,
                       hipStream_t s);
// symmetric scan: one mutual problem = one (a x b) distance matrix feeding both directions
struct SymDesc {
    const uint8_t* a;       // n1 rows: one per lane
    const uint8_t* b;       // n2 rows: streamed
    uint32_t* keys12;       // n1 x 2   row results (complete)
    uint32_t* keys21;       // n2 x 2   column results (written by the merge kernel)
    uint32_t* part21;       // [n_iblk][n2][2] column partials per 64-row block of a
    int32_t n1, n2;
    int32_t n_iblk;
    int32_t mutual;         // fused form (K1f, one workgroup per problem): ratio + mutual finalize happen in the scan kernel
    int32_t* matches_12;    //   n1 match-table entries (nullptr: not fused)
    int32_t* n_matches;     //   one counter, STORED (not accumulated) by the problem's workgroup; may be nullptr
    float nnr;
    int32_t flags;          // bit 0 (K1h): the INDEX of the second-best row key must be exact too (knnMatch output, key dumps)
    // K1f, two-launch column-split plans only (the brute-force map<->keyframe driver in one synchronisation, map2kf.hip): the
    // number of rows of a lives on the DEVICE (*n1_dev <= n1; n1 is the bound the tables and the launch are sized for), or nullptr
    const int32_t* n1_dev;
};
// rows_per_lane: 1 (K1b: 256-thread workgroups, 64 a-rows per wave) or 4 (K1b': 64-thread
// workgroups, 256 a-rows per wave).  sym_rows_per_block() = a-rows covered by one BlockDesc.
int sym_rows_per_block(int rows_per_lane);
int sym_rows_per_partial(int rows_per_lane);
int launch_scan_sym(int rows_per_lane, const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks,
                    int32_t* d_zero, int nzero, hipStream_t s);
// K1e: the symmetric scan on the matrix cores (hamming_mfma.hip); block tables / partials as for
// launch_scan_sym(rows_per_lane = 4): 256 a-rows per workgroup and per column partial
// multi_window: some problem of the launch has n2 > 2048 (row keys hold 64 tiles of 32 columns per window)
// directed: only keys12 of every SymDesc is produced (keys21 / part21 unused): non-mutual problems, knnMatch
int launch_scan_sym_mfma(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero,
                         int nzero, bool multi_window, bool directed, hipStream_t s);
// K1f (hamming_mfma_g.hip): same contract and tables as K1e; row direction = group minima + second best by recomputation
// fused: one block-table entry per PROBLEM (row0 = 0); the workgroup walks the problem's row blocks itself, then merges the
// column partials and applies the ratio test + mutual check (SymDesc::matches_12 / n_matches / nnr / mutual): no merge
// kernel, no finalize kernel, no counter zeroing for these problems.  Mutual problems need n2 <= PLSLAM_K1F_FUSED_MAX_N2.
constexpr int PLSLAM_K1F_FUSED_MAX_N2 = 4096;
int launch_scan_sym_mfma_g(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero,
                           int nzero, bool multi_window, bool directed, bool fused, hipStream_t s);
// parts: lanes sharing one column (1, 4 or 16); the block table has one entry per merge_partials16_cols(parts) columns
int merge_partials16_cols(int parts);
int launch_merge_partials16(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int parts, hipStream_t s);
// K1c'' + K2 behind a column-split K1f scan in one kernel (two-launch plan runs); the dump's row completion
int launch_split_post(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int parts, const ProblemDesc* d_probs, hipStream_t s);
int launch_split_rows_dump(const ProblemDesc* d_probs, const BlockDesc* d_fin_blocks, int nblocks, hipStream_t s);
// (Column split of a large problem, K1f: the columns are cut into ranges scanned as sub-problems of their own -- more
// workgroups than 256-row blocks alone give; the per-range row results are merged by the finalize kernel: ProblemDesc.)
// K1h (hamming_mfma_h.hip): K1f's contract and partial table; minimum-only bookkeeping in both directions, class-major layouts;
// its partials need launch_merge_fix16 (merge + second-best recomputation), same block table as launch_merge_partials16
inline bool mfma_form_is_h(int form) { return form == 0 || form == 4 || form == 5; }      // 0 = auto (= 5); K1h's tables: K1h and K1i
int launch_scan_sym_mfma_h(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero, int nzero,
                           bool directed, hipStream_t s);
// K1i (hamming_mfma_i.hip): K1h with the two M-tiles of a wave pipelined against each other; same tables, same merge kernel
// tail_counts (experiment builds -DPLSLAM_MI_TAIL=1 only; k1i_tail_built()): one zeroed counter per problem -- the last workgroup
// of a problem merges its column partials into keys21 itself, no merge kernel behind the scan
int launch_scan_sym_mfma_i(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero, int nzero,
                           bool directed, hipStream_t s, int32_t* tail_counts = nullptr);
bool k1i_tail_built();
int merge_fix16_cols(int parts);      // column slots per block-table entry of launch_merge_fix16
int launch_merge_fix16(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int parts, bool fix, hipStream_t s, int grid_cap = 0);
int mh_slot_of_column(int n2, int j);
// K1g (hamming_mfma_d.hip): the directed scan, one item per (directed scan, 256-row block of a); any n2 (windows inside)
int launch_scan_dir_mfma(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero, int nzero, hipStream_t s);
// PLSLAM_BUILD_LEGACY_SCANS (plslam_amd/build.py; default 0): the earlier generations of the matrix-core scan -- K1e
// (hamming_mfma.hip, mfma_form 1), K1g (hamming_mfma_d.hip, 3) and K1h's scan kernel (hamming_mfma_h.hip, 4) -- are compiled in.
// AUTO never picks them; without them plslam_ctx_set_option("mfma_form", 1 | 3 | 4) returns PLSLAM_ENOTSUP.
#ifndef PLSLAM_BUILD_LEGACY_SCANS
#define PLSLAM_BUILD_LEGACY_SCANS 0
#endif
inline bool mfma_form_built(int form) { return PLSLAM_BUILD_LEGACY_SCANS || form == 0 || form == 2 || form == 5; }
inline int launch_scan_mfma_form(int form, const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero,
                                 int nzero, bool multi_window, bool directed, hipStream_t s, bool fused = false,
                                 int32_t* tail_counts = nullptr)
{
#if PLSLAM_BUILD_LEGACY_SCANS
    if (form == 3 && directed && !fused) return launch_scan_dir_mfma(d_sym, d_blocks, nblocks, d_zero, nzero, s);
    if (form == 4 && !fused) return launch_scan_sym_mfma_h(d_sym, d_blocks, nblocks, d_zero, nzero, directed, s);
#endif
    if (mfma_form_is_h(form) && !fused) return launch_scan_sym_mfma_i(d_sym, d_blocks, nblocks, d_zero, nzero, directed, s, tail_counts);
#if PLSLAM_BUILD_LEGACY_SCANS
    if (form == 1) return launch_scan_sym_mfma(d_sym, d_blocks, nblocks, d_zero, nzero, multi_window, directed, s);
#endif
    return launch_scan_sym_mfma_g(d_sym, d_blocks, nblocks, d_zero, nzero, multi_window, directed, fused, s);
}
int launch_merge_partials(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, hipStream_t s);

int scan_rows_per_block(int variant, int block_threads);

// builds the plan for `probs` (DEVICE pointers) into the context's persistent host-path plan and
// enqueues it on the context stream; no synchronisation.  Caller holds ctx->mu.  (capi.hip)
// n1_dev0: the row count of problem 0 on the device (probs[0].n1 = its bound); PLSLAM_ENOTSUP when the plan cannot take it
int match_problems_on_ctx_stream(plslam_ctx* ctx, const plslam_match_problem* probs, int32_t nprob, const int32_t* n1_dev0 = nullptr);
// whether the context's options allow the n1_dev0 form at all (asked before anything is staged or enqueued for it)
bool ctx_takes_device_row_count(const plslam_ctx* ctx);
// dst[i] = src[idx[i]] for rows of row_bytes (a multiple of 8) bytes  (map2kf.hip)
int launch_gather_rows(const void* src, const int32_t* idx, int32_t n, int32_t row_bytes, void* dst,
                       hipStream_t s);

// --- LBA rows + gates (lba.hip) --------------------------------------------------------------
int launch_point_rows(const plslam_cam& K, double th, const double* T, const double* Xw,
                      const double* uv, const int32_t* lm, const int32_t* kf, int32_t nobs,
                      double* Jp, double* Jl, double* r, double* w, hipStream_t s, int32_t n_pose_slots = 0);
int launch_line_rows(const plslam_cam& K, double th, int compat, const double* T, const double* Lw,
                     const double* lobs, const int32_t* lm, const int32_t* kf, int32_t nobs,
                     double* Jp, double* Jl, double* r, double* w, hipStream_t s, int32_t n_pose_slots = 0);
int launch_point_gate(const plslam_cam& K, const double* Twf16, const double* Xw, const int32_t* m12,
                      int32_t nq, const double* pl, double th, uint8_t* mask, int32_t* count,
                      hipStream_t s);
int launch_line_gate(const plslam_cam& K, const double* Twf16, const double* Lw, const int32_t* m12,
                     int32_t nq, const double* le, double th, uint8_t* mask, int32_t* count,
                     hipStream_t s);
int launch_visible(const plslam_cam& K, const double* Twf16, const double* X, int32_t n, int lines,
                   uint8_t* vis, hipStream_t s);
// cells (n x [1|2] x 2 int32) of the projected landmarks in grid units; lines: also dir1 (n x 2)
int launch_project_cells(const plslam_cam& K, const double* Twf16, const double* X, int32_t n, int lines, double inv_w,
                         double inv_h, int32_t* cells, double* dir1, hipStream_t s);

// --- stereo gates (stereo_gates.hip) -------------------------------------------------------------
int launch_stereo_gates(const plslam_stereo_gate_problem* d_gates, const BlockDesc* d_blocks, int nblocks, hipStream_t s);
int check_stereo_gate_problem(const plslam_stereo_gate_problem& q);

// --- representative descriptor per landmark (median_desc.hip) ---------------------------------
// desc: total x 32 u8 (4-byte aligned), off: n_lm+1 CSR offsets (device), med_idx: n_lm, med_desc:
// n_lm x 32 or nullptr.  memset + 2 kernels on s.
int launch_median_desc(const uint8_t* desc, const int32_t* off, int32_t n_lm, int32_t total,
                       int32_t* med_idx, uint8_t* med_desc, hipStream_t s);

// --- StVO::matchGrid, the windowed matcher (match_grid.hip) --------------------------------------
struct GridDesc {              // one matchGrid problem; every pointer is a device pointer
    const uint8_t* d1;         // n1 x 32
    const uint8_t* d2;         // n2 x 32
    const int32_t* centres;    // n1 x n_centres x 2 window centres (x, y)
    const int32_t* cell_start; // cols*rows + 1   GridStructure in CSR form, cell id = x*rows + y
    const int32_t* cell_items;
    const double* dir1;        // n1 x 2 or nullptr (points)
    const double* dir2;        // n2 x 2 or nullptr
    int32_t* matches_12;       // n1
    int32_t* n_matches;        // 1 or nullptr
    uint32_t* scratch;         // grid_scratch_words() words: [tables when they do not fit LDS |] pair list
    int32_t* status;           // incremented when the pair list does not fit pair_cap; may be nullptr
    double sim_th, nnr;
    int32_t n1, n2, n_centres, cols, rows, mutual;
    int32_t w[4];              // width.first, width.second, height.first, height.second
    int32_t pair_cap;
    int32_t n_items;           // entries of cell_items the caller declared (cell_start[cols*rows] must not exceed it)
};
extern int g_grid_dense;       // ctx option "grid_dense" (process-wide): 1 = a small lone problem runs on k_match_grid_dense
size_t grid_fixed_words(int32_t n1, int32_t n2, int64_t ncell);   // tables kept in LDS when they fit
bool grid_fits_lds(int32_t n1, int32_t n2, int64_t ncell);
size_t grid_lds_bytes(int mode, int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs = false);
int grid_mode(int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs = false);   // 2: all in LDS, 1: tables in LDS, 0: global
size_t grid_scratch_words(int32_t n1, int32_t n2, int64_t ncell, int32_t pair_cap);
int64_t grid_store_capacity_host(const int32_t* centres, int32_t n1, int32_t n_centres, const int32_t* cell_start,
                                 int32_t cols, int32_t rows, const int32_t window[4], int mutual);
// one problem with DEVICE pointers on `s` (scratch: grid_scratch_words() words; status: one zeroed int32)
int launch_match_grid_one(const plslam_grid_problem& q, uint32_t* scratch, int32_t* status, GridDesc* d_desc_slot,
                          GridDesc* h_desc_slot, hipStream_t s);
// the same in two steps (the descriptor inside a larger upload of the caller): host-side check + fill, then the launch
int grid_prepare_one(const plslam_grid_problem& q, uint32_t* scratch, int32_t* status, GridDesc* h_desc_slot);
int grid_launch_prepared(const plslam_grid_problem& q, const GridDesc* d_desc_slot, hipStream_t s);
// launch groups: 3 = all in LDS / 256-lane workgroups (n1 <= 256), 2 = all in LDS, 1 = tables in LDS, 0 = global
int grid_group(int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs = false);
size_t grid_group_lds_bytes(int group, int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs = false);
// table order: the n[3] problems of group 3, then n[2], n[1], n[0]
int launch_match_grid(const GridDesc* d_probs, const int32_t n[4], const size_t lds_bytes[4], hipStream_t s);

// --- LBD float -> binary line descriptor (lbd.hip) ---------------------------------------------
// lbd: n x 72 f32, codes: n x 32 u8 (both 16-byte aligned)
int launch_lbd_binarise(const float* lbd, int32_t n, uint8_t* codes, hipStream_t s);

}  // namespace plslam
