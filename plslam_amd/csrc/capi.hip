// capi.hip -- the extern "C" boundary of libplslam_hip.so (see include/plslam_hip.h).
// Context, match plans, host-pointer convenience entry points, RCCL gather.
#include <dlfcn.h>
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "common.hpp"

namespace plslam {

static thread_local char g_err[512] = "";

void set_last_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int DevBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return PLSLAM_OK;
    if (p) {
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes + (bytes >> 2);  // grow by 25 % to damp re-allocation
    want = (want + 255) & ~size_t(255);
    PLSLAM_HIP_CHECK(hipMalloc(&p, want));
    cap = want;
    return PLSLAM_OK;
}

void DevBuf::release()
{
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

int HostBuf::reserve(size_t bytes)
{
    if (bytes <= cap) return PLSLAM_OK;
    if (p) {
        (void)hipHostFree(p);
        p = nullptr;
        dev = nullptr;
        cap = 0;
    }
    size_t want = bytes + (bytes >> 2);
    want = (want + 4095) & ~size_t(4095);
    PLSLAM_HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    dev = mapped_device_pointer(p);                // (asked once: hipPointerGetAttributes per call was 1-2 us of a 40 us call)
    return PLSLAM_OK;
}

void HostBuf::release()
{
    if (p) (void)hipHostFree(p);
    p = nullptr;
    dev = nullptr;
    cap = 0;
}

}  // namespace plslam

using namespace plslam;

// ---------------------------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------------------------
struct plslam_match_plan {
    plslam_ctx* ctx = nullptr;
    int variant = 0, block_threads = 0;
    int32_t nprob = 0, nscan = 0, nscan_blocks = 0, nfin_blocks = 0, ncounts = 0;
    int32_t fin_row = 0;               // > 0: the finalize table is dealt to the XCDs problem by problem, 8 rows of this length (option post_xcd 2)
    int32_t nsym = 0, nsym_blocks = 0, nmerge_blocks = 0, sym_rows = 1;
    bool sym_mfma = false;             // symmetric problems run on K1e (matrix cores)
    bool sym_mfma_multi = false;       // ... and some of them have n2 > 2048 (multi-window instantiation)
    int mfma_form = 0;                 // ctx option "mfma_form" at plan creation (0/2 = K1f, 1 = K1e)
    bool exact_second = false;         // K1h: exact key tables (ctx option at plan creation); else the finalize kernel completes keys21 lazily
    bool post_fused = false;           // K1h / K1i throughput plans: merge + finalize + gates in ONE kernel, a workgroup per problem (k_post_fused)
    size_t post_lds = 0;               // ... its dynamic LDS: 8 bytes per column of the widest problem
    bool fused = false;                // K1f, one workgroup per problem: merge + ratio + mutual inside the scan kernel
    int merge_parts = 1;               // K1f: lanes per column in the partial merge (tall problems: many row blocks, few columns)
    bool col_split = false;            // K1f on a FEW LARGE problems: columns cut into ranges scanned as sub-problems
    DevBuf rowtmp;                     // ... their per-range row results (merged by the finalize kernel)
    bool split_post = false;           // ... and everything behind the scan in ONE kernel (k_split_post): the run is two launches
    bool split_post_ok = false;        // (what plan_build found; a gate applied by the finalize kernel switches split_post off)
    int32_t ndir = 0, ndir_blocks = 0; // non-mutual problems on the directed form of K1e
    bool dir_multi = false;
    SymDesc* d_dirs = nullptr; BlockDesc* d_dir_blocks = nullptr;
    DevBuf keys, counts, partials;
    DevBuf tail_counts;                // experiment builds (-DPLSLAM_MI_TAIL=1) with ctx option "scan_tail": a counter per symmetric problem
    bool scan_tail = false;            // ... the scan's last workgroup of a problem merges its column partials: no merge launch
    DevBuf tables;                     // all launch tables, packed, uploaded with ONE copy
    std::vector<char> staging;         // host image of `tables` (kept alive: the copy is async)
    HostBuf staging_pin;               // ... in pinned memory when pin_tables (the context's host-path plan)
    bool pin_tables = false;
    ScanDesc* d_scans = nullptr; SymDesc* d_syms = nullptr; ProblemDesc* d_probs = nullptr;
    BlockDesc *d_scan_blocks = nullptr, *d_sym_blocks = nullptr, *d_merge_blocks = nullptr, *d_fin_blocks = nullptr;
    int32_t** d_count_dst = nullptr;
    int32_t* d_counts_zero = nullptr;  // contiguous int32 counters zeroed by the first scan kernel
    bool scatter_counts = false;       // user n_matches pointers are not one contiguous array
    // optional last stage: the stereo gates over the L<->R tables of the batch (plslam_match_plan_add_stereo_gates)
    DevBuf gate_tables;
    std::vector<char> gate_staging;
    std::vector<ProblemDesc> h_probs;  // host image of d_probs (the gate stage patches ProblemDesc::gate)
    bool probs_in_place = false;       // d_probs IS the page-locked image
    plslam_stereo_gate_problem* d_gates = nullptr;
    BlockDesc* d_gate_blocks = nullptr;
    int32_t ngate_blocks = 0, ngates = 0;
    int32_t* d_gate_counts = nullptr;  // contiguous counters of the gate problems (or nullptr)
    plslam_plan_info info{};
    bool profiling = false;
    // a run on one stream, captured once and replayed as a HIP graph (latency plans: a few small kernels whose launch
    // overheads are the run; option "graph")
    bool small = false;                // fewer waves than the chip has SIMDs (plan_build)
    hipGraphExec_t graph_exec = nullptr;
    bool graph_failed = false;
    void drop_graph() { if (graph_exec) (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; graph_failed = false; }
    struct Ev { hipEvent_t e0, e1, e2; };
    std::vector<Ev> evs;
    // split runs (plslam_match_plan_run_split): the scan on one stream, everything behind it on another
    hipEvent_t scan_done = nullptr, post_done = nullptr;
    bool post_pending = false;
    // plslam_match_plan_step_gather: the gather of this plan's table (and the root's widening) is over
    hipEvent_t gather_done = nullptr;
    bool gather_pending = false;
    size_t ev_used = 0;
    double acc_scan_ms = 0, acc_fin_ms = 0;
    int64_t acc_runs = 0;
    void free_all()
    {
        keys.release(); counts.release(); partials.release(); tail_counts.release(); tables.release(); staging_pin.release();
        gate_tables.release(); rowtmp.release();
        for (auto& e : evs) { (void)hipEventDestroy(e.e0); (void)hipEventDestroy(e.e1); (void)hipEventDestroy(e.e2); }
        evs.clear();
        if (scan_done) (void)hipEventDestroy(scan_done);
        if (post_done) (void)hipEventDestroy(post_done);
        if (gather_done) (void)hipEventDestroy(gather_done);
        scan_done = post_done = gather_done = nullptr;
        post_pending = gather_pending = false;
        drop_graph();
    }
};

// n1_dev0 (internal; the map<->keyframe driver's one-synchronisation brute-force form): the row count of problem 0 lives on
// the device, probs[0].n1 is its upper bound.  Only as a two-launch column-split plan of ONE mutual problem (K1f + k_split_post,
// which read the count themselves); anything else returns PLSLAM_ENOTSUP and the caller takes its two-synchronisation form.
// the context-only half of that test (the options the two-launch column-split plan cannot honour): callers ask BEFORE they
// stage or enqueue anything for the one-synchronisation form (map2kf.hip)
bool plslam::ctx_takes_device_row_count(const plslam_ctx* ctx)
{
    return (ctx->scan_variant == PLSLAM_SCAN_AUTO || ctx->scan_variant == PLSLAM_SCAN_MFMA) &&
           (ctx->mfma_form == 0 || ctx->mfma_form == 2) && ctx->col_split != 1 && ctx->split_post != 1 && ctx->fuse != 2;
}

static int plan_build(plslam_ctx* ctx, const plslam_match_problem* probs, int32_t nprob,
                      plslam_match_plan* P, const int32_t* n1_dev0 = nullptr)
{
    PLSLAM_REQUIRE(nprob >= 0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(nprob == 0 || probs != nullptr, PLSLAM_EINVAL);
    P->ctx = ctx;
    P->nprob = nprob;

    // AUTO: mutual problems take the symmetric scan (one distance feeds both directions) -- on the
    // matrix cores (K1e) -- and
    // the others its directed form (row direction only).  A forced variant applies to every problem
    // (SYMMETRIC = the XOR+popcount form).  Measured, scan time per launch, C2 batches of 64 / 256 / 1024 /
    // 4096 pairs: K1e 0.12 / 0.45 / 1.69 / 6.0 ms, K1b(') 0.24 / 0.78 / 2.85 / 10.9 ms.
    // A plan too small to put one wave on every SIMD under those (e.g. ONE StVO::match call of the
    // SLAM loop) takes the wave-per-query scan instead: 16 queries per workgroup, train tile in LDS.
    int64_t thr_waves = 0;   // waves the throughput kernels would launch: one per 64 rows of d1
    for (int32_t i = 0; i < nprob; ++i) thr_waves += (probs[i].n1 + 63) / 64;
    const int64_t simds = (int64_t)ctx->prop.multiProcessorCount * 4;
    // A plan of a FEW LARGE problems (C3: one local map against one frame, 10 000 x 1500 + 2 000 x 200, mapHandler.cpp:532-752)
    // has too few 256-row blocks to fill the chip, but far too much work for the latency kernel (50 us there): the
    // matrix-core scan takes it with the COLUMNS cut into ranges, one workgroup per (row block, range).
    int64_t sym_evals = 0;
    for (int32_t i = 0; i < nprob; ++i)
        if (probs[i].n1 > 0 && probs[i].n2 > 0) sym_evals += (int64_t)probs[i].n1 * probs[i].n2;
    const bool small_plan = thr_waves < simds;
    P->small = small_plan;
    P->drop_graph();                   // (a rebuilt plan launches other tables)
    const bool split_auto = ctx->scan_variant == PLSLAM_SCAN_AUTO && small_plan && sym_evals >= (int64_t(6) << 20) &&
                            ctx->mfma_form != 1;
    const bool split_forced = ctx->col_split == 2 && ctx->mfma_form != 1 &&
                              (ctx->scan_variant == PLSLAM_SCAN_AUTO || ctx->scan_variant == PLSLAM_SCAN_MFMA);
    P->col_split = ctx->col_split != 1 && (split_auto || split_forced);
    if (n1_dev0) {
        const bool can = nprob == 1 && probs[0].mutual && !probs[0].keep_prior && probs[0].n1 > 0 && probs[0].n2 > 0 &&
                         ctx_takes_device_row_count(ctx);
        if (!can) return PLSLAM_ENOTSUP;
        P->col_split = true;
    }
    const bool use_wpq = ctx->scan_variant == PLSLAM_SCAN_WAVE_PER_QUERY ||
                         (ctx->scan_variant == PLSLAM_SCAN_AUTO && small_plan && !P->col_split);
    const bool allow_sym = !use_wpq &&
                           (ctx->scan_variant == PLSLAM_SCAN_AUTO || ctx->scan_variant == PLSLAM_SCAN_SYMMETRIC ||
                            ctx->scan_variant == PLSLAM_SCAN_MFMA);
    // mfma_form 3: a mutual problem runs as TWO DIRECTED matrix-core scans (d1 -> d2 and d2 -> d1; what the reference's two
    // knnMatch calls evaluate) -- no column direction, no partial table, no merge kernel
    const bool dpair = allow_sym && ctx->mfma_form == 3 &&
                       (ctx->scan_variant == PLSLAM_SCAN_MFMA || ctx->scan_variant == PLSLAM_SCAN_AUTO);
    auto is_sym = [&](const plslam_match_problem& p) { return allow_sym && !dpair && p.mutual && p.n1 > 0 && p.n2 > 0; };

    // sym_rows 0 = auto: 4 rows of d1 per lane (4x fewer column partials, slightly faster) once the
    // plan has enough 256-row waves for >= 6 full rounds of the chip (17 single-wave workgroups fit a
    // CU's LDS); below that the 4x coarser work units lose more to tail quantisation than they gain
    // (measured: 266k vs 320k pairs/s at 512 pairs, 347k vs 344k at 2048, 364k vs 347k at 4096).
    P->sym_mfma = allow_sym && (ctx->scan_variant == PLSLAM_SCAN_MFMA || ctx->scan_variant == PLSLAM_SCAN_AUTO);
    P->sym_mfma_multi = false;
    P->mfma_form = ctx->mfma_form;
    P->exact_second = ctx->exact_second != 0;
    P->dir_multi = false;
    // (set per scanned (sub-)problem while the tables are built)
    P->sym_rows = P->sym_mfma ? 4 : ctx->sym_rows;      // K1e uses the 256-row tables of K1b'
    if (P->sym_rows == 0) {
        int64_t waves4 = 0;
        for (int32_t i = 0; i < nprob; ++i)
            if (is_sym(probs[i])) waves4 += (probs[i].n1 + 255) / 256;
        P->sym_rows = waves4 >= 6 * 17 * (int64_t)ctx->prop.multiProcessorCount ? 4 : 1;
    }
    const bool k1f = P->sym_mfma && P->mfma_form != 1;   // K1f: column partials per 64-row block, 16-bit keys
    // Fused form (K1f only): one workgroup per problem walks all row blocks and finishes the problem (column merge, ratio
    // test, mutual check, count) -- ONE kernel per plan run, no merge / finalize kernels, no keys21 round trip.  Measured
    // at C2 / 4096 pairs per step: 4.05 ms against 3.48 + 0.52 ms unfused -- the merge's VALU work (+5 %), which the
    // separate merge kernel hides under its HBM time, and the serial tail of every workgroup cost what the two launches
    // cost -- so AUTO does not select it; "fuse" = 2 does (it needs many more problems than the chip has workgroup slots,
    // 3 per CU, or the 6x coarser work units lose to tail quantisation).  Mutual problems keep their merged column keys in
    // LDS: n2 <= PLSLAM_K1F_FUSED_MAX_N2.
    {
        int64_t nmf = 0;
        bool fits = true;
        for (int32_t i = 0; i < nprob; ++i) {
            if (probs[i].n1 <= 0 || probs[i].n2 <= 0) continue;
            ++nmf;
            if (probs[i].mutual && probs[i].n2 > PLSLAM_K1F_FUSED_MAX_N2) fits = false;
            if (probs[i].keep_prior) fits = false;          // the in-kernel finalize always writes every row
        }
        (void)nmf;
        P->fused = k1f && fits && ctx->fuse == 2 && !P->col_split && !dpair;
    }
    const int rpp = sym_rows_per_partial(P->sym_rows);   // a-rows per column partial
    const int rps = sym_rows_per_block(P->sym_rows);     // a-rows per workgroup of the symmetric scan
    // column split (K1f only): ranges of `cstep` columns per problem so that the launch has about 3 workgroups per CU,
    // at least 4 tiles (128 columns) per range
    int64_t mf_row_blocks = 0;
    for (int32_t i = 0; i < nprob; ++i)
        if (probs[i].n1 > 0 && probs[i].n2 > 0) mf_row_blocks += (probs[i].n1 + 255) / 256;
    P->col_split = P->col_split && k1f && mf_row_blocks > 0 && !dpair;
    // AUTO form: K1h for throughput plans; a column-split plan (a few large problems, e.g. C3's one map against one frame) is
    // latency-bound -- 4-5 tiles per workgroup -- and K1f's lighter per-workgroup prologue / row finish wins there
    // (measured at C3: 22.4 us per run against 27.4 us)
    if (ctx->mfma_form == 0 && P->col_split) P->mfma_form = 2;
    auto split_of = [&](const plslam_match_problem& p, int32_t* cstep) -> int32_t {
        *cstep = 0;
        if (!P->col_split || p.n1 <= 0 || p.n2 <= 0) return 1;
        // (options "split_target": workgroups per CU the split aims at, 0 = 3; "split_min_tiles": tiles per range at least, 0 = 4)
        const int64_t target = (ctx->split_target > 0 ? ctx->split_target : 3) * (int64_t)ctx->prop.multiProcessorCount;
        const int64_t want = (target + mf_row_blocks - 1) / mf_row_blocks;
        const int32_t tiles = (p.n2 + 31) / 32;
        int32_t per = (int32_t)((tiles + want - 1) / want);
        const int32_t min_tiles = ctx->split_min_tiles > 0 ? ctx->split_min_tiles : 4;
        if (per < min_tiles) per = min_tiles;
        const int32_t ns = (tiles + per - 1) / per;
        if (ns <= 1) return 1;
        *cstep = per * 32;
        return ns;
    };
    {   // partial merge: share a column among several lanes when the plan has long columns and too few of them
        int64_t cols = 0;
        int32_t max_nwb = 0;
        for (int32_t i = 0; i < nprob; ++i)
            if (is_sym(probs[i])) { cols += probs[i].n2; max_nwb = std::max(max_nwb, (probs[i].n1 + 63) / 64); }
        const int64_t lanes = 64 * 4 * (int64_t)ctx->prop.multiProcessorCount * 4;      // ~4 waves per SIMD in flight
        P->merge_parts = (max_nwb >= 64 && cols * 16 <= lanes) ? 16 : (max_nwb >= 32 && cols * 4 <= lanes) ? 4 : 1;
    }
    // K1h's column partials: one word per (256-row block, column slot) -- two with exact key tables; K1f's: one word per
    // (64-row block, column slot); rows padded to 256 slots
    const bool h_parts = P->sym_mfma && mfma_form_is_h(P->mfma_form) && !P->fused;
    const int mcols = h_parts ? merge_fix16_cols(P->merge_parts) : merge_partials16_cols(P->merge_parts);
    // column partials in units of two words
    auto part_units = [&](int32_t n1, int32_t n2) -> int64_t {
        if (h_parts) return (int64_t)((n1 + 255) / 256) * ((n2 + 255) / 256) * (P->exact_second ? 256 : 128);
        return (int64_t)((n1 + 63) / 64) * ((n2 + 255) / 256) * 128;
    };
    int64_t rows = 0, part_rows = 0, tmp_rows = 0;
    for (int32_t i = 0; i < nprob; ++i) {
        const plslam_match_problem& p = probs[i];
        PLSLAM_REQUIRE(p.n1 >= 0 && p.n2 >= 0, PLSLAM_EINVAL);
        PLSLAM_REQUIRE(p.n1 == 0 || p.d1 != nullptr, PLSLAM_EINVAL);
        PLSLAM_REQUIRE(p.n2 == 0 || p.d2 != nullptr, PLSLAM_EINVAL);
        PLSLAM_REQUIRE(p.n1 == 0 || p.matches_12 != nullptr, PLSLAM_EINVAL);
        PLSLAM_REQUIRE((reinterpret_cast<uintptr_t>(p.d1) & 3) == 0, PLSLAM_EINVAL);
        PLSLAM_REQUIRE((reinterpret_cast<uintptr_t>(p.d2) & 3) == 0, PLSLAM_EINVAL);
        PLSLAM_REQUIRE(p.n2 <= PLSLAM_MAX_TRAIN_ROWS, PLSLAM_ERANGE);
        PLSLAM_REQUIRE(!p.mutual || p.n1 <= PLSLAM_MAX_TRAIN_ROWS, PLSLAM_ERANGE);
        rows += p.n1 + (p.mutual ? p.n2 : 0);
        // column partials, in units of two words: K1e one (best, second) pair per (256-row block, column); K1f one word
        // per (64-row block, column) with the rows padded to 256 columns
        if (is_sym(p)) {
            int32_t cstep = 0;
            const int32_t ns = split_of(p, &cstep);
            if (ns > 1) {
                for (int32_t s_ = 0; s_ < ns; ++s_)
                    part_rows += part_units(p.n1, std::min(cstep, p.n2 - s_ * cstep));
            } else {
                part_rows += k1f ? part_units(p.n1, p.n2) : (int64_t)((p.n1 + rpp - 1) / rpp) * p.n2;
            }
        }
        if (P->sym_mfma && p.n1 > 0 && p.n2 > 0) {
            int32_t cstep = 0;
            const int32_t ns = split_of(p, &cstep);
            if (ns > 1) tmp_rows += (int64_t)ns * p.n1;
        }
    }
    PLSLAM_REQUIRE(rows < (int64_t(1) << 31), PLSLAM_ERANGE);

    P->variant = use_wpq ? PLSLAM_SCAN_WAVE_PER_QUERY
                         : (allow_sym ? PLSLAM_SCAN_SYMMETRIC : PLSLAM_SCAN_LANE_PER_QUERY);
    P->block_threads = use_wpq ? 256 : (ctx->scan_block ? ctx->scan_block : 256);
    const int directed_variant = use_wpq ? PLSLAM_SCAN_WAVE_PER_QUERY : PLSLAM_SCAN_LANE_PER_QUERY;
    const int rpb = scan_rows_per_block(directed_variant, P->block_threads);

    int r = P->keys.reserve(sizeof(uint32_t) * 2 * (size_t)(rows > 0 ? rows : 1));
    if (r) return r;
    r = P->partials.reserve(sizeof(uint32_t) * 2 * (size_t)(part_rows > 0 ? part_rows : 1));
    if (r) return r;
    if (tmp_rows > 0 && (r = P->rowtmp.reserve(sizeof(uint32_t) * 2 * (size_t)tmp_rows))) return r;
    uint32_t* d_tmp = P->rowtmp.as<uint32_t>();
    uint32_t* d_keys = P->keys.as<uint32_t>();
    uint32_t* d_part = P->partials.as<uint32_t>();
    // #matches counters: accumulated with atomics by the finalize kernel, zeroed by the scan
    // kernel.  If the caller's n_matches pointers form one contiguous array, count in place.
    bool contiguous = nprob > 0, any_user = false;
    for (int32_t i = 0; i < nprob; ++i) {
        any_user = any_user || probs[i].n_matches != nullptr;
        contiguous = contiguous && probs[i].n_matches != nullptr &&
                     probs[i].n_matches == probs[0].n_matches + i;
    }
    int32_t* d_counts = nullptr;
    if (contiguous) {
        d_counts = probs[0].n_matches;
    } else {
        r = P->counts.reserve(sizeof(int32_t) * (size_t)(nprob > 0 ? nprob : 1));
        if (r) return r;
        d_counts = P->counts.as<int32_t>();
    }
    P->scatter_counts = any_user && !contiguous;
    P->d_counts_zero = d_counts;
    P->ncounts = nprob;

    // A column-split plan of mutual K1f problems runs in TWO launches: k_split_post merges the column partials and decides the
    // matches from the column side (hamming_mfma_g.hip); rows without a match keep the -1 the scan's first column range
    // writes.  Not with kept entries (keep_prior: a rejected row's old entry goes through the consistency loop) and not with a
    // stereo gate behind the table (add_stereo_gates switches back to merge + finalize).  Option "split_post": 0 = auto, 1 = never.
    {
        bool ok = P->col_split && k1f && !h_parts && !P->fused && ctx->split_post != 1 && nprob > 0;
        for (int32_t i = 0; ok && i < nprob; ++i)
            ok = is_sym(probs[i]) && !probs[i].keep_prior;
        P->split_post = P->split_post_ok = ok;
        if (n1_dev0 && !ok) return PLSLAM_ENOTSUP;
    }
    std::vector<ScanDesc> scans;
    std::vector<int32_t> scan_problem;   // scans[k] belongs to problem scan_problem[k]
    std::vector<SymDesc> syms, dirs;
    std::vector<ProblemDesc> pds;
    std::vector<BlockDesc> sblocks, fblocks, yblocks, mblocks, dblocks;
    int64_t key_row = 0, part_row = 0, tmp_row = 0, evals = 0, devals = 0, abytes = 0;
    std::vector<int32_t*> user_counts((size_t)nprob, nullptr);
    for (int32_t i = 0; i < nprob; ++i) {
        const plslam_match_problem& p = probs[i];
        ProblemDesc pd{};
        pd.n1 = p.n1; pd.n2 = p.n2; pd.nnr = p.nnr; pd.mutual = p.mutual ? 1 : 0;
        pd.keep_prior = p.keep_prior ? 1 : 0;
        pd.matches_12 = p.matches_12;
        pd.n_matches = d_counts + i;
        user_counts[i] = p.n_matches;
        uint32_t* k12 = d_keys + 2 * key_row;
        key_row += p.n1;
        uint32_t* k21 = p.mutual ? d_keys + 2 * key_row : nullptr;
        if (p.mutual) key_row += p.n2;
        pd.keys12 = k12;
        pd.keys21 = k21;
        pd.d1 = p.d1; pd.d2 = p.d2;
        pd.gate = -1;
        const bool mf_path = P->sym_mfma && p.n1 > 0 && p.n2 > 0;    // this problem runs on K1e / K1f
        // (finalize blocks start at multiples of 256 rows and run all 256 lanes: the lazy completion of K1h's / K1i's column keys
        // rotates the neighbouring rows' keys through DPP within aligned groups of 16 lanes -- hamming.hip, finalize_row)
        static_assert(256 % 16 == 0, "a finalize block must hold whole groups of 16 rows");
        if (!(P->fused && mf_path))
            for (int32_t r0 = 0; r0 < p.n1; r0 += 256) fblocks.push_back({i, r0});
        int32_t cstep = 0;
        const int32_t nsplit = mf_path ? split_of(p, &cstep) : 1;
        if (nsplit > 1) {
            // one sub-problem per column range: its own row results (relative column indices, merged by the finalize
            // kernel), its own partial area, its slice of keys21
            pd.split_tmp = d_tmp + 2 * tmp_row; pd.keys12_out = k12; pd.nsplit = nsplit; pd.cstep = cstep;
            pd.lazy21 = p.mutual && P->sym_mfma && mfma_form_is_h(P->mfma_form) && !P->fused && !P->exact_second;
            std::vector<SymDesc>& dst = p.mutual ? syms : dirs;
            std::vector<BlockDesc>& dstb = p.mutual ? yblocks : dblocks;
            for (int32_t s_ = 0; s_ < nsplit; ++s_) {
                const int32_t c0 = s_ * cstep, n2s = std::min(cstep, p.n2 - c0);
                SymDesc y{};
                y.flags = ctx->exact_second ? 1 : 0;
                y.a = p.d1; y.b = p.d2 + (size_t)c0 * 32;
                y.keys12 = d_tmp + 2 * (tmp_row + (int64_t)s_ * p.n1);
                y.n1 = p.n1; y.n2 = n2s;
                if (P->split_post) { y.mutual = i + 1; y.matches_12 = s_ == 0 ? p.matches_12 : nullptr; y.n1_dev = n1_dev0; }
                if (p.mutual) {
                    y.keys21 = k21 + 2 * (size_t)c0;
                    y.part21 = d_part + 2 * part_row;
                    y.n_iblk = (p.n1 + rpp - 1) / rpp;
                    part_row += part_units(p.n1, n2s);
                    // (K1h's merge walks column SLOTS, 32 per tile: the table covers n2 rounded up to a tile)
                    for (int32_t j0 = 0; j0 < ((n2s + 31) & ~31); j0 += mcols) mblocks.push_back({(int32_t)dst.size(), j0});
                }
                for (int32_t r0 = 0; r0 < p.n1; r0 += 256) dstb.push_back({(int32_t)dst.size(), r0});
                if (n2s > 2048) (p.mutual ? P->sym_mfma_multi : P->dir_multi) = true;
                dst.push_back(y);
            }
            tmp_row += (int64_t)nsplit * p.n1;
            evals += (int64_t)p.n1 * p.n2;
            devals += (p.mutual ? 2LL : 1LL) * p.n1 * p.n2;
            abytes += p.mutual ? 2 * 32LL * (p.n1 + p.n2) + 16LL * (p.n1 + p.n2) : 32LL * (p.n1 + p.n2) + 16LL * p.n1;
        } else if (is_sym(p)) {
            pd.lazy21 = P->sym_mfma && mfma_form_is_h(P->mfma_form) && !P->fused && !P->exact_second;
            SymDesc y{};
            y.flags = ctx->exact_second ? 1 : 0;
            y.a = p.d1; y.b = p.d2; y.keys12 = k12; y.keys21 = k21;
            y.part21 = d_part + 2 * part_row;
            pd.part21 = y.part21;
            y.n1 = p.n1; y.n2 = p.n2; y.n_iblk = (p.n1 + rpp - 1) / rpp;
            part_row += k1f ? part_units(p.n1, p.n2) : (int64_t)y.n_iblk * p.n2;
            if (P->split_post) { y.mutual = i + 1; y.matches_12 = p.matches_12; y.n1_dev = n1_dev0; }     // (a problem of one column range)
            if (P->fused) {
                y.mutual = 1; y.matches_12 = p.matches_12; y.n_matches = pd.n_matches; y.nnr = p.nnr;
                yblocks.push_back({(int32_t)syms.size(), 0});
            } else {
                for (int32_t r0 = 0; r0 < p.n1; r0 += rps) yblocks.push_back({(int32_t)syms.size(), r0});
                for (int32_t c0 = 0; c0 < (k1f ? (p.n2 + 31) & ~31 : p.n2); c0 += (k1f ? mcols : 256)) mblocks.push_back({(int32_t)syms.size(), c0});
            }
            if (P->sym_mfma && p.n2 > 2048) P->sym_mfma_multi = true;
            syms.push_back(y);
            evals += (int64_t)p.n1 * p.n2;
            devals += 2LL * p.n1 * p.n2;
            abytes += 2 * 32LL * (p.n1 + p.n2) + 16LL * (p.n1 + p.n2);
        } else if (P->sym_mfma && (!p.mutual || dpair) && p.n1 > 0 && p.n2 > 0) {
            // non-mutual problem on the matrix cores: the directed form of K1e (row direction only); mfma_form 3: also the two
            // directions of a mutual problem
            for (int dir = 0; dir < (p.mutual ? 2 : 1); ++dir) {
                SymDesc y{};
                y.flags = ctx->exact_second ? 1 : 0;
                y.a = dir ? p.d2 : p.d1; y.b = dir ? p.d1 : p.d2; y.keys12 = dir ? k21 : k12; y.keys21 = nullptr; y.part21 = nullptr;
                y.n1 = dir ? p.n2 : p.n1; y.n2 = dir ? p.n1 : p.n2; y.n_iblk = 0;
                if (P->fused && !p.mutual) {
                    y.mutual = 0; y.matches_12 = p.matches_12; y.n_matches = pd.n_matches; y.nnr = p.nnr;
                    dblocks.push_back({(int32_t)dirs.size(), 0});
                } else {
                    for (int32_t r0 = 0; r0 < y.n1; r0 += 256) dblocks.push_back({(int32_t)dirs.size(), r0});
                }
                if (y.n2 > 2048) P->dir_multi = true;
                dirs.push_back(y);
                evals += (int64_t)p.n1 * p.n2;
                devals += (int64_t)p.n1 * p.n2;
                abytes += 32LL * (p.n1 + p.n2) + 16LL * y.n1;
            }
        } else {
            if (p.n1 > 0) {
                ScanDesc sc{p.d1, p.d2, k12, p.n1, p.n2};
                for (int32_t r0 = 0; r0 < p.n1; r0 += rpb) sblocks.push_back({(int32_t)scans.size(), r0});
                scans.push_back(sc);
                scan_problem.push_back(i);
                evals += (int64_t)p.n1 * p.n2;
                devals += (int64_t)p.n1 * p.n2;
                abytes += 32LL * (p.n1 + p.n2) + 16LL * p.n1;
            }
            if (p.mutual && p.n2 > 0 && p.n1 > 0) {
                ScanDesc sc{p.d2, p.d1, k21, p.n2, p.n1};
                for (int32_t r0 = 0; r0 < p.n2; r0 += rpb) sblocks.push_back({(int32_t)scans.size(), r0});
                scans.push_back(sc);
                scan_problem.push_back(i);
                evals += (int64_t)p.n1 * p.n2;
                devals += (int64_t)p.n1 * p.n2;
                abytes += 32LL * (p.n1 + p.n2) + 16LL * p.n2;
            }
        }
        pds.push_back(pd);
    }
    // XCD-striped, longest-first block tables.  Hardware places workgroup b on XCD b % 8 and
    // dispatches in increasing b, and the scan kernels read table entry (b % 8) * L + b / 8, so row x
    // of the table (L entries) is XCD x's work in dispatch order.  Blocks are dealt out in GROUPS
    // (the blocks of one problem, at most 8: they stream the same descriptor sets, so they should
    // share one XCD's L2 at the same time), groups in descending train-stream length so that every
    // XCD runs its long blocks (ORB) first and the short ones (LBD) fill the drain phase.  Rows are
    // padded to equal length with no-op entries (item = -1).
    struct Group { int64_t cost; int32_t first, count; };
    // (Dealing the SHORT groups -- the LBD problems of a stereo batch, seven tiles of mostly memory latency -- evenly among the
    // long ones instead of running them together at the end measured 1-3 % SLOWER, 2.62-2.70 against 2.60-2.64 ms per
    // 4096-pair scan: round 4, option removed.)
    auto stripe = [](std::vector<BlockDesc>& blocks, std::vector<Group> groups) {
        std::stable_sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) { return a.cost > b.cost; });
        std::vector<BlockDesc> rows[8];
        size_t x = 0;
        for (const Group& g : groups) {
            // next XCD round-robin, but prefer the currently shortest row among the next candidates
            size_t best = x;
            for (size_t t = 0; t < 8; ++t) {
                const size_t c = (x + t) & 7;
                if (rows[c].size() < rows[best].size()) best = c;
            }
            for (int32_t k = 0; k < g.count; ++k) rows[best].push_back(blocks[(size_t)g.first + k]);
            x = (best + 1) & 7;
        }
        size_t L = 0;
        for (auto& r : rows) L = std::max(L, r.size());
        std::vector<BlockDesc> out(8 * L, BlockDesc{-1, 0});
        for (size_t c = 0; c < 8; ++c)
            for (size_t k = 0; k < rows[c].size(); ++k) out[c * L + k] = rows[c][k];
        blocks.swap(out);
    };
    // group_cap: measured on MI355X (512 / 2048 pairs per step): 1 -> 321k / 343k pairs/s with 2.55 GB of
    // HBM reads per 2048-pair launch; 2 -> 317k / 343k; >= 3 -> 310k / 335k (many waves streaming the same
    // rows at the same moment contend for the same cache lines) with 0.67 GB of reads.  Default 2.
    // For the 4-rows-per-lane kernel the cap is speed-neutral (364.7k / 364.9k / 364.4k / 365.9k pairs/s at
    // cap 1 / 2 / 3 / 6), so its groups keep a whole problem together.  0 = auto.
    const size_t group_cap = ctx->group_cap > 0 ? (size_t)ctx->group_cap : (P->sym_rows == 4 ? 8 : 2);
    auto groups_of = [group_cap](const std::vector<BlockDesc>& blocks, auto key_of, auto cost_of) {
        std::vector<Group> gs;
        for (size_t i = 0; i < blocks.size();) {
            size_t j = i;
            while (j < blocks.size() && j - i < group_cap && key_of(blocks[j]) == key_of(blocks[i])) ++j;
            gs.push_back({cost_of(blocks[i]), (int32_t)i, (int32_t)(j - i)});
            i = j;
        }
        return gs;
    };
    // (fused: one entry per problem, so a group is one workgroup and its cost the whole distance matrix)
    if (!yblocks.empty())
        stripe(yblocks, groups_of(yblocks, [](const BlockDesc& b) { return b.item; },
                                  [&](const BlockDesc& b) { return (int64_t)syms[b.item].n2 * (P->fused ? syms[b.item].n1 : 1); }));
    if (!dblocks.empty())
        stripe(dblocks, groups_of(dblocks, [](const BlockDesc& b) { return b.item; },
                                  [&](const BlockDesc& b) { return (int64_t)dirs[b.item].n2 * (P->fused ? dirs[b.item].n1 : 1); }));
    if (!use_wpq && !sblocks.empty())   // the two directed scans of a mutual problem are adjacent: same group key
        stripe(sblocks, groups_of(sblocks, [&](const BlockDesc& b) { return scan_problem[b.item]; },
                                  [&](const BlockDesc& b) { return (int64_t)scans[b.item].nt; }));

    // The stage behind the scan as ONE kernel (k_post_fused): every problem a mutual one on K1h / K1i with lazy column keys,
    // few row blocks, columns that fit the workgroup's LDS (option "post_fuse": 0 = auto, 1 = never, 2 = whenever the plan is
    // eligible).  AUTO does NOT select it: measured at C2 / 4096 pairs it moves 0.8 GB less per step (no merged column table
    // written and gathered back) but takes 0.335 ms against the separate kernels' 0.276 ms, and the step 2.78 against 2.71 ms --
    // a workgroup per problem is a chain of round trips (partials -> LDS -> rows -> gates) with 2 048 problems in flight,
    // where the separate kernels keep 8x as many independent lanes busy; the scan leaves no free registers beside it
    // (3 x 168 of 512 per lane), so whatever runs behind it displaces scan workgroups one for one and only its own
    // duration counts.
    {
        bool ok = h_parts && !P->exact_second && !P->col_split && ctx->post_fuse != 1 && nprob > 0 && scans.empty() && dirs.empty() &&
                  (int32_t)syms.size() == nprob;
        int32_t max_n2 = 0;
        for (int32_t i = 0; ok && i < nprob; ++i) {
            ok = pds[i].lazy21 && pds[i].nsplit <= 1 && pds[i].part21 != nullptr && probs[i].n2 <= POST_FUSED_MAX_N2 &&
                 (probs[i].n1 + 255) / 256 <= POST_FUSED_MAX_ROW_BLOCKS;
            max_n2 = std::max(max_n2, probs[i].n2);
        }
        ok = ok && ctx->post_fuse == 2;
        P->post_fused = ok;
        P->post_lds = ok ? sizeof(uint32_t) * 2 * (size_t)((max_n2 + 63) & ~63) : 0;
    }
    P->nscan = (int32_t)scans.size();
    P->nscan_blocks = (int32_t)sblocks.size();
    // option "post_xcd" = 2: the finalize table dealt to the XCDs PROBLEM BY PROBLEM (a problem's row blocks gather its column keys
    // through one L2, and consecutive problems sit on different XCDs, so the eight of them sweep memory together): rows of equal
    // length, no-op entries (item = -1) behind the short ones; table entry (b % 8) * row + b / 8 is workgroup b's
    P->fin_row = 0;
    if (ctx->post_xcd == 2 && fblocks.size() >= 64) {
        std::vector<BlockDesc> rows[8];
        size_t x = 0;
        for (size_t i = 0; i < fblocks.size();) {
            size_t j = i;
            while (j < fblocks.size() && fblocks[j].item == fblocks[i].item) ++j;
            size_t best = x;
            for (size_t t = 0; t < 8; ++t) {
                const size_t c = (x + t) & 7;
                if (rows[c].size() < rows[best].size()) best = c;
            }
            for (size_t k = i; k < j; ++k) rows[best].push_back(fblocks[k]);
            x = (best + 1) & 7;
            i = j;
        }
        size_t L = 0;
        for (auto& r_ : rows) L = std::max(L, r_.size());
        std::vector<BlockDesc> out(8 * L, BlockDesc{-1, 0});
        for (size_t c = 0; c < 8; ++c)
            for (size_t k = 0; k < rows[c].size(); ++k) out[c * L + k] = rows[c][k];
        fblocks.swap(out);
        P->fin_row = (int32_t)L;
    }
    P->nfin_blocks = (int32_t)fblocks.size();
    P->nsym = (int32_t)syms.size();
    P->nsym_blocks = (int32_t)yblocks.size();
    P->ndir = (int32_t)dirs.size();
    P->ndir_blocks = (int32_t)dblocks.size();
    P->nmerge_blocks = (int32_t)mblocks.size();
    P->scan_tail = ctx->scan_tail && k1i_tail_built() && P->nsym > 0 && P->sym_mfma && mfma_form_is_h(P->mfma_form) && !P->fused &&
                   !P->exact_second && P->merge_parts == 1 && !P->post_fused && !P->split_post && !P->col_split;
    if (P->scan_tail) {
        if ((r = P->tail_counts.reserve(sizeof(int32_t) * (size_t)P->nsym))) return r;
        PLSLAM_HIP_CHECK(hipMemsetAsync(P->tail_counts.p, 0, sizeof(int32_t) * (size_t)P->nsym, ctx->stream));   // (the kernel leaves them zero)
        PLSLAM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    P->info.distance_evals = evals;      // executed
    P->info.directed_evals = devals;     // what two directed knnMatch calls per mutual problem evaluate
    P->info.algorithmic_bytes = abytes;  // 32(Q+T)+16Q per DIRECTED scan, however executed
    P->info.n_scans = P->nscan + 2 * P->nsym + P->ndir;
    P->info.scan_blocks = P->nscan_blocks + P->nsym_blocks + P->ndir_blocks;
    P->info.scan_variant = P->nsym ? (P->sym_mfma ? PLSLAM_SCAN_MFMA : PLSLAM_SCAN_SYMMETRIC)
                                   : (P->ndir ? PLSLAM_SCAN_MFMA : directed_variant);
    P->info.scan_block_threads = P->nsym ? (P->sym_rows == 4 && !P->sym_mfma ? 64 : 256)
                                         : (P->ndir ? 256 : P->block_threads);

    // pack every launch table into one host image and upload it with a single copy
    struct Piece { const void* src; size_t bytes; size_t off; };
    Piece pc[10] = {{scans.data(), scans.size() * sizeof(ScanDesc), 0},
                   {syms.data(), syms.size() * sizeof(SymDesc), 0},
                   {pds.data(), pds.size() * sizeof(ProblemDesc), 0},
                   {sblocks.data(), sblocks.size() * sizeof(BlockDesc), 0},
                   {yblocks.data(), yblocks.size() * sizeof(BlockDesc), 0},
                   {mblocks.data(), mblocks.size() * sizeof(BlockDesc), 0},
                   {fblocks.data(), fblocks.size() * sizeof(BlockDesc), 0},
                   {user_counts.data(), P->scatter_counts ? user_counts.size() * sizeof(int32_t*) : 0, 0},
                   {dirs.data(), dirs.size() * sizeof(SymDesc), 0},
                   {dblocks.data(), dblocks.size() * sizeof(BlockDesc), 0}};
    size_t total = 0;
    for (Piece& x : pc) { x.off = total; total += (x.bytes + 255) & ~size_t(255); }
    if (total == 0) total = 256;
    char* stg = nullptr;
    if (P->pin_tables && total <= (size_t(1) << 20)) {
        if ((r = P->staging_pin.reserve(total))) return r;
        stg = P->staging_pin.as<char>();
    } else {
        P->staging.resize(total);
        stg = P->staging.data();
    }
    for (const Piece& x : pc)
        if (x.bytes) memcpy(stg + x.off, x.src, x.bytes);
    // Small plans of the host-pointer path: the kernels read their launch tables (a few hundred bytes per workgroup, once)
    // straight from the page-locked image -- one copy-engine command less in front of the first kernel.  (Those callers
    // synchronise the stream before the context builds its next plan, so the image is not rewritten under a kernel.)
    char* base = nullptr;
    if (stg == P->staging_pin.p && total <= (size_t(1) << 14)) base = static_cast<char*>(mapped_device_pointer(stg));
    const bool tables_in_place = base != nullptr;
    if (!tables_in_place) {
        if ((r = P->tables.reserve(total))) return r;
        base = P->tables.as<char>();
    }
    P->d_scans = reinterpret_cast<ScanDesc*>(base + pc[0].off);
    P->d_syms = reinterpret_cast<SymDesc*>(base + pc[1].off);
    P->d_probs = reinterpret_cast<ProblemDesc*>(base + pc[2].off);
    P->d_scan_blocks = reinterpret_cast<BlockDesc*>(base + pc[3].off);
    P->d_sym_blocks = reinterpret_cast<BlockDesc*>(base + pc[4].off);
    P->d_merge_blocks = reinterpret_cast<BlockDesc*>(base + pc[5].off);
    P->d_fin_blocks = reinterpret_cast<BlockDesc*>(base + pc[6].off);
    P->d_count_dst = reinterpret_cast<int32_t**>(base + pc[7].off);
    P->d_dirs = reinterpret_cast<SymDesc*>(base + pc[8].off);
    P->d_dir_blocks = reinterpret_cast<BlockDesc*>(base + pc[9].off);
    P->h_probs = pds;
    P->probs_in_place = tables_in_place;
    P->ngate_blocks = 0;                 // a rebuilt plan (the context's host-path plan) starts without a gate stage
    P->ngates = 0;

    // P->staging outlives the copy (it is a member), so no synchronisation is needed here; the
    // copy is ordered before the kernels of plan_run when they use the same stream, and the public
    // plan_create synchronises once so that any stream may be used afterwards.
    if (!tables_in_place) PLSLAM_HIP_CHECK(hipMemcpyAsync(base, stg, total, hipMemcpyHostToDevice, ctx->stream));
    return PLSLAM_OK;
}

// s: the scan kernel(s); sp: the stages behind them (merge of the column partials, finalize, gates, count scatter).  sp == s
// is the plain run.  With two streams the stages behind the scan of one run overlap the scan of the NEXT run on `s` (another
// plan, or this one: its next scan waits for this run's last stage, which reads what that scan overwrites): the scan is
// bound by instruction issue, the stages behind it by HBM, and a workgroup slot the scan frees is taken by either.
static int plan_run(plslam_match_plan* P, hipStream_t s, hipStream_t sp)
{
    const bool split = sp != s;
    if (split) {
        if (!P->scan_done) PLSLAM_HIP_CHECK(hipEventCreateWithFlags(&P->scan_done, hipEventDisableTiming));
        if (!P->post_done) PLSLAM_HIP_CHECK(hipEventCreateWithFlags(&P->post_done, hipEventDisableTiming));
    }
    if (P->post_pending) {            // an earlier split run: its last stage must be over before this scan rewrites its input
        PLSLAM_HIP_CHECK(hipStreamWaitEvent(s, P->post_done, 0));
        P->post_pending = false;
    }
    plslam_match_plan::Ev* ev = nullptr;
    if (P->profiling) {
        if (P->ev_used == P->evs.size()) {
            plslam_match_plan::Ev e{};
            PLSLAM_HIP_CHECK(hipEventCreate(&e.e0));
            PLSLAM_HIP_CHECK(hipEventCreate(&e.e1));
            PLSLAM_HIP_CHECK(hipEventCreate(&e.e2));
            P->evs.push_back(e);
        }
        ev = &P->evs[P->ev_used++];
        PLSLAM_HIP_CHECK(hipEventRecord(ev->e0, s));
    }
    // the first scan kernel that runs zeroes the #matches counters
    int r;
    bool zeroed = false;               // the first scan kernel that runs zeroes the #matches counters
    if (P->fused) {
        // fused problems STORE their counts from inside the scan kernel, so the counters cannot be zeroed by that kernel's
        // first workgroup: clear them ahead of it (only problems without rows or columns keep the zero)
        PLSLAM_HIP_CHECK(hipMemsetAsync(P->d_counts_zero, 0, sizeof(int32_t) * (size_t)P->ncounts, s));
        zeroed = true;
    }
    if (P->nsym_blocks > 0) {
        r = P->sym_mfma ? launch_scan_mfma_form(P->mfma_form, P->d_syms, P->d_sym_blocks, P->nsym_blocks, P->d_counts_zero,
                                                zeroed ? 0 : P->ncounts, P->sym_mfma_multi, false, s, P->fused,
                                                P->scan_tail ? P->tail_counts.as<int32_t>() : nullptr)
                        : launch_scan_sym(P->sym_rows, P->d_syms, P->d_sym_blocks,
                                          P->nsym_blocks, P->d_counts_zero, P->ncounts, s);
        if (r) return r;
        zeroed = true;
    }
    if (P->ndir_blocks > 0) {
        r = launch_scan_mfma_form(P->mfma_form, P->d_dirs, P->d_dir_blocks, P->ndir_blocks, P->d_counts_zero,
                                  zeroed ? 0 : P->ncounts, P->dir_multi, true, s, P->fused);
        if (r) return r;
        zeroed = true;
    }
    if (P->nscan_blocks > 0 || !zeroed) {
        r = launch_scan(P->ctx, P->variant == PLSLAM_SCAN_WAVE_PER_QUERY ? PLSLAM_SCAN_WAVE_PER_QUERY
                                                                         : PLSLAM_SCAN_LANE_PER_QUERY,
                        P->block_threads, P->d_scans,
                        P->d_scan_blocks, P->nscan_blocks, P->d_counts_zero,
                        zeroed ? 0 : P->ncounts, s);
        if (r) return r;
    }
    if (ev) PLSLAM_HIP_CHECK(hipEventRecord(ev->e1, s));   // e0..e1 = the scan kernel(s) alone
    if (split) {
        PLSLAM_HIP_CHECK(hipEventRecord(P->scan_done, s));
        PLSLAM_HIP_CHECK(hipStreamWaitEvent(sp, P->scan_done, 0));
        s = sp;
    }

    if (P->split_post) {
        r = launch_split_post(P->d_syms, P->d_merge_blocks, P->nmerge_blocks, P->merge_parts, P->d_probs, s);
        if (r) return r;
    } else if (P->post_fused) {
        if (P->ngates > 0 && P->d_gate_counts) PLSLAM_HIP_CHECK(hipMemsetAsync(P->d_gate_counts, 0, sizeof(int32_t) * (size_t)P->ngates, s));
        r = launch_post_fused(P->d_probs, P->nprob, P->ngates > 0 ? P->d_gates : nullptr, P->post_lds, s);
        if (r) return r;
    } else {
    r = P->scan_tail ? PLSLAM_OK          // (the scan's last workgroup per problem has written keys21)
        : P->sym_mfma && mfma_form_is_h(P->mfma_form) && !P->fused
            ? launch_merge_fix16(P->d_syms, P->d_merge_blocks, P->nmerge_blocks, P->merge_parts, P->exact_second, s, split ? P->ctx->post_workgroups : 0)
            : P->sym_mfma && P->mfma_form != 1 ? launch_merge_partials16(P->d_syms, P->d_merge_blocks, P->nmerge_blocks, P->merge_parts, s)
                                               : launch_merge_partials(P->d_syms, P->d_merge_blocks, P->nmerge_blocks, s);
    if (r) return r;
    // the gate stage: its counters are cleared first; gates over the plan's own tables run inside the finalize kernel
    if (P->ngates > 0 && P->d_gate_counts) PLSLAM_HIP_CHECK(hipMemsetAsync(P->d_gate_counts, 0, sizeof(int32_t) * (size_t)P->ngates, s));
    r = launch_finalize(P->d_probs, P->d_fin_blocks, P->nfin_blocks, P->ngates > 0 ? P->d_gates : nullptr, s, split ? P->ctx->post_workgroups : 0, P->ctx->post_xcd == 1, P->fin_row);
    if (r) return r;
    }
    if (P->ngate_blocks > 0) {
        r = launch_stereo_gates(P->d_gates, P->d_gate_blocks, P->ngate_blocks, s);
        if (r) return r;
    }
    if (ev) PLSLAM_HIP_CHECK(hipEventRecord(ev->e2, s));
    if (P->scatter_counts && (r = launch_scatter_counts(P->d_counts_zero, P->d_count_dst, P->nprob, s))) return r;
    if (split) {
        PLSLAM_HIP_CHECK(hipEventRecord(P->post_done, s));
        P->post_pending = true;
    }
    return PLSLAM_OK;
}

namespace plslam {
int match_problems_on_ctx_stream(plslam_ctx* ctx, const plslam_match_problem* probs, int32_t nprob, const int32_t* n1_dev0)
{
    if (!ctx->host_plan) ctx->host_plan = new (std::nothrow) plslam_match_plan();
    PLSLAM_REQUIRE(ctx->host_plan != nullptr, PLSLAM_ENOMEM);
    const int r = plan_build(ctx, probs, nprob, ctx->host_plan, n1_dev0);
    if (r == PLSLAM_ENOTSUP && n1_dev0) return r;            // (not an error: the caller has another form; no message)
    return r ? r : plan_run(ctx->host_plan, ctx->stream, ctx->stream);
}
}  // namespace plslam

// ---------------------------------------------------------------------------------------------
// extern "C"
// ---------------------------------------------------------------------------------------------
extern "C" {

const char* plslam_strerror(int code)
{
    switch (code) {
        case PLSLAM_OK: return "ok";
        case PLSLAM_EINVAL: return "invalid argument";
        case PLSLAM_ENODEV: return "no usable gfx950 HIP device";
        case PLSLAM_EHIP: return "HIP runtime error";
        case PLSLAM_ENOMEM: return "out of memory";
        case PLSLAM_ERANGE: return "size beyond documented limit";
        case PLSLAM_ENOTSUP: return "optional component unavailable";
        default: return "unknown error";
    }
}

const char* plslam_last_error(void) { return g_err; }
int plslam_abi_version(void) { return PLSLAM_ABI_VERSION; }

int plslam_ctx_create(int device_ordinal, plslam_ctx** out)
{
    PLSLAM_REQUIRE(out != nullptr, PLSLAM_EINVAL);
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_last_error("no HIP device visible (this library has no CPU fallback)");
        return PLSLAM_ENODEV;
    }
    PLSLAM_REQUIRE(device_ordinal >= 0 && device_ordinal < ndev, PLSLAM_ENODEV);
    plslam_ctx* c = new (std::nothrow) plslam_ctx();
    PLSLAM_REQUIRE(c != nullptr, PLSLAM_ENOMEM);
    c->device = device_ordinal;
    DeviceGuard g(device_ordinal);
    if (hipGetDeviceProperties(&c->prop, device_ordinal) != hipSuccess) {
        delete c;
        set_last_error("hipGetDeviceProperties failed");
        return PLSLAM_ENODEV;
    }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        set_last_error("device %d is %s; this library carries gfx950 code only", device_ordinal,
                       c->prop.gcnArchName);
        delete c;
        return PLSLAM_ENODEV;
    }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        set_last_error("hipStreamCreate failed");
        return PLSLAM_EHIP;
    }
    *out = c;
    return PLSLAM_OK;
}

void plslam_ctx_destroy(plslam_ctx* ctx)
{
    if (!ctx) return;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->in_a.release(); ctx->in_b.release(); ctx->out_a.release(); ctx->out_b.release();
    ctx->misc_a.release(); ctx->misc_b.release(); ctx->misc_c.release();
    ctx->pin_in.release();
    ctx->pin_out.release();
    ctx->pin_misc.release();
    ctx->lbd_ring.release();
    if (ctx->host_plan) {
        ctx->host_plan->free_all();
        delete ctx->host_plan;
    }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int plslam_ctx_set_option(plslam_ctx* ctx, const char* key, int value)
{
    PLSLAM_REQUIRE(ctx && key, PLSLAM_EINVAL);
    if (!strcmp(key, "scan_variant")) {
        PLSLAM_REQUIRE(value >= PLSLAM_SCAN_AUTO && value <= PLSLAM_SCAN_MFMA, PLSLAM_EINVAL);
        ctx->scan_variant = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "scan_block")) {
        PLSLAM_REQUIRE(value == 0 || value == 256 || value == 512 || value == 1024, PLSLAM_EINVAL);
        ctx->scan_block = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "sym_rows")) {
        PLSLAM_REQUIRE(value == 0 || value == 1 || value == 4, PLSLAM_EINVAL);
        ctx->sym_rows = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "group_cap")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 64, PLSLAM_EINVAL);
        ctx->group_cap = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "mfma_form")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 5, PLSLAM_EINVAL);
        if (!mfma_form_built(value)) {
            set_last_error("mfma_form 1 / 3 / 4 (K1e, K1g, K1h: earlier generations of the matrix-core scan) are not in this build: "
                           "PLSLAM_BUILD_LEGACY_SCANS=1 python -m plslam_amd.build");
            return PLSLAM_ENOTSUP;
        }
        ctx->mfma_form = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "fuse")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 2, PLSLAM_EINVAL);
        ctx->fuse = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "post_fuse")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 2, PLSLAM_EINVAL);
        ctx->post_fuse = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "exact_second")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 1, PLSLAM_EINVAL);
        ctx->exact_second = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "col_split")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 2, PLSLAM_EINVAL);
        ctx->col_split = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "graph")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 2, PLSLAM_EINVAL);
        ctx->graph = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "grid_dense")) {               // (process-wide: the lone small matchGrid problem on one dense workgroup)
        PLSLAM_REQUIRE(value >= 0 && value <= 1, PLSLAM_EINVAL);
        plslam::g_grid_dense = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "zero_copy_kb")) {
        PLSLAM_REQUIRE(value >= -(1 << 20) && value <= (1 << 20), PLSLAM_EINVAL);
        ctx->zero_copy_kb = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "scan_tail")) {                // (experiment builds only: see hamming_mfma_i.hip, PLSLAM_MI_TAIL)
        PLSLAM_REQUIRE(value == 0 || value == 1, PLSLAM_EINVAL);
        if (value && !plslam::k1i_tail_built()) return PLSLAM_ENOTSUP;
        ctx->scan_tail = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "post_xcd")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 2, PLSLAM_EINVAL);
        ctx->post_xcd = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "split_post")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 1, PLSLAM_EINVAL);
        ctx->split_post = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "split_target")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 64, PLSLAM_EINVAL);
        ctx->split_target = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "split_min_tiles")) {
        PLSLAM_REQUIRE(value >= 0 && value <= 64, PLSLAM_EINVAL);
        ctx->split_min_tiles = value;
        return PLSLAM_OK;
    }
    if (!strcmp(key, "post_workgroups")) {
        PLSLAM_REQUIRE(value >= 0, PLSLAM_EINVAL);
        ctx->post_workgroups = value;
        return PLSLAM_OK;
    }
    set_last_error("unknown option '%s'", key);
    return PLSLAM_EINVAL;
}

int plslam_ctx_get_option(plslam_ctx* ctx, const char* key, int* value)
{
    PLSLAM_REQUIRE(ctx && key && value, PLSLAM_EINVAL);
    if (!strcmp(key, "scan_variant")) { *value = ctx->scan_variant; return PLSLAM_OK; }
    if (!strcmp(key, "scan_block")) { *value = ctx->scan_block; return PLSLAM_OK; }
    if (!strcmp(key, "sym_rows")) { *value = ctx->sym_rows; return PLSLAM_OK; }
    if (!strcmp(key, "group_cap")) { *value = ctx->group_cap; return PLSLAM_OK; }
    if (!strcmp(key, "mfma_form")) { *value = ctx->mfma_form; return PLSLAM_OK; }
    if (!strcmp(key, "legacy_scans")) { *value = PLSLAM_BUILD_LEGACY_SCANS; return PLSLAM_OK; }     // (read-only: a fact of the build)
    if (!strcmp(key, "post_fuse")) { *value = ctx->post_fuse; return PLSLAM_OK; }
    if (!strcmp(key, "fuse")) { *value = ctx->fuse; return PLSLAM_OK; }
    if (!strcmp(key, "col_split")) { *value = ctx->col_split; return PLSLAM_OK; }
    if (!strcmp(key, "exact_second")) { *value = ctx->exact_second; return PLSLAM_OK; }
    if (!strcmp(key, "post_workgroups")) { *value = ctx->post_workgroups; return PLSLAM_OK; }
    if (!strcmp(key, "post_xcd")) { *value = ctx->post_xcd; return PLSLAM_OK; }
    if (!strcmp(key, "scan_tail")) { *value = ctx->scan_tail; return PLSLAM_OK; }
    if (!strcmp(key, "zero_copy_kb")) { *value = ctx->zero_copy_kb; return PLSLAM_OK; }
    if (!strcmp(key, "grid_dense")) { *value = plslam::g_grid_dense; return PLSLAM_OK; }
    if (!strcmp(key, "split_post")) { *value = ctx->split_post; return PLSLAM_OK; }
    if (!strcmp(key, "split_target")) { *value = ctx->split_target; return PLSLAM_OK; }
    if (!strcmp(key, "split_min_tiles")) { *value = ctx->split_min_tiles; return PLSLAM_OK; }
    if (!strcmp(key, "graph")) { *value = ctx->graph; return PLSLAM_OK; }
    set_last_error("unknown option '%s'", key);
    return PLSLAM_EINVAL;
}

int plslam_ctx_device_info(plslam_ctx* ctx, int32_t* cu_count, int32_t* clock_khz,
                           int32_t* lds_bytes, char* name, int32_t name_len)
{
    PLSLAM_REQUIRE(ctx != nullptr, PLSLAM_EINVAL);
    if (cu_count) *cu_count = ctx->prop.multiProcessorCount;
    if (clock_khz) *clock_khz = ctx->prop.clockRate;
    if (lds_bytes) *lds_bytes = (int32_t)ctx->prop.sharedMemPerBlock;
    if (name && name_len > 0) {
        snprintf(name, (size_t)name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    }
    return PLSLAM_OK;
}

// ---- plans ----------------------------------------------------------------------------------
int plslam_match_plan_create(plslam_ctx* ctx, const plslam_match_problem* probs, int32_t nprob,
                             plslam_match_plan** out)
{
    PLSLAM_REQUIRE(ctx && out, PLSLAM_EINVAL);
    *out = nullptr;
    DeviceGuard g(ctx->device);
    plslam_match_plan* P = new (std::nothrow) plslam_match_plan();
    PLSLAM_REQUIRE(P != nullptr, PLSLAM_ENOMEM);
    int r = plan_build(ctx, probs, nprob, P);
    if (!r && hipStreamSynchronize(ctx->stream) != hipSuccess) {
        set_last_error("hipStreamSynchronize failed after the plan upload");
        r = PLSLAM_EHIP;
    }
    if (r) {
        P->free_all();
        delete P;
        return r;
    }
    *out = P;
    return PLSLAM_OK;
}

int plslam_match_plan_add_stereo_gates(plslam_match_plan* plan, const plslam_stereo_gate_problem* gates, int32_t ngates)
{
    PLSLAM_REQUIRE(plan != nullptr && ngates >= 0 && (ngates == 0 || gates != nullptr), PLSLAM_EINVAL);
    DeviceGuard g(plan->ctx->device);
    // ---- validation first: nothing of the plan is touched until the whole request is known to be good -----------------
    // a gate whose input table is the matches_12 of one of the plan's problems (the usual case: the L<->R tables of the
    // batch) is applied by the finalize kernel itself, row by row, the moment the entry is decided (ProblemDesc::gate);
    // any other gate -- and every gate of a fused plan, which has no finalize kernel -- keeps its own workgroups
    std::vector<BlockDesc> blocks;
    std::vector<int32_t> gate_of(plan->h_probs.size(), -1);
    bool any_cnt = false, all_cnt = true;
    for (int32_t i = 0; i < ngates; ++i) {
        const int rc = check_stereo_gate_problem(gates[i]);
        if (rc) return rc;
        any_cnt = any_cnt || gates[i].n_stereo != nullptr;
        all_cnt = all_cnt && gates[i].n_stereo != nullptr && gates[i].n_stereo == gates[0].n_stereo + i;
        bool in_finalize = false;
        if (!plan->fused && gates[i].n_l > 0)
            for (size_t k = 0; k < plan->h_probs.size(); ++k) {
                const ProblemDesc& pd = plan->h_probs[k];
                if (pd.matches_12 == gates[i].matches_12 && pd.n1 == gates[i].n_l && gate_of[k] < 0) {
                    gate_of[k] = i;
                    in_finalize = true;
                    break;
                }
            }
        if (!in_finalize)
            for (int32_t r0 = 0; r0 < gates[i].n_l; r0 += 256) blocks.push_back({i, r0});
    }
    PLSLAM_REQUIRE(!any_cnt || all_cnt, PLSLAM_EINVAL);     // counters: none, or one contiguous array
    // ---- the stage is replaced: its tables and the problem table (ProblemDesc::gate) are rewritten under whatever run is
    // still in flight on whatever stream the caller used -- the call is rare, so it simply waits for the device ------------
    PLSLAM_HIP_CHECK(hipDeviceSynchronize());
    if (plan->graph_exec) plan->drop_graph();               // the captured run has another gate stage
    // from here on every exit leaves the plan consistent: first the state "no gate stage" (host image AND device table) ...
    plan->ngate_blocks = 0;
    plan->ngates = 0;
    plan->d_gate_counts = nullptr;
    plan->d_gates = nullptr;
    plan->split_post = plan->split_post_ok;
    auto upload_probs = [&]() -> int {
        if (plan->h_probs.empty()) return PLSLAM_OK;
        PLSLAM_HIP_CHECK(hipMemcpyAsync(plan->d_probs, plan->h_probs.data(), plan->h_probs.size() * sizeof(ProblemDesc),
                                        plan->probs_in_place ? hipMemcpyHostToHost : hipMemcpyHostToDevice, plan->ctx->stream));
        PLSLAM_HIP_CHECK(hipStreamSynchronize(plan->ctx->stream));
        return PLSLAM_OK;
    };
    bool had = false;
    for (ProblemDesc& pd : plan->h_probs) { had = had || pd.gate >= 0; pd.gate = -1; }
    if (had) { const int rc = upload_probs(); if (rc) return rc; }
    if (ngates == 0) return PLSLAM_OK;
    // ... then the new stage: its tables first (a failed allocation returns with the consistent "no gate stage" above) ...
    const size_t gbytes = ((size_t)ngates * sizeof(plslam_stereo_gate_problem) + 255) & ~size_t(255);
    const size_t total = gbytes + blocks.size() * sizeof(BlockDesc) + 256;
    plan->gate_staging.assign(total, 0);
    memcpy(plan->gate_staging.data(), gates, (size_t)ngates * sizeof(plslam_stereo_gate_problem));
    if (!blocks.empty()) memcpy(plan->gate_staging.data() + gbytes, blocks.data(), blocks.size() * sizeof(BlockDesc));
    int r = plan->gate_tables.reserve(total);
    if (r) return r;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(plan->gate_tables.p, plan->gate_staging.data(), total, hipMemcpyHostToDevice,
                                    plan->ctx->stream));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(plan->ctx->stream));
    // ... and, last, the gate indices of the problem table together with the counters that make a run use them: a failure of
    // the upload puts the indices back (the device table then still holds -1 everywhere or is rewritten in full next time)
    for (size_t k = 0; k < plan->h_probs.size(); ++k) plan->h_probs[k].gate = gate_of[k];
    r = upload_probs();
    if (r) {
        for (ProblemDesc& pd : plan->h_probs) pd.gate = -1;
        (void)upload_probs();
        return r;
    }
    plan->d_gates = plan->gate_tables.as<plslam_stereo_gate_problem>();
    plan->d_gate_blocks = reinterpret_cast<BlockDesc*>(plan->gate_tables.as<char>() + gbytes);
    plan->ngate_blocks = (int32_t)blocks.size();
    plan->ngates = ngates;
    plan->d_gate_counts = any_cnt ? gates[0].n_stereo : nullptr;
    for (int32_t gk : gate_of)
        if (gk >= 0) plan->split_post = false;              // the finalize kernel applies that gate: merge + finalize it is
    return PLSLAM_OK;
}

int plslam_match_plan_set_wire16(plslam_match_plan* plan, const int32_t* table32, int16_t* table16, size_t n_entries)
{
    PLSLAM_REQUIRE(plan != nullptr && (table16 == nullptr || table32 != nullptr), PLSLAM_EINVAL);
    DeviceGuard g(plan->ctx->device);
    // only the finalize kernel (k_finalize / k_post_fused: finalize_row) stores the mirror: a fused plan decides its entries in
    // the scan kernel, a column-split plan in k_split_post
    if (table16 && (plan->fused || plan->col_split)) {
        set_last_error("plslam_match_plan_set_wire16: this plan's tables are not written by the finalize kernel");
        return PLSLAM_ENOTSUP;
    }
    std::vector<int16_t*> mirror(plan->h_probs.size(), nullptr);
    if (table16)
        for (size_t k = 0; k < plan->h_probs.size(); ++k) {
            const ProblemDesc& pd = plan->h_probs[k];
            if (pd.n1 <= 0 || pd.matches_12 < table32 || pd.matches_12 >= table32 + n_entries) continue;
            PLSLAM_REQUIRE((size_t)(pd.matches_12 - table32) + (size_t)pd.n1 <= n_entries, PLSLAM_EINVAL);
            PLSLAM_REQUIRE(pd.n2 <= 32768, PLSLAM_EINVAL);
            // a kept prior entry of a NON-mutual problem reaches the table unchecked (upstream's resize(); a mutual problem
            // clears whatever lies outside [0, n2)): it could be any int32 and would wrap in the mirror
            if (pd.keep_prior && !pd.mutual) {
                set_last_error("plslam_match_plan_set_wire16: problem %zu keeps prior entries without the mutual check (unbounded values)", k);
                return PLSLAM_ENOTSUP;
            }
            mirror[k] = table16 + (pd.matches_12 - table32);
        }
    PLSLAM_HIP_CHECK(hipDeviceSynchronize());               // (as add_stereo_gates: rare, so it waits for whatever is in flight)
    if (plan->graph_exec) plan->drop_graph();
    for (size_t k = 0; k < plan->h_probs.size(); ++k) plan->h_probs[k].matches_16 = mirror[k];
    if (!plan->h_probs.empty()) {
        PLSLAM_HIP_CHECK(hipMemcpyAsync(plan->d_probs, plan->h_probs.data(), plan->h_probs.size() * sizeof(ProblemDesc),
                                        plan->probs_in_place ? hipMemcpyHostToHost : hipMemcpyHostToDevice, plan->ctx->stream));
        PLSLAM_HIP_CHECK(hipStreamSynchronize(plan->ctx->stream));
    }
    return PLSLAM_OK;
}

int plslam_match_plan_run(plslam_match_plan* plan, void* stream)
{
    PLSLAM_REQUIRE(plan != nullptr, PLSLAM_EINVAL);
    DeviceGuard g(plan->ctx->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : plan->ctx->stream;
    // Graph replay (option "graph": 0 = plans of fewer waves than the chip has SIMDs, 1 = never, 2 = always): the run's
    // launches are captured once on the caller's stream and replayed with one hipGraphLaunch.  Not while profiling (the
    // events belong to the run), not for a plan with a split run pending (its ordering event is not part of the graph).
    const int gopt = plan->ctx->graph;
    if (gopt != 1 && (gopt == 2 || plan->small) && !plan->profiling && !plan->post_pending && !plan->graph_failed) {
        if (!plan->graph_exec) {
            hipGraph_t gr = nullptr;
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int r = plan_run(plan, s, s);
                const hipError_t e = hipStreamEndCapture(s, &gr);
                if (r == PLSLAM_OK && e == hipSuccess && gr && hipGraphInstantiate(&plan->graph_exec, gr, nullptr, nullptr, 0) != hipSuccess)
                    plan->graph_exec = nullptr;
                if (gr) (void)hipGraphDestroy(gr);
            }
            (void)hipGetLastError();
            if (!plan->graph_exec) plan->graph_failed = true;       // (this device / runtime cannot: plain launches from now on)
        }
        if (plan->graph_exec) {
            PLSLAM_HIP_CHECK(hipGraphLaunch(plan->graph_exec, s));
            return PLSLAM_OK;
        }
    }
    return plan_run(plan, s, s);
}

int plslam_match_plan_run_split(plslam_match_plan* plan, void* scan_stream, void* post_stream)
{
    PLSLAM_REQUIRE(plan != nullptr, PLSLAM_EINVAL);
    DeviceGuard g(plan->ctx->device);
    hipStream_t s = scan_stream ? static_cast<hipStream_t>(scan_stream) : plan->ctx->stream;
    // A fused plan (option "fuse" = 2) writes matches_12 and the counts from inside its scan kernel: there is no stage behind
    // the scan whose completion could order the caller's consumers against the plan's NEXT scan, so such a plan is not
    // split -- everything goes to the scan stream, where stream order does it.
    if (plan->fused) return plan_run(plan, s, s);
    return plan_run(plan, s, post_stream ? static_cast<hipStream_t>(post_stream) : s);
}

int plslam_match_plan_set_profiling(plslam_match_plan* plan, int enable)
{
    PLSLAM_REQUIRE(plan != nullptr, PLSLAM_EINVAL);
    plan->profiling = enable != 0;
    return PLSLAM_OK;
}

int plslam_match_plan_elapsed(plslam_match_plan* plan, double* scan_ms, double* finalize_ms,
                              int64_t* runs)
{
    PLSLAM_REQUIRE(plan != nullptr, PLSLAM_EINVAL);
    DeviceGuard g(plan->ctx->device);
    for (size_t i = 0; i < plan->ev_used; ++i) {
        auto& e = plan->evs[i];
        PLSLAM_HIP_CHECK(hipEventSynchronize(e.e2));
        float a = 0.f, b = 0.f;
        PLSLAM_HIP_CHECK(hipEventElapsedTime(&a, e.e0, e.e1));
        PLSLAM_HIP_CHECK(hipEventElapsedTime(&b, e.e1, e.e2));
        plan->acc_scan_ms += a;
        plan->acc_fin_ms += b;
        plan->acc_runs += 1;
    }
    plan->ev_used = 0;
    if (scan_ms) *scan_ms = plan->acc_scan_ms;
    if (finalize_ms) *finalize_ms = plan->acc_fin_ms;
    if (runs) *runs = plan->acc_runs;
    plan->acc_scan_ms = plan->acc_fin_ms = 0;
    plan->acc_runs = 0;
    return PLSLAM_OK;
}

int plslam_match_plan_info(plslam_match_plan* plan, plslam_plan_info* info)
{
    PLSLAM_REQUIRE(plan && info, PLSLAM_EINVAL);
    *info = plan->info;
    return PLSLAM_OK;
}

// diagnostics: raw copies of a plan's key table and the symmetric scan's column partials (see the header)
int plslam_match_plan_dump(plslam_match_plan* plan, void* keys_out, size_t keys_cap, void* part_out, size_t part_cap,
                           size_t* keys_bytes, size_t* part_bytes)
{
    if (!plan || !keys_bytes || !part_bytes) return PLSLAM_EINVAL;
    DeviceGuard g(plan->ctx->device);
    (void)hipDeviceSynchronize();
    if (plan->split_post) {            // the two-launch form leaves the rows' merged pairs unwritten: complete them here
        (void)launch_split_rows_dump(plan->d_probs, plan->d_fin_blocks, plan->nfin_blocks, plan->ctx->stream);
        (void)hipDeviceSynchronize();
    }
    *keys_bytes = plan->keys.cap;
    *part_bytes = plan->partials.cap;
    if (keys_out && keys_cap >= plan->keys.cap && plan->keys.p) (void)hipMemcpy(keys_out, plan->keys.p, plan->keys.cap, hipMemcpyDeviceToHost);
    if (part_out && part_cap >= plan->partials.cap && plan->partials.p) (void)hipMemcpy(part_out, plan->partials.p, plan->partials.cap, hipMemcpyDeviceToHost);
    return PLSLAM_OK;
}

void plslam_match_plan_destroy(plslam_match_plan* plan)
{
    if (!plan) return;
    DeviceGuard g(plan->ctx->device);
    (void)hipDeviceSynchronize();
    plan->free_all();
    delete plan;
}

// ---- host-pointer matching -------------------------------------------------------------------
int plslam_match_plan_key_state(plslam_match_plan* plan, int32_t* flags)
{
    if (!plan || !flags) return PLSLAM_EINVAL;
    const bool h = plan->sym_mfma && mfma_form_is_h(plan->mfma_form) && !plan->fused;
    *flags = (h && !plan->exact_second ? PLSLAM_KEYS_ROW_SECOND_INDEX_INEXACT | PLSLAM_KEYS_COLUMN_SECOND_LAZY : 0) |
             (plan->post_fused ? PLSLAM_KEYS_COLUMNS_NOT_IN_MEMORY : 0);
    return PLSLAM_OK;
}

int plslam_match_batched(plslam_ctx* ctx, const uint8_t* d1, const int32_t* off1,
                         const uint8_t* d2, const int32_t* off2, int32_t B, float nnr, int mutual,
                         int32_t* matches_12, int32_t* n_matches)
{
    PLSLAM_REQUIRE(ctx != nullptr && B >= 0, PLSLAM_EINVAL);
    if (B == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(off1 && off2, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(off1[0] == 0 && off2[0] == 0, PLSLAM_EINVAL);
    for (int32_t b = 0; b < B; ++b)
        PLSLAM_REQUIRE(off1[b + 1] >= off1[b] && off2[b + 1] >= off2[b], PLSLAM_EINVAL);
    const int64_t r1 = off1[B], r2 = off2[B];
    PLSLAM_REQUIRE(r1 == 0 || (d1 && matches_12), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(r2 == 0 || d2, PLSLAM_EINVAL);

    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    int r;
    if ((r = ctx->in_a.reserve((size_t)r1 * 32 + 16))) return r;
    if ((r = ctx->in_b.reserve((size_t)r2 * 32 + 16))) return r;
    if ((r = ctx->out_a.reserve((size_t)r1 * 4 + 16))) return r;
    if ((r = ctx->out_b.reserve((size_t)B * 4))) return r;
    // Latency path (one StVO::match of the SLAM loop, a frame's handful of problems): stage through the
    // context's pinned buffers -- the CPU copies ~100 kB in a few microseconds and every hipMemcpyAsync
    // becomes a plain DMA enqueue instead of the runtime's pageable-memory path.
    const size_t in1 = (size_t)r1 * 32, in2 = (size_t)r2 * 32, out1 = (size_t)r1 * 4, out2 = (size_t)B * 4;
    const bool pinned = in1 + in2 <= (size_t(1) << 20);
    const size_t in2_off = (in1 + 255) & ~size_t(255), out2_off = (out1 + 255) & ~size_t(255);
    const uint8_t *dev1 = ctx->in_a.as<uint8_t>(), *dev2 = ctx->in_b.as<uint8_t>();
    int32_t* tab_dev = ctx->out_a.as<int32_t>();      // where the kernels write the table
    bool table_in_place = false;
    if (pinned) {
        // ONE page-locked image [d1 | d2] -> ONE upload; the table is written by the finalize kernel straight into
        // page-locked memory (the counts are then the number of entries >= 0: no download at all)
        if ((r = ctx->pin_in.reserve(in2_off + in2 + 256))) return r;
        if ((r = ctx->pin_out.reserve(out2_off + out2 + 256))) return r;
        if ((r = ctx->in_a.reserve(in2_off + in2 + 256))) return r;
        if (in1) memcpy(ctx->pin_in.as<char>(), d1, in1);
        if (in2) memcpy(ctx->pin_in.as<char>() + in2_off, d2, in2);
        dev1 = ctx->in_a.as<uint8_t>();
        dev2 = ctx->in_a.as<uint8_t>() + in2_off;
        if (r1 + r2) PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->in_a.p, ctx->pin_in.p, in2_off + in2, hipMemcpyHostToDevice, ctx->stream));
        if (void* m = mapped_device_pointer(ctx->pin_out.p)) {
            tab_dev = static_cast<int32_t*>(m);
            table_in_place = true;
        }
    } else {
        if (r1) PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->in_a.p, d1, in1, hipMemcpyHostToDevice, ctx->stream));
        if (r2) PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->in_b.p, d2, in2, hipMemcpyHostToDevice, ctx->stream));
    }

    std::vector<plslam_match_problem> probs((size_t)B);
    for (int32_t b = 0; b < B; ++b) {
        plslam_match_problem& p = probs[b];
        p.d1 = dev1 + (size_t)off1[b] * 32;
        p.d2 = dev2 + (size_t)off2[b] * 32;
        p.n1 = off1[b + 1] - off1[b];
        p.n2 = off2[b + 1] - off2[b];
        p.nnr = nnr;
        p.mutual = mutual ? 1 : 0;
        p.matches_12 = tab_dev + off1[b];
        p.n_matches = ctx->out_b.as<int32_t>() + b;
    }
    // the context keeps ONE plan object for the host-pointer path: its device buffers only grow, so
    // a call in the SLAM loop does no hipMalloc/hipFree
    if (!ctx->host_plan) ctx->host_plan = new (std::nothrow) plslam_match_plan();
    PLSLAM_REQUIRE(ctx->host_plan != nullptr, PLSLAM_ENOMEM);
    plslam_match_plan& P = *ctx->host_plan;
    P.pin_tables = true;
    r = plan_build(ctx, probs.data(), B, &P);
    if (!r) r = plan_run(&P, ctx->stream, ctx->stream);
    if (!r) {
        hipError_t e = hipSuccess;
        void* dst1 = pinned ? ctx->pin_out.p : (void*)matches_12;
        void* dst2 = pinned ? (void*)(ctx->pin_out.as<char>() + out2_off) : (void*)n_matches;
        if (r1 && !table_in_place) e = hipMemcpyAsync(dst1, ctx->out_a.p, out1, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && n_matches && !table_in_place)
            e = hipMemcpyAsync(dst2, ctx->out_b.p, out2, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess && pinned) {
            if (r1) memcpy(matches_12, dst1, out1);
            if (n_matches && !table_in_place) memcpy(n_matches, dst2, out2);
            if (n_matches && table_in_place)          // StVO::match on a fresh vector: the count IS the number of entries
                for (int32_t b = 0; b < B; ++b) {
                    int32_t n = 0;
                    for (int32_t i = off1[b]; i < off1[b + 1]; ++i) n += matches_12[i] >= 0;
                    n_matches[b] = n;
                }
        }
        if (e != hipSuccess) {
            set_last_error("%s:%d: D2H of match tables -> %s", __FILE__, __LINE__, hipGetErrorString(e));
            r = PLSLAM_EHIP;
        }
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    return r;
}

int plslam_match(plslam_ctx* ctx, const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2,
                 float nnr, int mutual, int32_t* matches_12, int32_t* n_matches)
{
    PLSLAM_REQUIRE(n1 >= 0 && n2 >= 0, PLSLAM_EINVAL);
    const int32_t off1[2] = {0, n1}, off2[2] = {0, n2};
    int32_t n = 0;
    const int r = plslam_match_batched(ctx, d1, off1, d2, off2, 1, nnr, mutual, matches_12, &n);
    if (!r && n_matches) *n_matches = n;
    return r;
}

int plslam_match_prior(plslam_ctx* ctx, const uint8_t* d1, int32_t n1, const uint8_t* d2, int32_t n2,
                       float nnr, int mutual, int32_t* matches_12, int32_t* n_matches)
{
    PLSLAM_REQUIRE(ctx != nullptr && n1 >= 0 && n2 >= 0, PLSLAM_EINVAL);
    if (n_matches) *n_matches = 0;
    if (n1 == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(d1 && matches_12 && (n2 == 0 || d2), PLSLAM_EINVAL);
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    hipStream_t s = ctx->stream;
    int r;
    if ((r = ctx->in_a.reserve((size_t)n1 * 32 + 16))) return r;
    if ((r = ctx->in_b.reserve((size_t)n2 * 32 + 16))) return r;
    if ((r = ctx->out_a.reserve((size_t)n1 * 4 + 16))) return r;
    if ((r = ctx->out_b.reserve(16))) return r;
    StreamSyncOnError guard(s);
    PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->in_a.p, d1, (size_t)n1 * 32, hipMemcpyHostToDevice, s));
    if (n2) PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->in_b.p, d2, (size_t)n2 * 32, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->out_a.p, matches_12, (size_t)n1 * 4, hipMemcpyHostToDevice, s));
    plslam_match_problem p{};
    p.d1 = ctx->in_a.as<uint8_t>(); p.d2 = ctx->in_b.as<uint8_t>(); p.n1 = n1; p.n2 = n2; p.nnr = nnr;
    p.mutual = mutual ? 1 : 0; p.matches_12 = ctx->out_a.as<int32_t>(); p.n_matches = ctx->out_b.as<int32_t>();
    p.keep_prior = 1;
    if ((r = plslam::match_problems_on_ctx_stream(ctx, &p, 1))) return r;
    int32_t n = 0;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(matches_12, ctx->out_a.p, (size_t)n1 * 4, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(&n, ctx->out_b.p, 4, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    guard.dismiss();
    if (n_matches) *n_matches = n;
    return PLSLAM_OK;
}

int plslam_knn2_hamming256(plslam_ctx* ctx, const uint8_t* q, int32_t nq, const uint8_t* t,
                           int32_t nt, int32_t* idx, int32_t* dist)
{
    PLSLAM_REQUIRE(ctx != nullptr && nq >= 0 && nt >= 0, PLSLAM_EINVAL);
    if (nq == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(q && idx && dist && (nt == 0 || t), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(nt <= PLSLAM_MAX_TRAIN_ROWS, PLSLAM_ERANGE);
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    int r;
    hipStream_t s = ctx->stream;
    StreamSyncOnError sg(s);
    const bool small = (nq + 63) / 64 < ctx->prop.multiProcessorCount * 4;
    // large query sets (or a forced variant): the directed form of the matrix-core scan; otherwise the popcount kernels
    const bool mfma = nt > 0 && (ctx->scan_variant == PLSLAM_SCAN_MFMA || (ctx->scan_variant == PLSLAM_SCAN_AUTO && !small));
    const int variant = ctx->scan_variant == PLSLAM_SCAN_WAVE_PER_QUERY || (ctx->scan_variant == PLSLAM_SCAN_AUTO && small)
                            ? PLSLAM_SCAN_WAVE_PER_QUERY : PLSLAM_SCAN_LANE_PER_QUERY;
    const int bt = variant == PLSLAM_SCAN_WAVE_PER_QUERY ? 256 : (ctx->scan_block ? ctx->scan_block : 256);
    const int rpb = mfma ? 256 : scan_rows_per_block(variant, bt);
    std::vector<BlockDesc> blocks;
    for (int32_t r0 = 0; r0 < nq; r0 += rpb) blocks.push_back({0, r0});
    if (mfma || variant == PLSLAM_SCAN_LANE_PER_QUERY) {   // these kernels read the XCD-striped layout (8 rows of L)
        const size_t n = blocks.size(), L = (n + 7) / 8;
        std::vector<BlockDesc> striped(8 * L, BlockDesc{-1, 0});
        for (size_t i = 0; i < n; ++i) striped[(i & 7) * L + (i >> 3)] = blocks[i];
        blocks.swap(striped);
    }
    // one image [q | t | scan descriptor | block table] -- page-locked and uploaded with ONE copy when it is small --, the
    // (idx, dist) pairs written by the unpack kernel straight into page-locked memory when the device can address it
    const size_t qb = (size_t)nq * 32, tb = (size_t)nt * 32, bb = blocks.size() * sizeof(BlockDesc);
    Carver c;
    const size_t oQ = c.take(qb), oT = c.take(tb + 16), oD = c.take(std::max(sizeof(SymDesc), sizeof(ScanDesc))), oB = c.take(bb);
    const size_t ob = ((size_t)nq * 8 + 255) & ~size_t(255);
    if ((r = ctx->in_a.reserve(c.off))) return r;
    if ((r = ctx->misc_a.reserve((size_t)nq * 8))) return r;                 // keys
    if ((r = ctx->out_a.reserve(2 * ob))) return r;                          // idx | dist
    if ((r = ctx->pin_out.reserve(2 * ob))) return r;
    char* d = ctx->in_a.as<char>();
    SymDesc y{};
    y.a = (const uint8_t*)(d + oQ); y.b = (const uint8_t*)(d + oT); y.keys12 = ctx->misc_a.as<uint32_t>();
    y.n1 = nq; y.n2 = nt;
    y.flags = 1;                                   // knnMatch returns the second neighbour's index
    const ScanDesc sd{(const uint8_t*)(d + oQ), (const uint8_t*)(d + oT), ctx->misc_a.as<uint32_t>(), nq, nt};
    const void* desc = mfma ? (const void*)&y : (const void*)&sd;
    const size_t desc_bytes = mfma ? sizeof(y) : sizeof(sd);
    if (c.off <= (size_t(1) << 20)) {
        if ((r = ctx->pin_in.reserve(c.off))) return r;
        char* h = ctx->pin_in.as<char>();
        memcpy(h + oQ, q, qb);
        if (tb) memcpy(h + oT, t, tb);
        memcpy(h + oD, desc, desc_bytes);
        memcpy(h + oB, blocks.data(), bb);
        PLSLAM_HIP_CHECK(hipMemcpyAsync(d, h, c.off, hipMemcpyHostToDevice, s));
    } else {
        PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oQ, q, qb, hipMemcpyHostToDevice, s));
        if (tb) PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oT, t, tb, hipMemcpyHostToDevice, s));
        PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oD, desc, desc_bytes, hipMemcpyHostToDevice, s));
        PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oB, blocks.data(), bb, hipMemcpyHostToDevice, s));
    }
    if (mfma)
        r = launch_scan_mfma_form(ctx->mfma_form, (const SymDesc*)(d + oD), (const BlockDesc*)(d + oB), (int)blocks.size(), nullptr,
                                  0, nt > 2048, true, s);
    else
        r = launch_scan(ctx, variant, bt, (const ScanDesc*)(d + oD), (const BlockDesc*)(d + oB), (int)blocks.size(), nullptr, 0, s);
    if (r) return r;
    char* ho = ctx->pin_out.as<char>();
    char* out_dev = static_cast<char*>(mapped_device_pointer(ho));
    const bool in_place = out_dev != nullptr;
    if (!in_place) out_dev = ctx->out_a.as<char>();
    if ((r = launch_unpack_keys(ctx->misc_a.as<uint32_t>(), nq * 2, (int32_t*)out_dev, (int32_t*)(out_dev + ob), s))) return r;
    if (!in_place) PLSLAM_HIP_CHECK(hipMemcpyAsync(ho, out_dev, 2 * ob, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    sg.dismiss();
    memcpy(idx, ho, (size_t)nq * 8);
    memcpy(dist, ho + ob, (size_t)nq * 8);
    return PLSLAM_OK;
}

// ---- LBA rows --------------------------------------------------------------------------------
int plslam_lba_point_rows_dev_n(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                                const double* T_kf_w, int32_t n_pose_slots, const double* Xw, const double* obs_uv,
                                const int32_t* lm_loc, const int32_t* kf_slot, int32_t nobs,
                                double* J_pose, double* J_lm, double* r, double* w, void* stream)
{
    PLSLAM_REQUIRE(ctx && K && nobs >= 0 && n_pose_slots >= 0, PLSLAM_EINVAL);
    if (nobs == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(T_kf_w && Xw && obs_uv && lm_loc && kf_slot && J_pose && J_lm && r && w, PLSLAM_EINVAL);
    DeviceGuard g(ctx->device);
    return launch_point_rows(*K, homog_th, T_kf_w, Xw, obs_uv, lm_loc, kf_slot, nobs, J_pose, J_lm, r, w,
                             stream ? static_cast<hipStream_t>(stream) : ctx->stream, n_pose_slots);
}

int plslam_lba_point_rows_dev(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                              const double* T_kf_w, const double* Xw, const double* obs_uv,
                              const int32_t* lm_loc, const int32_t* kf_slot, int32_t nobs,
                              double* J_pose, double* J_lm, double* r, double* w, void* stream)
{
    return plslam_lba_point_rows_dev_n(ctx, K, homog_th, T_kf_w, 0, Xw, obs_uv, lm_loc, kf_slot, nobs, J_pose, J_lm, r, w, stream);
}

int plslam_lba_line_rows_dev_n(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                               int compat_iter_pass, const double* T_kf_w, int32_t n_pose_slots, const double* Lw,
                               const double* l_obs, const int32_t* lm_loc, const int32_t* kf_slot,
                               int32_t nobs, double* J_pose, double* J_lm, double* r, double* w,
                               void* stream)
{
    PLSLAM_REQUIRE(ctx && K && nobs >= 0 && n_pose_slots >= 0, PLSLAM_EINVAL);
    if (nobs == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(T_kf_w && Lw && l_obs && lm_loc && kf_slot && J_pose && J_lm && r && w, PLSLAM_EINVAL);
    DeviceGuard g(ctx->device);
    return launch_line_rows(*K, homog_th, compat_iter_pass ? 1 : 0, T_kf_w, Lw, l_obs, lm_loc, kf_slot,
                            nobs, J_pose, J_lm, r, w,
                            stream ? static_cast<hipStream_t>(stream) : ctx->stream, n_pose_slots);
}

int plslam_lba_line_rows_dev(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                             int compat_iter_pass, const double* T_kf_w, const double* Lw,
                             const double* l_obs, const int32_t* lm_loc, const int32_t* kf_slot,
                             int32_t nobs, double* J_pose, double* J_lm, double* r, double* w,
                             void* stream)
{
    return plslam_lba_line_rows_dev_n(ctx, K, homog_th, compat_iter_pass, T_kf_w, 0, Lw, l_obs, lm_loc, kf_slot, nobs, J_pose, J_lm, r, w,
                                      stream);
}


static int lba_rows_host(plslam_ctx* ctx, const plslam_cam* K, double th, int lines, int compat,
                         const double* T, int32_t nkf, const double* LM, int64_t n_lm_doubles,
                         const double* obs, int obs_stride, const int32_t* lm_loc,
                         const int32_t* kf_slot, int32_t nobs, double* Jp, double* Jl, double* r,
                         double* w)
{
    PLSLAM_REQUIRE(ctx && K && nobs >= 0 && nkf >= 0 && n_lm_doubles >= 0, PLSLAM_EINVAL);
    if (nobs == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(T && LM && obs && lm_loc && kf_slot && Jp && Jl && r && w, PLSLAM_EINVAL);
    const int lmw = lines ? 6 : 3;
    const int lm_stride = lines ? (compat ? 3 : 6) : 3;
    for (int32_t o = 0; o < nobs; ++o) {  // index validation: the kernel trusts its inputs
        PLSLAM_REQUIRE(kf_slot[o] >= 0 && kf_slot[o] < nkf, PLSLAM_EINVAL);
        PLSLAM_REQUIRE(lm_loc[o] >= 0 && (int64_t)lm_loc[o] * lm_stride + 3 <= n_lm_doubles, PLSLAM_EINVAL);
        PLSLAM_REQUIRE(!lines || compat || (int64_t)lm_loc[o] * 6 + 6 <= n_lm_doubles, PLSLAM_EINVAL);
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    Carver ci, co;
    const size_t oT = ci.take((size_t)nkf * 128), oL = ci.take((size_t)n_lm_doubles * 8),
                 oO = ci.take((size_t)nobs * obs_stride * 8), oLm = ci.take((size_t)nobs * 4),
                 oKf = ci.take((size_t)nobs * 4);
    const size_t oJp = co.take((size_t)nobs * 48), oJl = co.take((size_t)nobs * lmw * 8),
                 oR = co.take((size_t)nobs * 8), oW = co.take((size_t)nobs * 8);
    int rc;
    if ((rc = ctx->in_a.reserve(ci.off))) return rc;
    if ((rc = ctx->out_a.reserve(co.off))) return rc;
    char* di = ctx->in_a.as<char>();
    char* dout = ctx->out_a.as<char>();
    hipStream_t s = ctx->stream;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(di + oT, T, (size_t)nkf * 128, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(di + oL, LM, (size_t)n_lm_doubles * 8, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(di + oO, obs, (size_t)nobs * obs_stride * 8, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(di + oLm, lm_loc, (size_t)nobs * 4, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(di + oKf, kf_slot, (size_t)nobs * 4, hipMemcpyHostToDevice, s));
    if (!lines)
        rc = launch_point_rows(*K, th, (double*)(di + oT), (double*)(di + oL), (double*)(di + oO),
                               (int32_t*)(di + oLm), (int32_t*)(di + oKf), nobs, (double*)(dout + oJp),
                               (double*)(dout + oJl), (double*)(dout + oR), (double*)(dout + oW), s, nkf);
    else
        rc = launch_line_rows(*K, th, compat, (double*)(di + oT), (double*)(di + oL), (double*)(di + oO),
                              (int32_t*)(di + oLm), (int32_t*)(di + oKf), nobs, (double*)(dout + oJp),
                              (double*)(dout + oJl), (double*)(dout + oR), (double*)(dout + oW), s, nkf);
    if (rc) return rc;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(Jp, dout + oJp, (size_t)nobs * 48, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(Jl, dout + oJl, (size_t)nobs * lmw * 8, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(r, dout + oR, (size_t)nobs * 8, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(w, dout + oW, (size_t)nobs * 8, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return PLSLAM_OK;
}

int plslam_lba_point_rows(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                          const double* T_kf_w, int32_t nkf, const double* Xw, int32_t npt,
                          const double* obs_uv, const int32_t* lm_loc, const int32_t* kf_slot,
                          int32_t nobs, double* J_pose, double* J_lm, double* r, double* w)
{
    return lba_rows_host(ctx, K, homog_th, 0, 0, T_kf_w, nkf, Xw, (int64_t)npt * 3, obs_uv, 2, lm_loc,
                         kf_slot, nobs, J_pose, J_lm, r, w);
}

int plslam_lba_line_rows(plslam_ctx* ctx, const plslam_cam* K, double homog_th,
                         int compat_iter_pass, const double* T_kf_w, int32_t nkf, const double* Lw,
                         int32_t n_lw, const double* l_obs, const int32_t* lm_loc,
                         const int32_t* kf_slot, int32_t nobs, double* J_pose, double* J_lm,
                         double* r, double* w)
{
    return lba_rows_host(ctx, K, homog_th, 1, compat_iter_pass ? 1 : 0, T_kf_w, nkf, Lw, n_lw, l_obs, 3,
                         lm_loc, kf_slot, nobs, J_pose, J_lm, r, w);
}

// ---- gates -----------------------------------------------------------------------------------
static int gate_host(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, int lines,
                     const double* LM, const int32_t* m12, int32_t nq, const double* feat, int32_t nt,
                     double th, uint8_t* mask, int32_t* n_inliers)
{
    PLSLAM_REQUIRE(ctx && K && Twf && nq >= 0 && nt >= 0, PLSLAM_EINVAL);
    if (n_inliers) *n_inliers = 0;
    if (nq == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(LM && m12 && mask && (nt == 0 || feat), PLSLAM_EINVAL);
    for (int32_t i = 0; i < nq; ++i) PLSLAM_REQUIRE(m12[i] < nt, PLSLAM_EINVAL);
    const int lw = lines ? 6 : 3, fw = lines ? 3 : 2;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    Carver c;
    const size_t oL = c.take((size_t)nq * lw * 8), oM = c.take((size_t)nq * 4),
                 oF = c.take((size_t)nt * fw * 8 + 16), oMask = c.take((size_t)nq), oCnt = c.take(4);
    int rc;
    if ((rc = ctx->in_a.reserve(c.off))) return rc;
    char* d = ctx->in_a.as<char>();
    hipStream_t s = ctx->stream;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oL, LM, (size_t)nq * lw * 8, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oM, m12, (size_t)nq * 4, hipMemcpyHostToDevice, s));
    if (nt) PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oF, feat, (size_t)nt * fw * 8, hipMemcpyHostToDevice, s));
    rc = lines ? launch_line_gate(*K, Twf, (double*)(d + oL), (int32_t*)(d + oM), nq, (double*)(d + oF), th,
                                  (uint8_t*)(d + oMask), (int32_t*)(d + oCnt), s)
               : launch_point_gate(*K, Twf, (double*)(d + oL), (int32_t*)(d + oM), nq, (double*)(d + oF), th,
                                   (uint8_t*)(d + oMask), (int32_t*)(d + oCnt), s);
    if (rc) return rc;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(mask, d + oMask, (size_t)nq, hipMemcpyDeviceToHost, s));
    int32_t cnt = 0;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(&cnt, d + oCnt, 4, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    if (n_inliers) *n_inliers = cnt;
    return PLSLAM_OK;
}

int plslam_map2kf_point_gate(plslam_ctx* ctx, const plslam_cam* K, const double* Twf,
                             const double* Xw, const int32_t* matches_12, int32_t nq,
                             const double* pl, int32_t nt, double max_epip, uint8_t* mask,
                             int32_t* n_inliers)
{
    return gate_host(ctx, K, Twf, 0, Xw, matches_12, nq, pl, nt, max_epip, mask, n_inliers);
}

int plslam_map2kf_line_gate(plslam_ctx* ctx, const plslam_cam* K, const double* Twf,
                            const double* Lw, const int32_t* matches_12, int32_t nq,
                            const double* le, int32_t nt, double max_epip, uint8_t* mask,
                            int32_t* n_inliers)
{
    return gate_host(ctx, K, Twf, 1, Lw, matches_12, nq, le, nt, max_epip, mask, n_inliers);
}

static int visible_host(plslam_ctx* ctx, const plslam_cam* K, const double* Twf, const double* X,
                        int32_t n, int lines, uint8_t* vis)
{
    PLSLAM_REQUIRE(ctx && K && Twf && n >= 0, PLSLAM_EINVAL);
    if (n == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(X && vis, PLSLAM_EINVAL);
    const int lw = lines ? 6 : 3;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    Carver c;
    const size_t oX = c.take((size_t)n * lw * 8), oV = c.take((size_t)n);
    int rc;
    if ((rc = ctx->in_a.reserve(c.off))) return rc;
    char* d = ctx->in_a.as<char>();
    hipStream_t s = ctx->stream;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oX, X, (size_t)n * lw * 8, hipMemcpyHostToDevice, s));
    if ((rc = launch_visible(*K, Twf, (double*)(d + oX), n, lines, (uint8_t*)(d + oV), s))) return rc;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(vis, d + oV, (size_t)n, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return PLSLAM_OK;
}

int plslam_map_point_visible(plslam_ctx* ctx, const plslam_cam* K, const double* Twf,
                             const double* Xw, int32_t n, uint8_t* vis)
{
    return visible_host(ctx, K, Twf, Xw, n, 0, vis);
}

int plslam_map_line_visible(plslam_ctx* ctx, const plslam_cam* K, const double* Twf,
                            const double* Lw, int32_t n, uint8_t* vis)
{
    return visible_host(ctx, K, Twf, Lw, n, 1, vis);
}

// ---- representative descriptors ---------------------------------------------------------------
int plslam_median_desc_batched_dev(plslam_ctx* ctx, const uint8_t* desc_lists, const int32_t* offsets,
                                   int32_t n_lm, int32_t total, int32_t* med_idx, uint8_t* med_desc,
                                   void* stream)
{
    PLSLAM_REQUIRE(ctx && n_lm >= 0 && total >= 0, PLSLAM_EINVAL);
    if (n_lm == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(offsets && med_idx && (total == 0 || desc_lists), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(((uintptr_t)desc_lists & 3) == 0 && ((uintptr_t)med_desc & 3) == 0, PLSLAM_EINVAL);
    DeviceGuard g(ctx->device);
    return launch_median_desc(desc_lists, offsets, n_lm, total, med_idx, med_desc,
                              stream ? static_cast<hipStream_t>(stream) : ctx->stream);
}

int plslam_median_desc_batched(plslam_ctx* ctx, const uint8_t* desc_lists, const int32_t* offsets,
                               int32_t n_lm, int32_t* med_idx, uint8_t* med_desc)
{
    PLSLAM_REQUIRE(ctx && n_lm >= 0, PLSLAM_EINVAL);
    if (n_lm == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(offsets && med_idx && offsets[0] == 0, PLSLAM_EINVAL);
    for (int32_t l = 0; l < n_lm; ++l) {
        const int64_t n = (int64_t)offsets[l + 1] - offsets[l];
        PLSLAM_REQUIRE(n >= 0 && n < (1 << 23), PLSLAM_EINVAL);
    }
    const int32_t total = offsets[n_lm];
    PLSLAM_REQUIRE(total == 0 || desc_lists, PLSLAM_EINVAL);
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    Carver c;
    const size_t oD = c.take((size_t)total * 32), oO = c.take((size_t)(n_lm + 1) * 4),
                 oI = c.take((size_t)n_lm * 4), oM = c.take((size_t)n_lm * 32);
    int rc;
    if ((rc = ctx->in_a.reserve(c.off))) return rc;
    char* d = ctx->in_a.as<char>();
    hipStream_t s = ctx->stream;
    if (total) PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oD, desc_lists, (size_t)total * 32, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oO, offsets, (size_t)(n_lm + 1) * 4, hipMemcpyHostToDevice, s));
    if ((rc = launch_median_desc((const uint8_t*)(d + oD), (const int32_t*)(d + oO), n_lm, total,
                                 (int32_t*)(d + oI), med_desc ? (uint8_t*)(d + oM) : nullptr, s)))
        return rc;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(med_idx, d + oI, (size_t)n_lm * 4, hipMemcpyDeviceToHost, s));
    if (med_desc) PLSLAM_HIP_CHECK(hipMemcpyAsync(med_desc, d + oM, (size_t)n_lm * 32, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return PLSLAM_OK;
}

// ---- LBD binarisation ------------------------------------------------------------------------
int plslam_lbd_binarise_dev(plslam_ctx* ctx, const float* lbd_f32, int32_t n, uint8_t* desc_u8,
                            void* stream)
{
    PLSLAM_REQUIRE(ctx && n >= 0, PLSLAM_EINVAL);
    if (n == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(lbd_f32 && desc_u8, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(((uintptr_t)lbd_f32 & 15) == 0 && ((uintptr_t)desc_u8 & 15) == 0, PLSLAM_EINVAL);
    DeviceGuard g(ctx->device);
    return launch_lbd_binarise(lbd_f32, n, desc_u8,
                               stream ? static_cast<hipStream_t>(stream) : ctx->stream);
}

int plslam_lbd_binarise(plslam_ctx* ctx, const float* lbd_f32, int32_t n, uint8_t* desc_u8)
{
    PLSLAM_REQUIRE(ctx && n >= 0, PLSLAM_EINVAL);
    if (n == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(lbd_f32 && desc_u8, PLSLAM_EINVAL);
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    Carver c;
    const size_t oF = c.take((size_t)n * PLSLAM_LBD_FLOATS * 4), oC = c.take((size_t)n * 32);
    int rc;
    if ((rc = ctx->in_a.reserve(c.off))) return rc;
    char* d = ctx->in_a.as<char>();
    hipStream_t s = ctx->stream;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oF, lbd_f32, (size_t)n * PLSLAM_LBD_FLOATS * 4,
                                    hipMemcpyHostToDevice, s));
    if ((rc = launch_lbd_binarise((const float*)(d + oF), n, (uint8_t*)(d + oC), s))) return rc;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(desc_u8, d + oC, (size_t)n * 32, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return PLSLAM_OK;
}

// ---- host-to-host pipeline ---------------------------------------------------------------------
struct plslam_match_pipeline {
    plslam_ctx* ctx = nullptr;
    size_t arena_bytes = 0, out_entries = 0;
    int32_t nprob = 0, depth = 0;
    int64_t submitted = 0;
    hipStream_t s_up = nullptr, s_down = nullptr;
    struct Slot {
        DevBuf arena, out, cnt;
        plslam_match_plan plan;
        hipEvent_t up = nullptr, run = nullptr, done = nullptr;   // upload finished / kernels finished / download finished
        bool used = false;
    };
    std::vector<Slot> slots;
};

int plslam_match_pipeline_create(plslam_ctx* ctx, size_t arena_bytes, const plslam_arena_problem* probs, int32_t nprob,
                                 size_t out_entries, int32_t depth, plslam_match_pipeline** out)
{
    PLSLAM_REQUIRE(ctx && out && probs && nprob > 0 && depth >= 2 && depth <= 8 && arena_bytes > 0, PLSLAM_EINVAL);
    *out = nullptr;
    for (int32_t i = 0; i < nprob; ++i) {
        const plslam_arena_problem& q = probs[i];
        PLSLAM_REQUIRE(q.n1 >= 0 && q.n2 >= 0 && q.d1_off >= 0 && q.d2_off >= 0 && q.out_off >= 0, PLSLAM_EINVAL);
        PLSLAM_REQUIRE((q.d1_off & 3) == 0 && (q.d2_off & 3) == 0, PLSLAM_EINVAL);
        PLSLAM_REQUIRE((size_t)q.d1_off + (size_t)q.n1 * 32 <= arena_bytes && (size_t)q.d2_off + (size_t)q.n2 * 32 <= arena_bytes,
                       PLSLAM_EINVAL);
        PLSLAM_REQUIRE((size_t)q.out_off + (size_t)q.n1 <= out_entries, PLSLAM_EINVAL);
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    plslam_match_pipeline* P = new (std::nothrow) plslam_match_pipeline();
    PLSLAM_REQUIRE(P != nullptr, PLSLAM_ENOMEM);
    P->ctx = ctx; P->arena_bytes = arena_bytes; P->out_entries = out_entries; P->nprob = nprob; P->depth = depth;
    P->slots.resize((size_t)depth);
    int r = PLSLAM_OK;
    auto fail = [&](int code) { plslam_match_pipeline_destroy(P); return code; };
    if (hipStreamCreateWithFlags(&P->s_up, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&P->s_down, hipStreamNonBlocking) != hipSuccess)
        return fail(PLSLAM_EHIP);
    std::vector<plslam_match_problem> mp((size_t)nprob);
    for (auto& sl : P->slots) {
        if ((r = sl.arena.reserve(arena_bytes + 16)) || (r = sl.out.reserve(out_entries * 4 + 16)) ||
            (r = sl.cnt.reserve((size_t)nprob * 4)))
            return fail(r);
        // entries no problem covers (stride padding between tables) are copied to the host with every batch: the same in
        // every slot (-1), not whatever the allocation held
        if (hipMemsetAsync(sl.out.p, 0xFF, out_entries * 4 + 16, ctx->stream) != hipSuccess ||
            hipMemsetAsync(sl.cnt.p, 0, (size_t)nprob * 4, ctx->stream) != hipSuccess)
            return fail(PLSLAM_EHIP);
        if (hipEventCreateWithFlags(&sl.up, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sl.run, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess)
            return fail(PLSLAM_EHIP);
        for (int32_t i = 0; i < nprob; ++i) {
            const plslam_arena_problem& q = probs[i];
            plslam_match_problem& p = mp[(size_t)i];
            p.d1 = sl.arena.as<uint8_t>() + q.d1_off; p.d2 = sl.arena.as<uint8_t>() + q.d2_off;
            p.n1 = q.n1; p.n2 = q.n2; p.nnr = q.nnr; p.mutual = q.mutual ? 1 : 0;
            p.matches_12 = sl.out.as<int32_t>() + q.out_off;
            p.n_matches = sl.cnt.as<int32_t>() + i;
        }
        if ((r = plan_build(ctx, mp.data(), nprob, &sl.plan))) return fail(r);
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(PLSLAM_EHIP);      // the plans' table uploads
    *out = P;
    return PLSLAM_OK;
}

// device -> mapped host memory, written by the GPU itself
__global__ void __launch_bounds__(256) k_store_to_host(const int4* __restrict__ src, int4* __restrict__ dst, size_t n16,
                                                       const int32_t* __restrict__ src_tail, int32_t* __restrict__ dst_tail,
                                                       int ntail)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

int plslam_match_pipeline_submit(plslam_match_pipeline* P, const void* arena_host, int32_t* out_host, int32_t* counts_host)
{
    PLSLAM_REQUIRE(P && arena_host && out_host, PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    plslam_match_pipeline::Slot& sl = P->slots[(size_t)(P->submitted % P->depth)];
    // the slot's previous batch: its tables must have left the device buffers (its kernels have then finished too)
    if (sl.used) PLSLAM_HIP_CHECK(hipEventSynchronize(sl.done));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(sl.arena.p, arena_host, P->arena_bytes, hipMemcpyHostToDevice, P->s_up));
    PLSLAM_HIP_CHECK(hipEventRecord(sl.up, P->s_up));
    PLSLAM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, sl.up, 0));
    int r = plan_run(&sl.plan, ctx->stream, ctx->stream);
    if (r) return r;
    // Results go home.  Page-locked host memory is written by a kernel on the compute stream, right behind the finalize:
    // a copy-engine download would queue between two uploads, and with the copy engines taking transfers in order the
    // next upload then waits for this batch's kernels -- upload, kernels, download ran strictly one after the other
    // (measured: 1.03 ms per batch of 256 C2 pairs; 0.53 ms = the upload alone once the download is a kernel).
    void* d_out = plslam::mapped_device_pointer(out_host);
    void* d_cnt = counts_host ? plslam::mapped_device_pointer(counts_host) : nullptr;
    // (the kernel stores 16 bytes at a time: a page-locked but offset pointer -- a slice of a pinned buffer -- takes the copy path)
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_cnt)) & 15) == 0;
    if (d_out && (!counts_host || d_cnt) && aligned16 && (P->out_entries % 4) == 0 && P->nprob <= 256 * 1024) {
        const size_t n16 = P->out_entries / 4;
        hipLaunchKernelGGL(k_store_to_host, dim3(64), dim3(256), 0, ctx->stream, sl.out.as<int4>(), (int4*)d_out, n16,
                           (const int32_t*)nullptr, (int32_t*)nullptr, 0);
        if (d_cnt)   // the counters: a second tiny launch keeps the kernel's interface trivial
            hipLaunchKernelGGL(k_store_to_host, dim3((unsigned)((P->nprob + 1023) / 1024)), dim3(256), 0, ctx->stream,
                               sl.cnt.as<int4>(), (int4*)d_cnt, (size_t)P->nprob / 4, sl.cnt.as<int32_t>() + (P->nprob & ~3),
                               (int32_t*)d_cnt + (P->nprob & ~3), P->nprob & 3);
        PLSLAM_HIP_CHECK(hipGetLastError());
        PLSLAM_HIP_CHECK(hipEventRecord(sl.done, ctx->stream));
    } else {        // pageable host memory: the runtime stages the copy itself
        PLSLAM_HIP_CHECK(hipEventRecord(sl.run, ctx->stream));
        PLSLAM_HIP_CHECK(hipStreamWaitEvent(P->s_down, sl.run, 0));
        PLSLAM_HIP_CHECK(hipMemcpyAsync(out_host, sl.out.p, P->out_entries * 4, hipMemcpyDeviceToHost, P->s_down));
        if (counts_host)
            PLSLAM_HIP_CHECK(hipMemcpyAsync(counts_host, sl.cnt.p, (size_t)P->nprob * 4, hipMemcpyDeviceToHost, P->s_down));
        PLSLAM_HIP_CHECK(hipEventRecord(sl.done, P->s_down));
    }
    sl.used = true;
    ++P->submitted;
    return PLSLAM_OK;
}

int plslam_match_pipeline_wait(plslam_match_pipeline* P)
{
    PLSLAM_REQUIRE(P != nullptr, PLSLAM_EINVAL);
    DeviceGuard g(P->ctx->device);
    for (auto& sl : P->slots)
        if (sl.used) PLSLAM_HIP_CHECK(hipEventSynchronize(sl.done));
    return PLSLAM_OK;
}

void plslam_match_pipeline_destroy(plslam_match_pipeline* P)
{
    if (!P) return;
    DeviceGuard g(P->ctx->device);
    (void)hipDeviceSynchronize();
    for (auto& sl : P->slots) {
        sl.arena.release(); sl.out.release(); sl.cnt.release();
        sl.plan.free_all();
        if (sl.up) (void)hipEventDestroy(sl.up);
        if (sl.run) (void)hipEventDestroy(sl.run);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (P->s_up) (void)hipStreamDestroy(P->s_up);
    if (P->s_down) (void)hipStreamDestroy(P->s_down);
    delete P;
}

void* plslam_pinned_alloc(size_t bytes)
{
    void* p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void plslam_pinned_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

// ---- RCCL gather -----------------------------------------------------------------------------
namespace {
typedef int (*nccl_group_fn)(void);
typedef int (*nccl_p2p_fn)(void*, size_t, int, int, void*, hipStream_t);  // send/recv share a shape
struct Rccl {
    void* h = nullptr;
    nccl_group_fn group_start = nullptr, group_end = nullptr;
    nccl_p2p_fn send = nullptr, recv = nullptr;
    bool tried = false;
} g_rccl;
std::mutex g_rccl_mu;

std::string g_rccl_path;          // plslam_rccl_use(): the library file the host's communicators come from

bool rccl_load()
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl.tried) return g_rccl.h != nullptr;
    g_rccl.tried = true;
    // the communicator a caller hands in belongs to ONE loaded copy of RCCL (PyTorch ships its own beside /opt/rocm's): the
    // entry points must come from that copy -- plslam_rccl_use(path) names it; without it the usual search order
    const char* names[] = {g_rccl_path.empty() ? nullptr : g_rccl_path.c_str(), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : names) {
        if (!n) continue;
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
    }
    if (!g_rccl.h) return false;
    g_rccl.group_start = (nccl_group_fn)dlsym(g_rccl.h, "ncclGroupStart");
    g_rccl.group_end = (nccl_group_fn)dlsym(g_rccl.h, "ncclGroupEnd");
    g_rccl.send = (nccl_p2p_fn)dlsym(g_rccl.h, "ncclSend");
    g_rccl.recv = (nccl_p2p_fn)dlsym(g_rccl.h, "ncclRecv");
    if (!g_rccl.group_start || !g_rccl.group_end || !g_rccl.send || !g_rccl.recv) {
        dlclose(g_rccl.h);
        g_rccl.h = nullptr;
    }
    return g_rccl.h != nullptr;
}
}  // namespace

// which librccl the communicators of this process come from (before the first gather; NULL / "" = the default search)
int plslam_rccl_use(const char* path)
{
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    PLSLAM_REQUIRE(!g_rccl.tried, PLSLAM_EINVAL);          // (already loaded: too late to choose)
    g_rccl_path = path ? path : "";
    return PLSLAM_OK;
}

// 1 when librccl's group / send / recv entry points are loaded (loads them on the first call), 0 when they cannot be: a caller that
// is about to take the one-call gather step asks BEFORE its first step -- a failure there is a set-up failure, not a step's
int plslam_rccl_available(void)
{
    return rccl_load() ? 1 : 0;
}

namespace plslam {
__global__ void __launch_bounds__(256) k_widen16(const int16_t* __restrict__ src, int32_t* __restrict__ dst, int64_t n)
{
    // four entries per lane: 8 bytes in, 16 out
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const short4 v = *reinterpret_cast<const short4*>(src + i);
        *reinterpret_cast<int4*>(dst + i) = make_int4(v.x, v.y, v.z, v.w);
    } else {
        for (int64_t k = i; k < n; ++k) dst[k] = src[k];
    }
}
}  // namespace plslam

// One step of the N > 1 path in ONE call (round 6; the Python of round 5 spent 0.06-0.2 ms of host time per 0.3 ms step on it):
// plan run (scan on scan_stream, everything behind it on post_stream) -> gather of the finished table to `root` as ONE ncclGroup
// of point-to-point transfers (each peer on its own xGMI link) -> on the root with the int16 wire format the widening to the
// int32 tables of the C ABI -- everything enqueued, nothing waited for on the host.  The plan remembers its gather: the next step
// of the SAME plan makes both of its streams wait for it before anything rewrites the table (also the scan stream: column-split
// and fused plans write the table from the scan kernel -- ADVICE r5).
int plslam_match_plan_step_gather(plslam_match_plan* plan, const plslam_gather_step* st)
{
    PLSLAM_REQUIRE(plan && st, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(st->nranks > 0 && st->rank >= 0 && st->rank < st->nranks && st->root >= 0 && st->root < st->nranks, PLSLAM_EINVAL);
    PLSLAM_REQUIRE((st->wire_bytes == 4 || st->wire_bytes == 2) && st->n_entries >= 0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(st->n_entries == 0 || st->send, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(st->rank != st->root || st->n_entries == 0 || st->recv, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(st->nranks == 1 || st->comm, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(st->scan_stream && st->post_stream && st->scan_stream != st->post_stream, PLSLAM_EINVAL);
    // (one rank WITH a communicator: the rank sends its table to itself through RCCL -- what a forced one-rank group measures is
    // the collective's launch, as at N > 1; without one the slice is copied)
    const bool use_rccl = st->comm != nullptr && st->n_entries > 0;
    if (use_rccl && !rccl_load()) {
        set_last_error("librccl.so could not be loaded: %s", dlerror());
        return PLSLAM_ENOTSUP;
    }
    DeviceGuard g(plan->ctx->device);
    hipStream_t ss = static_cast<hipStream_t>(st->scan_stream), sp = static_cast<hipStream_t>(st->post_stream);
    hipStream_t sc = st->comm_stream ? static_cast<hipStream_t>(st->comm_stream) : sp;
    if (!plan->gather_done) PLSLAM_HIP_CHECK(hipEventCreateWithFlags(&plan->gather_done, hipEventDisableTiming));
    if (plan->gather_pending) {
        PLSLAM_HIP_CHECK(hipStreamWaitEvent(ss, plan->gather_done, 0));
        PLSLAM_HIP_CHECK(hipStreamWaitEvent(sp, plan->gather_done, 0));
        plan->gather_pending = false;
    }
    int r = plan->fused ? plan_run(plan, ss, ss) : plan_run(plan, ss, sp);
    if (r) return r;
    hipStream_t last = plan->fused ? ss : sp;             // the stream the table is complete on
    if (sc != last) {
        // (post_done is recorded by a split run; a fused plan's table is complete on the scan stream)
        if (plan->fused) {
            if (!plan->post_done) PLSLAM_HIP_CHECK(hipEventCreateWithFlags(&plan->post_done, hipEventDisableTiming));
            PLSLAM_HIP_CHECK(hipEventRecord(plan->post_done, ss));
        }
        PLSLAM_HIP_CHECK(hipStreamWaitEvent(sc, plan->post_done, 0));
    }
    const size_t bytes = (size_t)st->n_entries * (size_t)st->wire_bytes;
    const bool root = st->rank == st->root;
    const bool self_send = use_rccl && st->nranks == 1 && static_cast<char*>(st->recv) != st->send;
    if (use_rccl && (st->nranks > 1 || self_send)) {
        const int ncclInt8 = 0;
        int rc = g_rccl.group_start();
        if (root) {
            for (int p = 0; p < st->nranks && rc == 0; ++p) {
                if (p == st->root && !self_send) continue;
                rc = g_rccl.recv(static_cast<char*>(st->recv) + (size_t)p * bytes, bytes, ncclInt8, p, st->comm, sc);
            }
            if (self_send && rc == 0) rc = g_rccl.send(const_cast<void*>(st->send), bytes, ncclInt8, st->root, st->comm, sc);
        } else if (rc == 0) {
            rc = g_rccl.send(const_cast<void*>(st->send), bytes, ncclInt8, st->root, st->comm, sc);
        }
        const int rc2 = g_rccl.group_end();
        if (rc || rc2) {
            set_last_error("RCCL point-to-point gather failed (ncclResult %d/%d)", rc, rc2);
            return PLSLAM_EHIP;
        }
    }
    if (root && bytes) {
        char* mine = static_cast<char*>(st->recv) + (size_t)st->root * bytes;
        if (mine != st->send && !self_send) PLSLAM_HIP_CHECK(hipMemcpyAsync(mine, st->send, bytes, hipMemcpyDeviceToDevice, sc));
        if (st->wire_bytes == 2 && st->wide) {
            const int64_t n = st->n_entries * st->nranks;
            hipLaunchKernelGGL(plslam::k_widen16, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, sc,
                               static_cast<const int16_t*>(st->recv), st->wide, n);
            PLSLAM_HIP_CHECK(hipGetLastError());
        }
    }
    PLSLAM_HIP_CHECK(hipEventRecord(plan->gather_done, sc));
    plan->gather_pending = true;
    return PLSLAM_OK;
}

// host wait for the plan's last plslam_match_plan_step_gather (table gathered, root's widening done)
int plslam_match_plan_gather_sync(plslam_match_plan* plan)
{
    PLSLAM_REQUIRE(plan != nullptr, PLSLAM_EINVAL);
    if (!plan->gather_done || !plan->gather_pending) return PLSLAM_OK;
    DeviceGuard g(plan->ctx->device);
    PLSLAM_HIP_CHECK(hipEventSynchronize(plan->gather_done));
    return PLSLAM_OK;
}

int plslam_gather_match_tables(plslam_ctx* ctx, void* comm, int nranks, int rank, int root,
                               const int32_t* local, int64_t n_local, int32_t* gathered,
                               void* stream)
{
    PLSLAM_REQUIRE(ctx && comm && nranks > 0 && rank >= 0 && rank < nranks, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(root >= 0 && root < nranks && n_local >= 0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(n_local == 0 || local, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(rank != root || n_local == 0 || gathered, PLSLAM_EINVAL);
    if (!rccl_load()) {
        set_last_error("librccl.so could not be loaded: %s", dlerror());
        return PLSLAM_ENOTSUP;
    }
    DeviceGuard g(ctx->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    const int ncclInt32 = 2;  // ncclDataType_t: ncclInt8=0, ncclUint8=1, ncclInt32=2
    int rc = g_rccl.group_start();
    if (rank == root) {
        for (int p = 0; p < nranks && rc == 0; ++p) {
            if (p == root) continue;
            rc = g_rccl.recv(gathered + (size_t)p * n_local, (size_t)n_local, ncclInt32, p, comm, s);
        }
    } else if (rc == 0) {
        rc = g_rccl.send(const_cast<int32_t*>(local), (size_t)n_local, ncclInt32, root, comm, s);
    }
    const int rc2 = g_rccl.group_end();
    if (rc || rc2) {
        set_last_error("RCCL point-to-point gather failed (ncclResult %d/%d)", rc, rc2);
        return PLSLAM_EHIP;
    }
    if (rank == root && n_local)
        PLSLAM_HIP_CHECK(hipMemcpyAsync(gathered + (size_t)root * n_local, local, (size_t)n_local * 4,
                                        hipMemcpyDeviceToDevice, s));
    return PLSLAM_OK;
}

}  // extern "C"
