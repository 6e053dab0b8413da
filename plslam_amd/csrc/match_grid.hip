// K14 -- StVO::matchGrid, the windowed ("fast_matching") matcher, as ONE kernel launch per batch of problems.
//
// Reference: stvo-pl matching.cpp::matchGrid (both overloads) + gridStructure.cpp::GridStructure::get -- the
// un-vendored dependency, [RECALL]; call sites src/mapHandler.cpp:271 (points KF<->KF), :418 (lines), :591 (map
// points <-> KF), :706 (map lines <-> KF); grids filled by the callers at :258-264, :395-411, :580-584, :683-699.
//
// Upstream is a sequential loop over the rows i1 of desc1: the candidates of row i1 are the grid items inside the
// window(s) around its cell(s); with Config::bestLRMatches() a candidate (i1, i2) is SKIPPED unless its distance is
// strictly below the smallest distance any earlier row had to i2 (`if (d < distances[i2]) {...} else continue;`),
// and matches_21[i2] follows the rows that lowered distances[i2].  That loop-carried dependence has an order-free
// form, which is what runs here:
//     (i1, i2) takes part in row i1's best / second best   <=>   no row i1' < i1 has d(i1', i2) <= d(i1, i2),
// i.e. the live entries of a column are its strict prefix minima in row order (its "records": rows ascending,
// distances strictly descending, ending in the column's lexicographic minimum (d, i1) = matches_21[i2]).  The
// tests check this form against a literal restatement of the sequential loop.  Ties between equally distant best
// candidates go to the lowest i2 (upstream: the iteration order of a std::unordered_set<int>, implementation-
// defined).
//
// One workgroup of 1024 lanes per problem (256 for problems of at most 256 rows).  A problem is a few 10^4 distances:
// what costs is the length of dependent memory chains and the address rate of scattered loads, so the grid's cell_start,
// its items, the desc2 rows (and line directions) and the per-column / per-row words live in LDS whenever they fit
// (64 x 48 cells, 1500 + 1500 rows: 96 KB); the same code runs on global scratch when they do not.
//   PA  a lane per row: candidates in batches of 4, d = popcount(desc1[i1] ^ desc2[i2]).  Without bestLRMatches the
//       best two (d << 23 | i2) keys are folded on the spot.  With it every candidate proposes itself as its column's
//       first record (atomic min of i1 << 9 | d: the smallest row wins) and is kept for the record passes:
//       * FLAT mode (everything in LDS and row + column numbers of at most 23 bits together -- 4096 x 2048, 8192 x 1024 ..:
//         the shipped sizes): a candidate is one word that names its row, d << (b1 + b2) | i1 << b2 | i2, appended to the region of the wave that found it in the LDS that is still free
//         (its share of the global store takes what does not fit).  colbest[i2] = the smallest (d, i1) shown for the column
//         so far: a candidate that an EARLIER row matches or beats is dead whatever happens later and is never stored.
//       * otherwise: one word per candidate in the lane's slots of a transposed global store -- slot k of lane t at
//         [k][t], every access coalesced.
//   PB  record passes (bestLRMatches only).  A candidate that IS its column's newest record is live: it joins its row's
//       best two and leaves; one that is still below the record's distance proposes itself as the next record (the
//       smallest ROW below the last record's distance) and stays; the others are dropped.  About ln(candidates per column)
//       + 2 passes over geometrically shrinking lists; the last record of a column is its lexicographic minimum
//       (d, i1) = matches_21.
//       * FLAT mode: candidate-parallel -- each wave streams its own region and compacts the survivors in place (ballot +
//         prefix), so the lanes are evenly loaded whatever the rows' list lengths; records carry their pass number in the
//         top bits and alternate between two arrays, so nothing is cleared or installed between passes (one barrier per
//         pass); rows' best two via two LDS atomic mins; when at most 256 candidates are left one wave finishes alone.
//       * otherwise: a lane streams its rows' slots (row-private best two), a sweep over the columns installs the
//         proposals between passes.
//   PC  per row: ratio test `best_d < best_d2 * nnr` in fp64 (int * double upstream; best_d2 = INT_MAX when absent,
//       so a lone live candidate passes), mutual check against the column's final state, count.
// Fences: every exchange through global memory in this kernel (candidate store, the scratch tables of the modes that do not
// fit LDS) is between lanes of ONE workgroup, so the fences are workgroup-scope (__threadfence_block).  A device-scope fence
// is an L2 write-back + invalidate on this chip (buffer_wbl2 / buffer_inv), whose cost grows with what every OTHER workgroup
// on the die has written: with it a problem took 107 us among 127 others against 66 us alone.
// Duplicated candidates (an item sitting in several cells of the window, the two windows of a line overlapping) are
// harmless: the atomic min and the best-two fold are idempotent.
#include <algorithm>
#include <cstring>
#include <new>

#include "common.hpp"

namespace plslam {
namespace {

constexpr int GRID_THREADS = 1024;
constexpr uint32_t REC_D_BITS = 9, REC_D_MASK = 511u;                  // column state: i1 << 9 | d
constexpr int CB = 4;                                                  // candidates per batch
constexpr int GRID_SPLIT = 4;                                          // flat PA: at most this many lanes share a row's window columns
constexpr uint32_t GRID_TAIL = 256;                                   // candidates left when one wave finishes the passes alone
constexpr int PB_BATCH = 8;                                            // stored candidates per batch of a record pass
#ifndef PLSLAM_GRID_RECORDS
#define PLSLAM_GRID_RECORDS 1       // 0: experiment builds that list every candidate pair of a lone problem (k_grid_candidates)
#endif
#ifndef PLSLAM_GRID_RUNS_PER_COLUMN
#define PLSLAM_GRID_RUNS_PER_COLUMN 4   // k_grid_records' list is bucketed by column while a column has at most this many runs on average
#endif
#ifndef PLSLAM_GRID_FAST
#define PLSLAM_GRID_FAST 1          // 0: experiment builds without the shortest bookkeeping of k_grid_records' list
#endif
#ifndef PLSLAM_GRID_COLUMNS
#define PLSLAM_GRID_COLUMNS 1       // 0: experiment builds without the column-bucketed path of a lone problem
#endif
constexpr uint32_t REC_SLOT = 8;                                       // k_grid_records: list words per item of the grid
constexpr size_t GRID_LDS_MAX_BYTES = 152 * 1024;                      // dynamic LDS of the LDS instantiations
constexpr size_t GRID_LDS_FIXED_MAX_BYTES = 144 * 1024;                // tables that MUST fit for MODE 1

// Pointers read out of the problem table are generic to the compiler (it would emit FLAT instructions and, for the
// mode-dependent ones, could not tell LDS from global memory): every pointer below carries its address space.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // a builtin vector (HIP's uint4 class cannot live behind an address space)
#define PLSLAM_AS_GLOBAL __attribute__((address_space(1)))
#define PLSLAM_AS_LDS __attribute__((address_space(3)))
template <class T, bool IN_LDS> struct as_ptr { using type = PLSLAM_AS_GLOBAL T*; };
template <class T> struct as_ptr<T, true> { using type = PLSLAM_AS_LDS T*; };

template <int MODE>   // 0: every table in global scratch; 1: cell_start + column / row words in LDS; 2: items and desc2 rows too
struct GridPtrs {
    typename as_ptr<const uint32_t, (MODE >= 1)>::type cs;      // cell_start
    typename as_ptr<const int32_t, (MODE == 2)>::type items;    // cell_items
    typename as_ptr<const u32x4, (MODE == 2)>::type d2;         // desc2 rows, 2 x 16 bytes
    typename as_ptr<uint32_t, (MODE >= 1)>::type state, next, row_k1, row_k2;
    PLSLAM_AS_GLOBAL const int32_t* centres;
    PLSLAM_AS_GLOBAL const double* dir1;
    typename as_ptr<const double, (MODE == 2)>::type dir2;      // directions of the desc2 lines
};

__device__ __forceinline__ void best2_fold(uint32_t& k1, uint32_t& k2, uint32_t key)
{   // idempotent insertion into the two smallest distinct keys (k1 <= k2); value selects only -- a branchy form makes
    // the compiler address k1 / k2 through private memory
    const uint32_t lo = key < k1 ? key : k1, hi = key < k1 ? k1 : key;
    k2 = key == k1 ? k2 : (hi < k2 ? hi : k2);
    k1 = lo;
}

// Slot k of lane `tid` in the transposed candidate store of a round: row k of a [depth][NT] array, rotated by one
// wave per row -- a wave's successive slots then fall into different 256-byte channels of L2 / HBM instead of all
// into the same one (row pitch 4 KB = 16 channels x 256 B)
template <int NT>
__device__ __forceinline__ size_t slot_index(uint32_t k, int tid)
{
    return (size_t)k * NT + ((uint32_t)(tid + (k << 6)) & (uint32_t)(NT - 1));
}

struct RowWindows {     // GridStructure::get ranges of one window centre (clamped to the grid: they fit 32 bits)
    int32_t min_x, max_x, min_y, max_y;
};
__device__ __forceinline__ RowWindows window_of(const GridDesc& g, PLSLAM_AS_GLOBAL const int32_t* p)
{
    const int64_t x = p[0], y = p[1];
    RowWindows r;   // the sums in 64 bits: centres and windows may be any int32
    r.min_x = (int32_t)(x - g.w[0] > 0 ? (x - g.w[0] < g.cols ? x - g.w[0] : g.cols) : 0);
    r.max_x = (int32_t)(x + g.w[1] + 1 < g.cols ? (x + g.w[1] + 1 > 0 ? x + g.w[1] + 1 : 0) : g.cols);
    r.min_y = (int32_t)(y - g.w[2] > 0 ? (y - g.w[2] < g.rows ? y - g.w[2] : g.rows) : 0);
    r.max_y = (int32_t)(y + g.w[3] + 1 < g.rows ? (y + g.w[3] + 1 > 0 ? y + g.w[3] + 1 : 0) : g.rows);
    return r;
}

// number of grid items inside row i1's windows (duplicates, out-of-range items and candidates the direction test
// will drop included): the upper bound its slots in the candidate store are sized by
template <int MODE>
__device__ __forceinline__ uint32_t count_items(const GridDesc& g, const GridPtrs<MODE>& P, int32_t i1)
{
    uint32_t n = 0;
    for (int32_t c = 0; c < g.n_centres; ++c) {
        const RowWindows r = window_of(g, P.centres + ((int64_t)i1 * g.n_centres + c) * 2);
        if (r.min_y >= r.max_y) continue;
        for (int32_t x_ = r.min_x; x_ < r.max_x; ++x_) n += P.cs[x_ * g.rows + r.max_y] - P.cs[x_ * g.rows + r.min_y];
    }
    return n;
}

// GridStructure::get over every window centre of row i1, in batches: f(i2[CB]) with i2[j] = -1 for the slots
// that are empty or fail `if (i2 < 0 || i2 >= desc2.rows) continue;` / the direction test of the line overload
// (part, split): only the window columns min_x + part, + split, ... -- a row's window shared out over `split` lanes
template <int MODE, class F>
__device__ __forceinline__ void for_candidates(const GridDesc& g, const GridPtrs<MODE>& P, int32_t i1, F&& f, int32_t part = 0,
                                               int32_t split = 1)
{
    double a0 = 0.0, a1 = 0.0;
    const bool dirs = g.dir1 != nullptr && g.dir2 != nullptr;
    if (dirs) {
        a0 = P.dir1[2 * (int64_t)i1];
        a1 = P.dir1[2 * (int64_t)i1 + 1];
    }
    for (int32_t c = 0; c < g.n_centres; ++c) {
        const RowWindows r = window_of(g, P.centres + ((int64_t)i1 * g.n_centres + c) * 2);
        if (r.min_y >= r.max_y) continue;
        for (int32_t x_ = r.min_x + part; x_ < r.max_x; x_ += split) {
            // cells (x_, min_y .. max_y-1) are adjacent in the CSR order (id = x*rows + y)
            const int32_t s = (int32_t)P.cs[x_ * g.rows + r.min_y], e = (int32_t)P.cs[x_ * g.rows + r.max_y];
            for (int32_t k = s; k < e; k += CB) {
                int32_t i2[CB];
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    i2[j] = k + j < e ? P.items[k + j] : -1;
                    if ((uint32_t)i2[j] >= (uint32_t)g.n2) i2[j] = -1;
                }
                if (dirs) {
                    double b0[CB], b1[CB];
#pragma unroll
                    for (int j = 0; j < CB; ++j) {
                        const int64_t t = i2[j] < 0 ? 0 : i2[j];
                        b0[j] = P.dir2[2 * t];
                        b1[j] = P.dir2[2 * t + 1];
                    }
#pragma unroll
                    for (int j = 0; j < CB; ++j) {
                        const double dot = a0 * b0[j] + a1 * b1[j];
                        if (fabs(dot) < g.sim_th) i2[j] = -1;     // NaN (zero-length direction) compares false: kept
                    }
                }
                f(i2);
            }
        }
    }
}

// NT lanes per workgroup: 1024, or 256 for problems of at most 256 rows (a 200-line problem would leave 12 of 16 waves
// idle at every barrier -- and, in a batch, occupy a whole CU)
template <int MODE, int NT, bool BYVAL = false>
__global__ __launch_bounds__(NT) void k_match_grid(const GridDesc* __restrict__ probs, uint32_t lds_words, const uint32_t* __restrict__ pre_arg,
                                                   uint32_t pre_slots, const GridDesc one, const int32_t* __restrict__ n1_dev)
{
    // BYVAL: ONE problem, its descriptor by value in `one` (a dependent round trip less in front of everything; a run-time
    // choice between the two copies a 152-byte struct through private memory).  n1_dev (BYVAL only): where the row count lives
    // when a kernel upstream decides it -- one.n1 is then its upper bound, and still what the scratch layout is counted by.
    // pre_slots > 0: the list is k_grid_records': pre_slots words per item of the grid's CSR list (records first, KEY_NONE behind
    // them), then pre[0] more words; 0: k_grid_candidates' dense list of pre[0] words.
    // pre != nullptr (one LDS-resident mutual problem alone on the chip, MODE 2): PA's distances were evaluated by
    // k_grid_candidates on many workgroups -- every candidate word lies in the second half of the problem's candidate store,
    // pre[0] = their number --, this kernel does the bookkeeping over them (candidate-parallel: first records, the filter of
    // candidates an earlier row matches or beats, the waves' shares), then runs the record passes and PC.
    // (The two per-column atomic minima of the bookkeeping as GLOBAL atomics in k_grid_candidates instead: that kernel 10.5 ->
    // 16.1 us, this one 42.6 -> 39.4 us -- device-scope atomics are slower than one CU's LDS pipe is busy.)
    constexpr bool LDS = MODE >= 1;
    extern __shared__ u32x4 s_dyn4[];
    __shared__ uint32_t s_part[NT];
    __shared__ uint32_t s_max2[2];
    __shared__ uint32_t s_tail[GRID_TAIL];
    __shared__ uint32_t s_cur[NT / 64];              // flat mode: candidates appended to each wave's region
    __shared__ uint32_t s_seg[NT / 64 + 1];          // flat mode: first LDS word of each wave's region
    PLSLAM_AS_LDS uint32_t* s_dyn = (PLSLAM_AS_LDS uint32_t*)reinterpret_cast<uint32_t*>(s_dyn4);

    GridDesc g = BYVAL ? one : probs[blockIdx.x];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int32_t n1_layout = g.n1;
    if (BYVAL && n1_dev) {
        const int32_t n1_now = *(PLSLAM_AS_GLOBAL const int32_t*)n1_dev;
        g.n1 = n1_now >= 0 && n1_now < n1_layout ? n1_now : n1_layout;
    }
    const int32_t n1 = g.n1, n2 = g.n2;
    const int32_t ncell = g.cols * g.rows;
    const int32_t n_rounds = (n1 + NT - 1) / NT;
#ifdef PLSLAM_GRID_TIMING   // experiment builds only: phase boundaries in 10 ns ticks, printed by one lane
    uint64_t ts[6], t_move = 0, tp[16], tc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t dbg_total = 0;
    uint64_t t_cnt = 0;
    int ntc = 0;
#define COLS_STAMP() do { if (ntc < 8) tc[ntc++] = wall_clock64(); } while (0)
    const uint64_t c_start = clock64();
    int nts = 0, npass = 0;
#define GRID_STAMP() do { if (nts < 6) ts[nts++] = wall_clock64(); } while (0)
#else
#define GRID_STAMP() do { } while (0)
#define COLS_STAMP() do { } while (0)
#endif
    GRID_STAMP();
    PLSLAM_AS_GLOBAL uint32_t* gscratch = (PLSLAM_AS_GLOBAL uint32_t*)g.scratch;
    PLSLAM_AS_GLOBAL const int32_t* g_cell_start = (PLSLAM_AS_GLOBAL const int32_t*)g.cell_start;
    PLSLAM_AS_GLOBAL const int32_t* g_items = (PLSLAM_AS_GLOBAL const int32_t*)g.cell_items;
    PLSLAM_AS_GLOBAL const u32x4* g_d1 = (PLSLAM_AS_GLOBAL const u32x4*)g.d1;
    PLSLAM_AS_GLOBAL const u32x4* g_d2 = (PLSLAM_AS_GLOBAL const u32x4*)g.d2;
    PLSLAM_AS_GLOBAL int32_t* g_matches = (PLSLAM_AS_GLOBAL int32_t*)g.matches_12;
    // tables: state n2 | next n2 | row_k1 n1 | row_k2 n1 | [cell_start copy (LDS only)]
    const uint32_t fixed_words = (uint32_t)(LDS ? ncell + 1 : 0) + 2u * (uint32_t)n2 + 2u * (uint32_t)n1;
    // MODE 2: items at the next 16-byte boundary behind the tables, desc2 rows behind them (launcher guarantees the fit)
    const uint32_t items_off = (fixed_words + 3u) & ~3u, d2_off = (items_off + (uint32_t)g.n_items + 3u) & ~3u;
    GridPtrs<MODE> P;
    P.centres = (PLSLAM_AS_GLOBAL const int32_t*)g.centres;
    P.dir1 = (PLSLAM_AS_GLOBAL const double*)g.dir1;
    const bool has_dirs = g.dir1 != nullptr && g.dir2 != nullptr;
    if constexpr (LDS) {    // column / row words first: what lies behind them is free once PA is done
        P.state = s_dyn;
        P.cs = s_dyn + 2 * (n2 + n1);
    } else {
        P.cs = (PLSLAM_AS_GLOBAL const uint32_t*)g.cell_start;
        P.state = gscratch;
    }
    P.next = P.state + n2;                                  // this pass's proposals; state = newest record i1 << 9 | d
    P.row_k1 = P.next + n2;                                 // (d << 23 | i2) best live candidate of the row
    P.row_k2 = P.row_k1 + n1;                               // second best
    if constexpr (MODE == 2) {
        P.items = (PLSLAM_AS_LDS const int32_t*)(s_dyn + items_off);
        P.d2 = (PLSLAM_AS_LDS const u32x4*)(s_dyn + d2_off);
        P.dir2 = (PLSLAM_AS_LDS const double*)(s_dyn + d2_off + 8u * (uint32_t)n2);    // 16-byte aligned: d2_off is
    } else {
        P.items = g_items;
        P.d2 = g_d2;
        P.dir2 = (PLSLAM_AS_GLOBAL const double*)g.dir2;
    }
    // global scratch behind the tables: per-row slot counts, per-round slot depth, the candidate store (x 2)
    PLSLAM_AS_GLOBAL uint32_t* rcnt = gscratch + (LDS ? 0u : fixed_words);   // n1
    PLSLAM_AS_GLOBAL uint32_t* round_k = rcnt + n1_layout;                   // n_rounds
    PLSLAM_AS_GLOBAL uint32_t* store = round_k + (n1_layout + NT - 1) / NT;  // 2 x pair_cap words; round r at 1024 * sum_{r' < r} round_k

    // ---- "flat" mode (everything in LDS, bestLRMatches, row and column numbers of at most 23 bits together): a candidate is ONE
    // word that names its row, d << (fb1 + fb2) | i1 << fb2 | i2 (fb2 = bits of a column number, fb1 = what is left, at most
    // 14: a column's record needs 9 bits for its pass number above i1 << 9 | d); PA appends the candidates of a wave's rows to the wave's own region of the
    // LDS that is still free (spilling into its share of the global store if it must), the record passes run on those
    // regions.  colbest[i2] = the smallest (d, i1) any row has shown for the column so far: a candidate that some EARLIER row
    // matches or beats is dead whatever else happens and is never stored.
    constexpr uint32_t NW = NT / 64;
    const uint32_t wv = (uint32_t)tid >> 6;
    bool flat = false;
    uint32_t fb1 = 11, fb2 = 11;                    // bits of a row / column number in the flat words
    uint32_t seg_words = 0, tail_cap = 0, reg_off = 0;
    PLSLAM_AS_LDS uint32_t* colbest = nullptr;
    // (the launcher may know an upper bound of n1 only -- the row count then comes from the device, GridDesc::n1 patched behind
    // the upload --: whether the candidates were listed by k_grid_candidates is decided here, by the same test as there)
    const uint32_t* pre = pre_arg;
    if constexpr (MODE == 2) {
        fb2 = 1;
        while (fb2 < 22 && (1u << fb2) < (uint32_t)n2) ++fb2;
        fb1 = 23u - fb2 > 14u ? 14u : 23u - fb2;
        flat = g.mutual && (uint32_t)n2 <= (1u << fb2) && (uint32_t)n1 <= (1u << fb1);
        if (!flat) pre = nullptr;
        // (pre: neither the items nor the desc2 rows are needed here -- their LDS goes to the candidates)
        const uint32_t pa_end = pre ? items_off : d2_off + 8u * (uint32_t)n2 + (has_dirs ? 4u * (uint32_t)n2 : 0u);
        colbest = s_dyn + pa_end;
        reg_off = pa_end + (uint32_t)n2;
        tail_cap = (uint32_t)g.pair_cap / NW;
    }
    // The free LDS is shared out in proportion to the (row, window part) tasks a wave owns (task t belongs to lane t % NT).
    // seg_at(w) = first word of wave w's region, seg_at(NW) = end.
    const uint32_t free_words = flat && lds_words > reg_off ? lds_words - reg_off : 0u;
    // flat PA: a row's window columns go to 4 lanes (GRID_SPLIT).  Measured against 1 and 2 and against a choice by row count
    // (1500 x 1500 points: PA 40 -> 47 us but the passes 21 -> 16.5 us, because the stored candidates spread evenly over the
    // waves' regions; 200 x 200 lines: PA 48 -> 30 us; batched +6 % / +5 %)
    constexpr int32_t split_log = 2;
    static_assert((1 << split_log) == GRID_SPLIT, "");
    const int32_t n_tasks = n1 << split_log;
    auto seg_at = [&](uint32_t w) -> uint32_t {
        uint32_t before = 0;
        for (int32_t r = 0; r * NT < n_tasks; ++r) {
            const int32_t left = n_tasks - r * NT;                        // tasks of round r
            before += (uint32_t)(left < (int32_t)(64u * w) ? left : (int32_t)(64u * w));
        }
        return reg_off + free_words * before / (uint32_t)(n_tasks > 0 ? n_tasks : 1);   // < 2^16 words x <= 8192 tasks: 32 bits do
    };
    // (read behind P0's barrier; a listed problem that takes the column-bucketed bookkeeping never needs them -- the loop
    // and its divisions on 17 lanes would hold the other waves at that barrier --: it fills them in if it falls back)
    uint32_t seg_first = 0;
    // candidate k of this wave: its LDS region first, then its share of the global store
    auto cand_load = [&](uint32_t k) -> uint32_t {
        return k < seg_words ? s_dyn[seg_first + k] : store[(size_t)wv * tail_cap + (k - seg_words)];
    };
    auto cand_store = [&](uint32_t k, uint32_t v) {
        if (k < seg_words) s_dyn[seg_first + k] = v;
        else store[(size_t)wv * tail_cap + (k - seg_words)] = v;
    };

    // ---- P0: tables ----
    // A listed problem whose PC never counts rows without candidates (nnr <= 1) reads nothing of the grid here: the column-
    // bucketed bookkeeping below takes the LDS behind the row words, cell_start copy included.  Its list's length, the first
    // COLS_EARLY words per lane of the list and the grid's item count are requested NOW: one round trip under the table
    // initialisation instead of three behind it.
    constexpr uint32_t COLS_HOLD = 32, COLS_EARLY = 12;
    const bool count_empty = 2147483647.0 < 2147483647.0 * g.nnr;     // PC's nnr > 1 rule
    const bool cols_maybe = PLSLAM_GRID_COLUMNS && MODE == 2 && NT == 1024 && pre != nullptr && !count_empty;
    const uint32_t col_off = 2u * (uint32_t)(n2 + n1);
    // (k_grid_records' list: the words of item tid and item tid + NT, and the first two words per lane of the records that
    // did not fit their items' -- those are listed from the END of the store downwards, so their place is known now)
    constexpr uint32_t ITEMS_EARLY = 2;
    uint32_t c_early[COLS_EARLY], total_early = 0, items_end_early = 0;
    u32x4 it_early[ITEMS_EARLY][REC_SLOT / 4];
    uint32_t ov_early[ITEMS_EARLY];
    const bool slots_early = cols_maybe && pre_slots == REC_SLOT;
    if (cols_maybe) {
        total_early = ((PLSLAM_AS_GLOBAL const uint32_t*)pre)[0];
        items_end_early = (uint32_t)g_cell_start[ncell];
        PLSLAM_AS_GLOBAL const uint32_t* raw = store + (uint32_t)g.pair_cap;
        if (slots_early) {
#pragma unroll
            for (uint32_t j = 0; j < ITEMS_EARLY; ++j) {
                const uint32_t item = (uint32_t)tid + j * NT;
                const bool in = ((uint64_t)item + 1u) * REC_SLOT <= (uint64_t)(uint32_t)g.pair_cap;
#pragma unroll
                for (uint32_t v = 0; v < REC_SLOT / 4; ++v) {
                    const u32x4 none4 = {KEY_NONE, KEY_NONE, KEY_NONE, KEY_NONE};
                    it_early[j][v] = in ? *(PLSLAM_AS_GLOBAL const u32x4*)(raw + (size_t)item * REC_SLOT + 4u * v) : none4;
                }
                ov_early[j] = item < (uint32_t)g.pair_cap ? raw[(uint32_t)g.pair_cap - 1u - item] : KEY_NONE;
            }
        } else {
#pragma unroll
            for (uint32_t j = 0; j < COLS_EARLY; ++j) {
                const uint32_t k = (uint32_t)tid + j * NT;
                c_early[j] = k < (uint32_t)g.pair_cap ? raw[k] : KEY_NONE;      // (words behind the list's end: masked later)
            }
        }
    }
    if (LDS && !cols_maybe) {
        for (int32_t j = tid; j <= ncell; j += NT) s_dyn[2 * (n2 + n1) + j] = (uint32_t)g_cell_start[j];
    }
    if (MODE == 2 && !pre) {
        PLSLAM_AS_LDS int32_t* li = (PLSLAM_AS_LDS int32_t*)(s_dyn + items_off);
        for (int32_t j = tid; j < g.n_items; j += NT) li[j] = g_items[j];
        PLSLAM_AS_LDS u32x4* lt = (PLSLAM_AS_LDS u32x4*)(s_dyn + d2_off);
        for (int32_t j = tid; j < 2 * n2; j += NT) lt[j] = g_d2[j];
        if (has_dirs) {     // a candidate's direction test sits between its item and its descriptor: not a global round trip
            PLSLAM_AS_LDS double* ld = (PLSLAM_AS_LDS double*)(s_dyn + d2_off + 8u * (uint32_t)n2);
            PLSLAM_AS_GLOBAL const double* gd = (PLSLAM_AS_GLOBAL const double*)g.dir2;
            for (int32_t j = tid; j < 2 * n2; j += NT) ld[j] = gd[j];
        }
    }
    for (int32_t j = tid; j < n2; j += NT) {
        P.state[j] = KEY_NONE;
        P.next[j] = KEY_NONE;
        if (flat && !cols_maybe) colbest[j] = KEY_NONE;
    }
    if (cols_maybe)
        for (int32_t j = tid; j <= n2; j += NT) s_dyn[col_off + j] = 0u;          // the columns' counts
    if (tid < (int)NW) s_cur[tid] = 0u;
    for (int32_t i = tid; i < n1; i += NT) {
        P.row_k1[i] = KEY_NONE;
        P.row_k2[i] = KEY_NONE;
    }
    if (!cols_maybe && tid <= (int)NW) s_seg[tid] = seg_at((uint32_t)tid);
    __syncthreads();
    if (!cols_maybe) {
        seg_first = s_seg[wv];
        seg_words = s_seg[wv + 1] - seg_first;              // this wave's region
    }
    if ((cols_maybe ? items_end_early : (uint32_t)P.cs[ncell]) > (uint32_t)g.n_items) {      // the grid holds more items than the caller declared
        for (int32_t i = tid; i < n1; i += NT) g_matches[i] = -1;
        if (tid == 0) {
            if (g.n_matches) *g_(g.n_matches) = -1;
            if (g.status) (void)atomic_add_global(g.status, 1);
        }
        return;
    }
    GRID_STAMP();

    // ---- PA: distances ----
    uint32_t store_words = 0;        // slots claimed so far (uniform)
    uint32_t has_items = 0;          // bit r: this lane's row of round r has grid items inside its windows (mutual only)
    bool cols_done = false;          // (uniform) the column-bucketed path has produced the records and the rows' best two
    bool cols_fast = false;          // (uniform) ... straight from k_grid_records' list: a column's state is d << fb1 | row
    if (flat) {
        if (count_empty)                                                  // PC's nnr > 1 rule needs to know (cell_start is gone by then)
            for (int32_t r = 0; r < n_rounds && r < 32; ++r)
                if (r * NT + tid < n1 && count_items(g, P, r * NT + tid) > 0u) has_items |= 1u << r;
        if (pre) {
            // the candidate words of k_grid_candidates: bookkeeping in two candidate-parallel sweeps.  First every candidate
            // proposes itself as its column's first record and shows its (d, row) to colbest; then, colbest being final, a
            // candidate that an earlier row matches or beats is dropped and the others go to the waves' regions, evenly.
            // (the grid's item count has been checked against the caller's n_items, at most 2^31, by now)
            const uint32_t n_slots = pre_slots * (cols_maybe ? items_end_early : pre_slots ? (uint32_t)P.cs[ncell] : 0u);
            const uint64_t total64 = (uint64_t)n_slots + (cols_maybe ? total_early : *(PLSLAM_AS_GLOBAL const uint32_t*)pre);
            const uint32_t total = total64 > (uint64_t)(uint32_t)g.pair_cap ? (uint32_t)g.pair_cap + 1u : (uint32_t)total64;
            PLSLAM_AS_GLOBAL const uint32_t* raw = store + (uint32_t)g.pair_cap;
            // word k of the list: the items' words, then (from the end of the store downwards) what did not fit them
            auto list_at = [&](uint32_t k) -> uint32_t { return k < n_slots || !pre_slots ? raw[k] : raw[(uint32_t)g.pair_cap - 1u - (k - n_slots)]; };
#ifdef PLSLAM_GRID_TIMING
            dbg_total = total;
#endif
#ifdef PLSLAM_GRID_DEBUG_LIST
            if (tid == 0 && n2 <= 4) {
                printf("[list] n1 %d n2 %d slots %u total %u pair_cap %d items_end %u overflow %u:", n1, n2, n_slots, total, g.pair_cap, items_end_early, total_early);
                for (uint32_t k = 0; k < total && k < 40; ++k) printf(" %08x", list_at(k));
                printf("\n");
            }
#endif
            if (total > (uint32_t)g.pair_cap) {                         // (uniform) the list did not fit: report, match nothing
                for (int32_t i = tid; i < n1; i += NT) g_matches[i] = -1;
                if (tid == 0) {
                    if (g.n_matches) *g_(g.n_matches) = -1;
                    if (g.status) (void)atomic_add_global(g.status, 1);
                }
                return;
            }
            const uint32_t mk1 = (1u << fb1) - 1u, mk2 = (1u << fb2) - 1u;
            // ---- the list fits LDS whole (a keyframe pair's ~28 k candidates do; the ~5 k records k_grid_records leaves of them
            // easily): bucket it by COLUMN and read each column's records off its own segment, no passes over the whole list.
            // A column's records are, by the definition the passes implement, r_1 = min (i1 << 9 | d) over its candidates,
            // r_{k+1} = that min over the candidates with d below r_k's; its live candidates are exactly its records (they join
            // their rows' best two), its final state is the last one.  The counting sort: per-column counts by LDS atomics, an
            // exclusive scan, a second round of atomics for the places; the list itself is read ONCE, into registers
            // (COLS_HOLD words per lane).  The segments lie over the cell_start copy's place and everything behind it.
            // ---- the shortest way: the list is k_grid_records', and no column has two runs (every item of the grid in one
            // cell: points).  The list then holds each column's records and nothing else -- no liveness to settle: every word
            // joins its row's best two, a column's state is its smallest (d, row).  A lane takes an item's words (records first:
            // the first is one iff the run has any -- the counts P0 zeroed take those, and a column counted twice has two runs:
            // the tables are wiped and the bucketed bookkeeping below runs).  The returning atomics of a lane go out together.
            const uint32_t n_over = total - n_slots;                    // (total <= pair_cap here)
            if (PLSLAM_GRID_FAST && slots_early && items_end_early <= ITEMS_EARLY * NT && n_over <= ITEMS_EARLY * NT) {
                PLSLAM_AS_LDS uint32_t* off = s_dyn + col_off;
                bool dup = false;
                auto fold = [&](uint32_t w, bool first, uint32_t& was_) {           // a record word: column state, row's best
                    const uint32_t i2 = w & mk2, i1 = (w >> fb2) & mk1, d = w >> (fb1 + fb2);
                    if (first) dup = dup || atomicAdd((uint32_t*)&off[i2], 1u) != 0u;
                    atomicMin((uint32_t*)&P.state[i2], (d << fb1) | i1);
                    was_ = atomicMin((uint32_t*)&P.row_k1[i1], (d << KEY_IDX_BITS) | i2);
                };
                auto fold2 = [&](uint32_t w, uint32_t was_) {                       // ... whichever lost goes to the second best
                    const uint32_t i2 = w & mk2, i1 = (w >> fb2) & mk1, d = w >> (fb1 + fb2), key = (d << KEY_IDX_BITS) | i2;
                    if (was_ != key) atomicMin((uint32_t*)&P.row_k2[i1], was_ > key ? was_ : key);
                };
#pragma unroll
                for (uint32_t j = 0; j < ITEMS_EARLY; ++j) {
                    if (j * NT >= items_end_early) break;
                    uint32_t w[REC_SLOT], was[REC_SLOT];
                    const bool mine = (uint32_t)tid + j * NT < items_end_early;
#pragma unroll
                    for (uint32_t v = 0; v < REC_SLOT; ++v) w[v] = mine ? it_early[j][v / 4][v % 4] : KEY_NONE;
#pragma unroll
                    for (uint32_t v = 0; v < REC_SLOT; ++v) {
                        if (!__any(w[v] != KEY_NONE)) break;                          // (records first: no lane has a later one either)
                        was[v] = 0u;
                        if (w[v] != KEY_NONE) fold(w[v], v == 0u, was[v]);
                    }
#pragma unroll
                    for (uint32_t v = 0; v < REC_SLOT; ++v) {
                        if (!__any(w[v] != KEY_NONE)) break;
                        if (w[v] != KEY_NONE) fold2(w[v], was[v]);
                    }
                }
#pragma unroll
                for (uint32_t j = 0; j < ITEMS_EARLY; ++j) {
                    if (j * NT >= n_over) break;
                    if ((uint32_t)tid + j * NT < n_over) {
                        uint32_t was_ = 0u;
                        fold(ov_early[j], false, was_);
                        fold2(ov_early[j], was_);
                    }
                }
                if (__syncthreads_or(dup)) {
                    for (int32_t j = tid; j < n2; j += NT) P.state[j] = KEY_NONE;
                    for (int32_t j = tid; j <= n2; j += NT) off[j] = 0u;
                    for (int32_t i = tid; i < n1; i += NT) {
                        P.row_k1[i] = KEY_NONE;
                        P.row_k2[i] = KEY_NONE;
                    }
                    __syncthreads();
                } else {
                    cols_done = true;
                    cols_fast = true;
                }
            }
            // (a lane per column: worth it while the columns are many and short -- 200 columns of 100 candidates each, a map's
            // lines against a keyframe's, took 107 us this way against ~30 us of record passes)
            // (k_grid_records' list holds records only: a few per run whatever the windows)
            if (!cols_done && cols_maybe && total <= COLS_HOLD * NT &&
                (pre_slots > 0u ? (uint64_t)items_end_early <= (uint64_t)PLSLAM_GRID_RUNS_PER_COLUMN * (uint32_t)n2 : (uint64_t)total <= 24ull * (uint32_t)n2) &&
                (uint64_t)col_off + (uint32_t)n2 + 1u + total <= (uint64_t)lds_words) {
                PLSLAM_AS_LDS uint32_t* off = s_dyn + col_off;              // n2 + 1: counts (zeroed by P0), then the segments' first words
                PLSLAM_AS_LDS uint32_t* seg = off + n2 + 1;
                uint32_t c[COLS_HOLD];
#ifdef PLSLAM_GRID_TIMING
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                const uint32_t nj = (total + NT - 1) / NT;                  // (uniform) words per lane that exist at all: the unrolled
                COLS_STAMP();                                              // steps behind them are skipped, not predicated away
#pragma unroll
                for (uint32_t j = 0; j < COLS_HOLD; ++j) c[j] = KEY_NONE;
                if (!slots_early) {
#pragma unroll
                    for (uint32_t j = 0; j < COLS_EARLY; ++j) c[j] = (uint32_t)tid + j * NT < total ? c_early[j] : KEY_NONE;
                }
                if (slots_early || total > COLS_EARLY * NT) {
#pragma unroll
                    for (uint32_t j = 0; j < COLS_HOLD; ++j) {
                        if (j * NT >= total) break;
                        const uint32_t k = (uint32_t)tid + j * NT;
                        if (slots_early || j >= COLS_EARLY) c[j] = k < total ? list_at(k) : KEY_NONE;
                    }
                }
                COLS_STAMP();
#pragma unroll
                for (uint32_t j = 0; j < COLS_HOLD; ++j)
                    if (j < nj && c[j] != KEY_NONE) atomicAdd((uint32_t*)&off[c[j] & mk2], 1u);
#ifdef PLSLAM_GRID_TIMING
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                t_cnt = wall_clock64();
#endif
                __syncthreads();
                COLS_STAMP();
                {   // exclusive scan of the counts: a run of columns per lane, wave scans, the waves' totals through s_part
                    const int32_t per = (n2 + NT - 1) / NT, b = tid * per, e = b + per < n2 ? b + per : n2;
                    uint32_t sum = 0;
                    for (int32_t j = b; j < e; ++j) sum += off[j];
                    uint32_t incl = sum;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
                        if (lane >= o) incl += t;
                    }
                    if (lane == 63) s_part[wv] = incl;
                    __syncthreads();
                    uint32_t run = incl - sum;
                    for (uint32_t w = 0; w < wv; ++w) run += s_part[w];
                    for (int32_t j = b; j < e; ++j) {
                        const uint32_t k = off[j];
                        off[j] = run;
                        P.next[j] = run;                                  // the column's cursor
                        run += k;
                    }
                    if (tid == NT - 1) off[n2] = run;                     // every word that is not KEY_NONE (lanes behind the last column: all of them)
                    __syncthreads();
                }
                COLS_STAMP();
#pragma unroll
                for (uint32_t j = 0; j < COLS_HOLD; ++j)
                    if (j < nj && c[j] != KEY_NONE) {
                        const uint32_t i2 = c[j] & mk2, i1 = (c[j] >> fb2) & mk1, d = c[j] >> (fb1 + fb2);
                        seg[atomicAdd((uint32_t*)&P.next[i2], 1u)] = (i1 << REC_D_BITS) | d;
                    }
                __syncthreads();
                COLS_STAMP();
                // A lane per column; the segment goes to registers once (REG_N words, chunks of 8 that no lane of the wave
                // needs are skipped).
                constexpr int REG_N = 32, CHUNK = 8;
                for (int32_t base2 = 0; base2 < n2; base2 += NT) {
                    const int32_t i2 = base2 + tid;
                    const bool act = i2 < n2;
                    const uint32_t b = act ? off[i2] : 0u, e = act ? off[i2 + 1] : 0u, len = e - b;
                    uint32_t longest = len;                            // (uniform) the wave's longest segment
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        const uint32_t t = (uint32_t)__shfl_xor((int)longest, o);
                        longest = t > longest ? t : longest;
                    }
                    longest = (uint32_t)__builtin_amdgcn_readfirstlane((int)longest);
#define need_chunk(q) (longest > (uint32_t)((q) * CHUNK))
                    uint32_t w[REG_N];
#pragma unroll
                    for (int q = 0; q < REG_N / CHUNK; ++q) {
                        if (need_chunk(q)) {
#pragma unroll
                            for (int j = q * CHUNK; j < (q + 1) * CHUNK; ++j) {
                                const uint32_t v = seg[b + ((uint32_t)j < len ? (uint32_t)j : 0u)];   // (len 0: any word, unused)
                                w[j] = (uint32_t)j < len ? v : KEY_NONE;
                            }
                        } else {
#pragma unroll
                            for (int j = q * CHUNK; j < (q + 1) * CHUNK; ++j) w[j] = KEY_NONE;
                        }
                    }
                    uint32_t last = KEY_NONE;
                    if (!need_chunk(1)) {
                        // (uniform) segments of at most 8 words -- what k_grid_records leaves: sort them (w = row << 9 | d orders
                        // by row; 19 compare-exchanges, a min and a max each), then a word is a record iff its distance is below
                        // every earlier word's.  No sweeps, and the records' atomics go out together: two LDS round trips.
#define GRID_CE(a, b) do { const uint32_t lo_ = w[a] < w[b] ? w[a] : w[b], hi_ = w[a] < w[b] ? w[b] : w[a]; w[a] = lo_; w[b] = hi_; } while (0)
                        GRID_CE(0, 1); GRID_CE(2, 3); GRID_CE(4, 5); GRID_CE(6, 7);
                        GRID_CE(0, 2); GRID_CE(1, 3); GRID_CE(4, 6); GRID_CE(5, 7);
                        GRID_CE(1, 2); GRID_CE(5, 6);
                        GRID_CE(0, 4); GRID_CE(1, 5); GRID_CE(2, 6); GRID_CE(3, 7);
                        GRID_CE(2, 4); GRID_CE(3, 5);
                        GRID_CE(1, 2); GRID_CE(3, 4); GRID_CE(5, 6);
#undef GRID_CE
                        uint32_t run = REC_D_MASK + 1u, was[CHUNK], dd[CHUNK];
                        bool live[CHUNK];
#pragma unroll
                        for (int j = 0; j < CHUNK; ++j) {
                            const uint32_t dj = w[j] == KEY_NONE ? REC_D_MASK + 1u : w[j] & REC_D_MASK;
                            live[j] = dj < run;
                            if (live[j]) last = w[j];                                         // (the last record: the last live word)
                            run = dj < run ? dj : run;
                            dd[j] = dj;
                        }
#pragma unroll
                        for (int j = 0; j < CHUNK; ++j) {
                            const uint32_t key = (dd[j] << KEY_IDX_BITS) | (uint32_t)i2;
                            was[j] = key;
                            if (live[j]) was[j] = atomicMin((uint32_t*)&P.row_k1[w[j] >> REC_D_BITS], key);
                        }
#pragma unroll
                        for (int j = 0; j < CHUNK; ++j) {
                            const uint32_t key = (dd[j] << KEY_IDX_BITS) | (uint32_t)i2;
                            if (live[j] && was[j] != key) atomicMin((uint32_t*)&P.row_k2[w[j] >> REC_D_BITS], was[j] > key ? was[j] : key);
                        }
                    } else {
                        // a sweep is a mask, a compare, a select and a min per word; lanes whose column is done idle until the
                        // wave's last column is (a column of ~19 candidates has ~3.5 records, the worst of 64 about 8)
                        uint32_t cur_d = len ? REC_D_MASK + 1u : 0u;
                        while (__any(cur_d != 0u)) {
                            uint32_t best = KEY_NONE;
#pragma unroll
                            for (int q = 0; q < REG_N / CHUNK; ++q) {
                                if (need_chunk(q)) {
#pragma unroll
                                    for (int j = q * CHUNK; j < (q + 1) * CHUNK; ++j) {
                                        const uint32_t t = (w[j] & REC_D_MASK) < cur_d && w[j] != KEY_NONE ? w[j] : KEY_NONE;
                                        best = t < best ? t : best;
                                    }
                                }
                            }
                            if (len > (uint32_t)REG_N && cur_d)                               // (long columns) the rest from LDS, 8 reads in flight
                                for (uint32_t k = b + REG_N; k < e; k += CHUNK) {
                                    uint32_t v[CHUNK];
#pragma unroll
                                    for (int j = 0; j < CHUNK; ++j) v[j] = seg[k + j < e ? k + j : b];        // (a repeat changes no minimum)
#pragma unroll
                                    for (int j = 0; j < CHUNK; ++j)
                                        if ((v[j] & REC_D_MASK) < cur_d && v[j] < best) best = v[j];
                                }
                            if (best != KEY_NONE) {
                                cur_d = best & REC_D_MASK;
                                last = best;
                                const uint32_t key = (cur_d << KEY_IDX_BITS) | (uint32_t)i2, i1 = best >> REC_D_BITS;
                                const uint32_t was = atomicMin((uint32_t*)&P.row_k1[i1], key);
                                if (was != key) atomicMin((uint32_t*)&P.row_k2[i1], was > key ? was : key);
                            } else
                                cur_d = 0u;
                        }
                    }
                    if (act) P.state[i2] = last;
                }
#undef need_chunk
                cols_done = true;
                __syncthreads();
                COLS_STAMP();
            } else if (cols_maybe && !cols_done) {                        // (uniform) the passes after all: what P0 left out
                __syncthreads();
                for (int32_t j = tid; j < n2; j += NT) colbest[j] = KEY_NONE;
                if (tid <= (int)NW) s_seg[tid] = seg_at((uint32_t)tid);
                __syncthreads();
                seg_first = s_seg[wv];
                seg_words = s_seg[wv + 1] - seg_first;
            }
            // (the list comes from L2: PRE_UN independent loads in flight per lane -- one load per loop trip made the two sweeps
            // 18 us of dependent round trips)
            constexpr int PRE_UN = 8;
            for (uint32_t k0 = (uint32_t)tid; k0 < (cols_done ? 0u : total); k0 += NT * PRE_UN) {
                uint32_t c[PRE_UN];
#pragma unroll
                for (int j = 0; j < PRE_UN; ++j) c[j] = k0 + (uint32_t)j * NT < total ? list_at(k0 + (uint32_t)j * NT) : KEY_NONE;
#pragma unroll
                for (int j = 0; j < PRE_UN; ++j)
                    if (c[j] != KEY_NONE) {
                        const uint32_t i2 = c[j] & mk2, i1 = (c[j] >> fb2) & mk1, d = c[j] >> (fb1 + fb2);
                        atomicMin((uint32_t*)&P.next[i2], (i1 << REC_D_BITS) | d);
                        atomicMin((uint32_t*)&colbest[i2], (d << fb1) | i1);
                    }
            }
            __syncthreads();
            const uint32_t lo = cols_done ? 0u : (uint32_t)((uint64_t)total * wv / NW);
            const uint32_t hi = cols_done ? 0u : (uint32_t)((uint64_t)total * (wv + 1) / NW);
            const uint64_t below_ = (1ull << lane) - 1ull;
            uint32_t out = 0;
            for (uint32_t base = lo; base < hi; base += 64 * PRE_UN) {
                uint32_t c[PRE_UN];
#pragma unroll
                for (int j = 0; j < PRE_UN; ++j) {
                    const uint32_t k = base + 64u * (uint32_t)j + (uint32_t)lane;
                    c[j] = k < hi ? list_at(k) : KEY_NONE;
                }
#pragma unroll
                for (int j = 0; j < PRE_UN; ++j) {
                    bool keep_ = false;
                    if (c[j] != KEY_NONE) {
                        const uint32_t i2 = c[j] & mk2, i1 = (c[j] >> fb2) & mk1, d = c[j] >> (fb1 + fb2), cb = colbest[i2];
                        keep_ = !((cb >> fb1) <= d && (cb & mk1) < i1);
                    }
                    const uint64_t m = __ballot(keep_);
                    if (keep_) {
                        const uint32_t pos = out + (uint32_t)__popcll(m & below_);
                        if (pos < seg_words + tail_cap) cand_store(pos, c[j]);
                    }
                    out += (uint32_t)__popcll(m);
                }
            }
            if (lane == 0) s_cur[wv] = out;
        }
        // a lane per (row, part of the row's window columns): rows differ a lot in their number of candidates, quarters of
        // rows much less, and there are four times as many of them to even out the lanes of a wave
        for (int32_t task = tid; task < (pre ? 0 : n_tasks); task += NT) {
            const int32_t i1 = task >> split_log;
            const u32x4 qa = g_d1[2 * (int64_t)i1], qb = g_d1[2 * (int64_t)i1 + 1];
            for_candidates(g, P, i1, [&](const int32_t (&i2)[CB]) {
                u32x4 ta[CB], tb[CB];
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    const int64_t t = i2[j] < 0 ? 0 : i2[j];
                    ta[j] = P.d2[2 * t];
                    tb[j] = P.d2[2 * t + 1];
                }
                uint32_t d[CB], was[CB];
#pragma unroll
                for (int j = 0; j < CB; ++j)
                    d[j] = (uint32_t)(__popc(qa.x ^ ta[j].x) + __popc(qa.y ^ ta[j].y) + __popc(qa.z ^ ta[j].z) +
                                      __popc(qa.w ^ ta[j].w) + __popc(qb.x ^ tb[j].x) + __popc(qb.y ^ tb[j].y) +
                                      __popc(qb.z ^ tb[j].z) + __popc(qb.w ^ tb[j].w));
#pragma unroll
                for (int j = 0; j < CB; ++j) {                        // the batch's atomics back to back, one wait
                    was[j] = 0u;                                      // (row 0 at distance 0 beats everything: "dead")
                    if (i2[j] >= 0) {
                        // the first record pass, fused: the smallest row of a column is its first record
                        atomicMin((uint32_t*)&P.next[i2[j]], ((uint32_t)i1 << REC_D_BITS) | d[j]);
                        was[j] = atomicMin((uint32_t*)&colbest[i2[j]], (d[j] << fb1) | (uint32_t)i1);
                    }
                }
                bool keep[CB];
                uint32_t n_keep = 0;
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    // stored unless an earlier row is known to match or beat it (or the slot is empty)
                    keep[j] = i2[j] >= 0 && !((was[j] >> fb1) <= d[j] && (was[j] & ((1u << fb1) - 1u)) < (uint32_t)i1);
                    n_keep += keep[j] ? 1u : 0u;
                }
                if (n_keep) {
                    uint32_t pos = atomicAdd(&s_cur[wv], n_keep);     // one claim per batch
#pragma unroll
                    for (int j = 0; j < CB; ++j)
                        if (keep[j]) {
                            if (pos < seg_words + tail_cap) cand_store(pos, (d[j] << (fb1 + fb2)) | ((uint32_t)i1 << fb2) | (uint32_t)i2[j]);
                            ++pos;
                        }
                }
            }, task & ((1 << split_log) - 1), 1 << split_log);
        }
        __threadfence_block();
        if (__syncthreads_or(s_cur[wv] > seg_words + tail_cap)) {       // a wave's share of the store does not fit: report, match nothing
            for (int32_t i = tid; i < n1; i += NT) g_matches[i] = -1;
            if (tid == 0) {
                if (g.n_matches) *g_(g.n_matches) = -1;
                if (g.status) (void)atomic_add_global(g.status, 1);
            }
            return;
        }
    }
    for (int32_t r = 0; r < (flat ? 0 : n_rounds); ++r) {
        const int32_t i1 = r * NT + tid;
        uint32_t depth = 0;
        if (g.mutual) {              // slot depth of this round = the largest item count of one of its rows
            uint32_t c = i1 < n1 ? count_items(g, P, i1) : 0u;
            if (c && r < 32) has_items |= 1u << r;      // for PC: the cell_start copy may be gone by then
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const uint32_t o = (uint32_t)__shfl_xor((int)c, off);
                c = c > o ? c : o;
            }
            uint32_t& s_max = s_max2[r & 1];     // alternating slots: no barrier needed after the read below
            if (tid == 0) s_max = 0u;
            __syncthreads();
            if (lane == 0) atomicMax(&s_max, c);
            __syncthreads();
            depth = s_max;
            if (tid == 0) round_k[r] = depth;
            if ((uint64_t)store_words + (uint64_t)depth * NT > (uint64_t)(uint32_t)g.pair_cap) {
                // the store does not fit: report, match nothing
                for (int32_t i = tid; i < n1; i += NT) g_matches[i] = -1;
                if (tid == 0) {
                    if (g.n_matches) *g_(g.n_matches) = -1;
                    if (g.status) (void)atomic_add_global(g.status, 1);
                }
                return;
            }
        }
        if (i1 < n1) {
            const u32x4 qa = g_d1[2 * (int64_t)i1], qb = g_d1[2 * (int64_t)i1 + 1];
            uint32_t k1 = KEY_NONE, k2 = KEY_NONE, kout = 0;
            PLSLAM_AS_GLOBAL uint32_t* slot = store + store_words;
            for_candidates(g, P, i1, [&](const int32_t (&i2)[CB]) {
                u32x4 ta[CB], tb[CB];
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    const int64_t t = i2[j] < 0 ? 0 : i2[j];
                    ta[j] = P.d2[2 * t];
                    tb[j] = P.d2[2 * t + 1];
                }
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    const uint32_t d = (uint32_t)(__popc(qa.x ^ ta[j].x) + __popc(qa.y ^ ta[j].y) + __popc(qa.z ^ ta[j].z) +
                                                  __popc(qa.w ^ ta[j].w) + __popc(qb.x ^ tb[j].x) + __popc(qb.y ^ tb[j].y) +
                                                  __popc(qb.z ^ tb[j].z) + __popc(qb.w ^ tb[j].w));
                    const uint32_t key = (d << KEY_IDX_BITS) | (uint32_t)i2[j];
                    if (i2[j] >= 0) {
                        if (g.mutual) {
                            // the first record pass, fused: no column has a record yet, so every candidate proposes
                            atomicMin((uint32_t*)&P.next[i2[j]], ((uint32_t)i1 << REC_D_BITS) | d);
                            slot[slot_index<NT>(kout, tid)] = key;
                            ++kout;
                        } else
                            best2_fold(k1, k2, key);
                    }
                }
            });
            if (g.mutual)
                rcnt[i1] = kout;
            else {
                P.row_k1[i1] = k1;
                P.row_k2[i1] = k2;
            }
        }
        store_words += depth * NT;
    }
    __threadfence_block();
    __syncthreads();
    GRID_STAMP();

    // ---- PB: record passes ----
    if (g.mutual && !cols_done) {
        // one row's share of a pass: stream its remaining candidates from `slot` (stride apart), keep the survivors
        // in `out`; returns how many
        auto pass_row = [&](int32_t i1, auto slot, auto out, auto at, uint32_t cnt) -> uint32_t {
            uint32_t k1 = P.row_k1[i1], k2 = P.row_k2[i1], kout = 0;
            for (uint32_t k0 = 0; k0 < cnt; k0 += PB_BATCH) {         // PB_BATCH independent loads in flight
                uint32_t key[PB_BATCH], st[PB_BATCH];
#pragma unroll
                for (int j = 0; j < PB_BATCH; ++j) key[j] = k0 + j < cnt ? slot[at(k0 + j)] : KEY_NONE;
#pragma unroll
                for (int j = 0; j < PB_BATCH; ++j) st[j] = P.state[key[j] == KEY_NONE ? 0u : key[j] & KEY_IDX_MASK];
#pragma unroll
                for (int j = 0; j < PB_BATCH; ++j) {
                    const uint32_t i2 = key[j] & KEY_IDX_MASK, d = key[j] >> KEY_IDX_BITS;
                    const uint32_t me = ((uint32_t)i1 << REC_D_BITS) | d;
                    // installed by the previous sweep: live
                    if (key[j] != KEY_NONE && st[j] == me) best2_fold(k1, k2, key[j]);
                    const uint32_t cur = st[j] == KEY_NONE ? 512u : st[j] & REC_D_MASK;
                    if (key[j] != KEY_NONE && d < cur) {             // still below the newest record: propose, keep
                        atomicMin((uint32_t*)&P.next[i2], me);
                        out[at(kout)] = key[j];
                        ++kout;
                    }
                }
            }
            P.row_k1[i1] = k1;
            P.row_k2[i1] = k2;
            return kout;
        };
        auto sweep = [&]() -> int {       // install the proposals; workgroup-wide "anything new?"
            if (!LDS) __threadfence_block();
            __syncthreads();
            int any = 0;
            for (int32_t i2 = tid; i2 < n2; i2 += NT) {
                const uint32_t nx = P.next[i2];
                if (nx != KEY_NONE) {
                    P.state[i2] = nx;
                    P.next[i2] = KEY_NONE;
                    any = 1;
                }
            }
            if (!LDS) __threadfence_block();
#ifdef PLSLAM_GRID_TIMING
            ++npass;
#endif
            return __syncthreads_or(any);
        };

        int more = sweep();               // installs the proposals PA made: every column's first record
        if (flat) {
            // ---- candidate-parallel passes: each wave works on its own region and compacts the survivors of a pass in
            // place (ballot + prefix: no other wave touches the region).
            // A column's record carries its pass number in the top bits, newest = smallest:
            //     rec(t, i1, d) = (TMAX - t) << (9 + fb1) | i1 << 9 | d     (TMAX = all ones of the 23 - fb1 >= 9 bits above),
            // pass t reads the records of pass t-1 from one array and proposes with an atomic min into the other: a proposal
            // of pass t beats whatever pass t-2 left there, so nothing is cleared and nothing is installed -- one barrier per
            // pass.  (Every candidate that survives pass t-1 proposed in it: its column's word in the array pass t reads IS
            // a pass t-1 record.  A column has at most 257 records: the pass number fits.)  Live candidates join their
            // row's best two with two LDS atomic mins: k1 takes the key, whichever of (old k1, key) lost goes to k2 -- every
            // key of the row except the final minimum reaches k2 exactly once.
            const uint32_t t_shift = REC_D_BITS + fb1, t_max = (1u << (32u - t_shift)) - 1u;
            const uint32_t m1 = (1u << fb1) - 1u, m2 = (1u << fb2) - 1u;
            for (int32_t i2 = tid; i2 < n2; i2 += NT) {                // the first records: pass 0
                const uint32_t v = P.state[i2];
                if (v != KEY_NONE) P.state[i2] = (t_max << t_shift) | v;
            }
            __syncthreads();
#ifdef PLSLAM_GRID_TIMING
            t_move = wall_clock64();
#endif
            constexpr int UN = 4;                                      // chunks of 64 candidates in flight per lane
            const uint64_t below = (1ull << lane) - 1ull;
            // one pass over the `alive` candidates of wave w's region (s_tail when w == NW); returns the survivors
            auto run_pass = [&](uint32_t w, uint32_t alive, uint32_t t) -> uint32_t {
                auto rd = (t & 1) ? P.state : P.next;
                auto wr = (t & 1) ? P.next : P.state;
                const uint32_t tag_rd = (t_max - (t - 1)) << t_shift, tag_wr = (t_max - t) << t_shift;
                uint32_t out = 0;
                for (uint32_t base = 0; base < alive; base += 64 * UN) {
                    uint32_t c[UN], rec[UN];
                    bool keep[UN];
                    // (uniform) the whole step inside the LDS region: plain LDS traffic, all loads in flight together
                    const bool in_lds = w != NW && base + 64 * UN <= seg_words;
                    PLSLAM_AS_LDS uint32_t* reg = s_dyn + seg_first;        // (in_lds: w is this wave)
#pragma unroll
                    for (int j = 0; j < UN; ++j) {
                        const uint32_t k = base + 64 * j + lane;
                        c[j] = k >= alive ? KEY_NONE : in_lds ? reg[k] : w == NW ? s_tail[k] : cand_load(k);
                    }
#pragma unroll
                    for (int j = 0; j < UN; ++j) rec[j] = rd[c[j] == KEY_NONE ? 0u : c[j] & m2];
#pragma unroll
                    for (int j = 0; j < UN; ++j) {
                        const uint32_t i2 = c[j] & m2, i1 = (c[j] >> fb2) & m1, d = c[j] >> (fb1 + fb2);
                        const uint32_t me = (i1 << REC_D_BITS) | d;
                        keep[j] = false;
                        if (c[j] != KEY_NONE) {
                            if (rec[j] == (tag_rd | me)) {                    // the record of the last pass: live
                                const uint32_t key = (d << KEY_IDX_BITS) | i2;
                                const uint32_t was = atomicMin((uint32_t*)&P.row_k1[i1], key);
                                // (a duplicate of the key -- an item in two cells of the window -- changes nothing)
                                if (was != key) atomicMin((uint32_t*)&P.row_k2[i1], was > key ? was : key);
                            } else if (d < (rec[j] & REC_D_MASK)) {           // still below it: propose, stay
                                atomicMin((uint32_t*)&wr[i2], tag_wr | me);
                                keep[j] = true;
                            }
                        }
                    }
#pragma unroll
                    for (int j = 0; j < UN; ++j) {                           // every read of this step is done: in place
                        const uint64_t m = __ballot(keep[j]);
                        if (keep[j]) {
                            const uint32_t k = out + (uint32_t)__popcll(m & below);
                            if (in_lds) reg[k] = c[j];
                            else if (w == NW) s_tail[k] = c[j];
                            else cand_store(k, c[j]);
                        }
                        out += (uint32_t)__popcll(m);
                    }
                }
                return out;
            };
            uint32_t alive = s_cur[wv], t = 1;
            bool tail = false;
            while (more) {
                alive = run_pass(wv, alive, t);
                ++t;
#ifdef PLSLAM_GRID_TIMING
                if (npass < 16) tp[npass] = wall_clock64();
                ++npass;
#endif
                PLSLAM_AS_LDS uint32_t* left_of = (PLSLAM_AS_LDS uint32_t*)s_part + (t & 1u) * NW;   // alternating: one barrier per pass
                if (lane == 0) left_of[wv] = alive;
                if (alive > seg_words) __threadfence_block();                // survivors in the global share: wave 0 may gather them
                __syncthreads();
                uint32_t left = 0;
                for (uint32_t w = 0; w < NW; ++w) left += left_of[w];
                more = left != 0u;
                if (left != 0u && left <= GRID_TAIL) {                   // few survivors: no more workgroup barriers
                    tail = true;
                    break;
                }
            }
            if (tail) {
                // the last passes of a problem carry a handful of candidates each: wave 0 gathers them and finishes alone
                // (its LDS operations are ordered; the other waves wait at the barrier below)
                if (wv == 0) {
                    uint32_t n_tail = 0;
                    PLSLAM_AS_LDS const uint32_t* left_of = (PLSLAM_AS_LDS const uint32_t*)s_part + (t & 1u) * NW;
                    for (uint32_t w = 0; w < NW; ++w) {
                        const uint32_t cw = left_of[w], first = s_seg[w], words = s_seg[w + 1] - first;
                        for (uint32_t k = lane; k < cw; k += 64)
                            s_tail[n_tail + k] = k < words ? s_dyn[first + k] : store[(size_t)w * tail_cap + (k - words)];
                        n_tail += cw;
                    }
                    while (n_tail) {
                        n_tail = run_pass(NW, n_tail, t);
                        ++t;
#ifdef PLSLAM_GRID_TIMING
                        if (npass < 16) tp[npass] = wall_clock64();
                        ++npass;
#endif
                    }
                }
                __syncthreads();
            }
            for (int32_t i2 = tid; i2 < n2; i2 += NT) {                  // the newest record of either array, untagged
                const uint32_t a = P.state[i2], b = P.next[i2], v = a < b ? a : b;
                P.state[i2] = v == KEY_NONE ? KEY_NONE : v & ((1u << t_shift) - 1u);
            }
            __syncthreads();
        }
        if (!flat) {
            int flip = 0;
            while (more) {
                // survivors of a pass go to the other half of the store: reads and writes never alias, so a batch's
                // loads do not wait for the previous batch's stores
                PLSLAM_AS_GLOBAL const uint32_t* __restrict__ src = store + (flip ? (uint32_t)g.pair_cap : 0u);
                PLSLAM_AS_GLOBAL uint32_t* __restrict__ dst = store + (flip ? 0u : (uint32_t)g.pair_cap);
                flip ^= 1;
                uint32_t off = 0;
                for (int32_t r = 0; r < n_rounds; ++r) {
                    const int32_t i1 = r * NT + tid;
                    if (i1 < n1) {
                        const uint32_t cnt = rcnt[i1];
                        if (cnt)
                            rcnt[i1] = pass_row(i1, src + off, dst + off, [tid](uint32_t k) { return slot_index<NT>(k, tid); }, cnt);
                    }
                    off += round_k[r] * NT;
                }
                more = sweep();
            }
        }
    }
    GRID_STAMP();

    // ---- PC: ratio test, mutual check, count ----
    uint32_t cnt = 0;
    for (int32_t i1 = tid; i1 < n1; i1 += NT) {
        const uint32_t k1 = P.row_k1[i1], k2 = P.row_k2[i1];
        int32_t m = -1;
        if (k1 != KEY_NONE) {
            const double best_d = (double)(int32_t)(k1 >> KEY_IDX_BITS);
            const double best_d2 = k2 == KEY_NONE ? 2147483647.0 : (double)(int32_t)(k2 >> KEY_IDX_BITS);
            if (best_d < best_d2 * g.nnr) {
                const int32_t i2 = (int32_t)(k1 & KEY_IDX_MASK);
                if (!g.mutual || (cols_fast ? P.state[i2] & ((1u << fb1) - 1u) : P.state[i2] >> REC_D_BITS) == (uint32_t)i1) m = i2;
            }
        }
        else if (2147483647.0 < 2147483647.0 * g.nnr &&
                 ((g.mutual && i1 / NT < 32) ? ((has_items >> (i1 / NT)) & 1u) != 0u : count_items(g, P, i1) > 0u))
            cnt += 1;   // upstream, nnr > 1 only: a row whose candidates all fail keeps best_d = best_d2 = INT_MAX, passes
                        // `best_d < best_d2 * nnr`, gets matches_12 = best_idx = -1 and is COUNTED
        g_matches[i1] = m;
        cnt += m >= 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o);
    __syncthreads();                                         // (s_part may still be read as the scan's wave totals)
    if (lane == 0) s_part[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0 && g.n_matches) {
        uint32_t all = 0;
        for (int w = 0; w < NT / 64; ++w) all += s_part[w];
        *g_(g.n_matches) = (int32_t)all;
    }
    GRID_STAMP();
#ifdef PLSLAM_GRID_TIMING
    if (tid == 0 && (blockIdx.x == 0 || (blockIdx.x & 1023) == 600))
        printf("[k_match_grid n1=%d n2=%d] P0 %d PA %d PB %d (move %d, %d passes) PC %d (x10 ns)\n", n1, n2, (int)(ts[1] - ts[0]),
               (int)(ts[2] - ts[1]), (int)(ts[3] - ts[2]), t_move ? (int)(t_move - ts[2]) : -1, npass, (int)(ts[4] - ts[3]));
    if (tid == 0 && blockIdx.x == 0 && pre)
        printf("   columns path %d: %u candidates, lds %u words | load %d count %d scan %d place %d records %d (since PA began: %d)\n",
               (int)cols_done + (int)cols_fast, dbg_total, lds_words, (int)(tc[1] - tc[0]), (int)(tc[2] - tc[1]), (int)(tc[3] - tc[2]), (int)(tc[4] - tc[3]),
               (int)(tc[5] - tc[4]), (int)(tc[0] - ts[1]));
    if (tid == 0 && blockIdx.x == 0 && pre)
        printf("   count: atomics done %d after the loads' stamp; %d shader MHz\n", (int)(t_cnt - tc[1]), (int)((clock64() - c_start) / ((wall_clock64() - ts[0]) / 100)));
    if (tid == 0 && (blockIdx.x == 0 || (blockIdx.x & 1023) == 600) && t_move) {
        printf("   passes:");
        for (int i = 0; i < npass && i < 16; ++i) printf(" %d", (int)(tp[i] - (i ? tp[i - 1] : t_move)));
        printf(" | stored %u of wave 0, region %u words | %d shader MHz\n", s_cur[0], seg_words,
               (int)((clock64() - c_start) / ((wall_clock64() - ts[0]) / 100)));
    }
#endif
#undef GRID_STAMP
#undef COLS_STAMP
}


// PA of ONE LDS-resident mutual problem spread over the chip (k_match_grid with `pre` does the rest): a lane per (row, window
// column) -- `split` = the window's width in cells, at most GRID_SPLIT_MAX: a lane's chain of dependent reads is centre -> cell
// offsets -> items -> desc2 rows, once --, 256 tasks per workgroup, every table read from global memory (the grid and the desc2 rows are
// a few tens of KB: L2).  A single workgroup spends two thirds of its time here -- a few 10^4 distances behind scattered
// reads, with the lanes of a wave unevenly loaded -- while 255 CUs idle.  The candidate words (d << (b1 + b2) | i1 << b2 | i2,
// as in the flat mode) of a workgroup are collected in LDS and appended to the list in the SECOND half of the problem's
// candidate store (one global atomic per workgroup; aux[0] = the list's length, zero when the kernel starts); their order in the
// list is whatever the scheduling made it -- nothing downstream depends on it (every combination is a min of keys).
constexpr uint32_t GRID_CAND_BUF = 6144;            // candidate words a workgroup collects before they go out (24 KB)
__global__ __launch_bounds__(256) void k_grid_candidates(const GridDesc* __restrict__ probs, uint32_t* __restrict__ aux, int split)
{
    __shared__ uint32_t s_buf[GRID_CAND_BUF];
    __shared__ uint32_t s_n, s_base;
    const GridDesc g = probs[0];
    const int tid = (int)threadIdx.x;
    const int32_t n1 = g.n1, n2 = g.n2;
    const int32_t ncell = g.cols * g.rows;
    uint32_t fb2 = 1;
    while (fb2 < 22 && (1u << fb2) < (uint32_t)n2) ++fb2;
    const uint32_t fb1 = 23u - fb2 > 14u ? 14u : 23u - fb2;
    GridPtrs<0> P;
    P.cs = (PLSLAM_AS_GLOBAL const uint32_t*)g.cell_start;
    P.items = (PLSLAM_AS_GLOBAL const int32_t*)g.cell_items;
    P.d2 = (PLSLAM_AS_GLOBAL const u32x4*)g.d2;
    P.centres = (PLSLAM_AS_GLOBAL const int32_t*)g.centres;
    P.dir1 = (PLSLAM_AS_GLOBAL const double*)g.dir1;
    P.dir2 = (PLSLAM_AS_GLOBAL const double*)g.dir2;
    P.state = P.next = P.row_k1 = P.row_k2 = nullptr;
    // the scratch layout of k_match_grid<2, 1024>: slot counts n1 | round depths | candidate store 2 x pair_cap
    PLSLAM_AS_GLOBAL uint32_t* rcnt = (PLSLAM_AS_GLOBAL uint32_t*)g.scratch;
    PLSLAM_AS_GLOBAL uint32_t* raw = rcnt + n1 + (n1 + GRID_THREADS - 1) / GRID_THREADS + (uint32_t)g.pair_cap;
    PLSLAM_AS_GLOBAL const u32x4* g_d1 = (PLSLAM_AS_GLOBAL const u32x4*)g.d1;
    uint32_t* const counter = aux;
    if (!(g.mutual && (uint32_t)n2 <= (1u << fb2) && (uint32_t)n1 <= (1u << fb1))) return;   // (k_match_grid evaluates its own then)
    if (tid == 0) s_n = 0u;
    __syncthreads();
    const int64_t task = (int64_t)blockIdx.x * 256 + tid;
    if (task < (int64_t)n1 * split && (uint32_t)P.cs[ncell] <= (uint32_t)g.n_items) {      // (an inconsistent grid: k_match_grid reports it)
        const int32_t i1 = (int32_t)(task / split), part = (int32_t)(task - (int64_t)i1 * split);
        const u32x4 qa = g_d1[2 * (int64_t)i1], qb = g_d1[2 * (int64_t)i1 + 1];
        for_candidates(g, P, i1, [&](const int32_t (&i2)[CB]) {
            u32x4 ta[CB], tb[CB];
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                const int64_t t = i2[j] < 0 ? 0 : i2[j];
                ta[j] = P.d2[2 * t];
                tb[j] = P.d2[2 * t + 1];
            }
            uint32_t n_keep = 0;
#pragma unroll
            for (int j = 0; j < CB; ++j) n_keep += i2[j] >= 0 ? 1u : 0u;
            if (n_keep) {
                uint32_t pos = atomicAdd(&s_n, n_keep);               // one claim per batch
#pragma unroll
                for (int j = 0; j < CB; ++j)
                    if (i2[j] >= 0) {
                        const uint32_t d = (uint32_t)(__popc(qa.x ^ ta[j].x) + __popc(qa.y ^ ta[j].y) + __popc(qa.z ^ ta[j].z) +
                                                      __popc(qa.w ^ ta[j].w) + __popc(qb.x ^ tb[j].x) + __popc(qb.y ^ tb[j].y) +
                                                      __popc(qb.z ^ tb[j].z) + __popc(qb.w ^ tb[j].w));
                        const uint32_t word = (d << (fb1 + fb2)) | ((uint32_t)i1 << fb2) | (uint32_t)i2[j];
                        if (pos < GRID_CAND_BUF) s_buf[pos] = word;
                        else {                                        // (a very dense grid) straight to the list
                            const uint32_t gp = (uint32_t)atomic_add_global(counter, 1);
                            if (gp < (uint32_t)g.pair_cap) raw[gp] = word;
                        }
                        ++pos;
                    }
            }
        }, part, split);
    }
    __syncthreads();
    const uint32_t n = s_n < GRID_CAND_BUF ? s_n : GRID_CAND_BUF;
    if (tid == 0) s_base = n ? (uint32_t)atomic_add_global(counter, (int)n) : 0u;
    __syncthreads();
    for (uint32_t k = (uint32_t)tid; k < n; k += 256u)
        if (s_base + k < (uint32_t)g.pair_cap) raw[s_base + k] = s_buf[k];
}


// The same list, pre-filtered, COLUMN-wise: a workgroup per REC_G vertically adjacent grid cells.  The rows whose windows
// touch the group come out of one sweep over every row's window centres (a few KB from L2; each wave sweeps a quarter of the
// rows and compacts its finds IN ROW ORDER, each with the mask of the group's cells its windows hold); a wave then takes a
// column (an item of one of the cells), evaluates its distance to those rows -- lane j the j-th row -- and a prefix minimum
// across the lanes says which of them are the column's records (d below every earlier row's): those words alone are kept.  A
// column of ~19 candidates has ~3 records, so what k_match_grid bookkeeps shrinks from ~28 k to ~4 k words for a keyframe
// pair, and no lane walks the dependent chain centre -> cell offsets -> items -> desc2 rows of k_grid_candidates.
// What k_match_grid needs: every live candidate, and candidates only.  A column whose item sits in SEVERAL cells (line
// segments) gets the records of each cell's row set -- a superset of its records (a record of the union is a record of any
// subset that holds it), and the bookkeeping downstream drops the rest: a dead candidate has an earlier RECORD at or below its
// distance, and every record is kept.
// Where they go: item k of the grid's CSR list (one (cell, column) run) owns words k * REC_SLOT ... + REC_SLOT - 1 of the list,
// records first, KEY_NONE behind them -- no counter to claim, nothing returns to the wave (a round trip of a global atomic is
// ~1 us here, and every workgroup of the launch wanted the same word).  The records a run has beyond REC_SLOT (a column in a
// hundred) are listed from the END of the store downwards, aux[0] counting them (their place does not depend on the grid).  The FIRST word of an item's slots is a record
// exactly when the run has any: k_match_grid counts those per column to see whether a column has one run (the list then holds
// its records and nothing else) or several.
// The descriptor comes BY VALUE (kernel arguments): one dependent round trip less in front of everything.
constexpr int REC_NT = 256;
constexpr int REC_G = 8;                            // cells per workgroup: same grid column x, consecutive y
constexpr int REC_ROWS_MAX = 16384;                 // rows of a problem that takes this path (the row lists of a group: 48 KB of LDS)
constexpr int64_t REC_GROUPS_MAX = 1 << 16;         // beyond this k_grid_candidates lists the pairs
__global__ __launch_bounds__(REC_NT) void k_grid_records(const GridDesc g, uint32_t* __restrict__ aux, const int32_t* __restrict__ n1_dev)
{
    // n1_dev: where the row count lives when a kernel upstream decides it (g.n1 is then its upper bound, and still what the
    // scratch layout is counted by)
    constexpr int NW = REC_NT / 64, PER_WAVE = REC_ROWS_MAX / NW, SWEEP_UN = 16;
    static_assert(REC_G <= 8, "a row's cells fit an 8-bit mask");
    __shared__ uint16_t s_rows[NW][PER_WAVE];         // wave w's finds among rows [w * q, (w + 1) * q), ascending
    __shared__ uint8_t s_mask[NW][PER_WAVE];
    __shared__ int32_t s_cs[REC_G + 1];
    __shared__ uint32_t s_wn[NW];
#ifdef PLSLAM_GRID_TIMING
    uint64_t tr[8];
    int ntr = 0;
#define REC_STAMP() do { if (ntr < 8) tr[ntr++] = wall_clock64(); } while (0)
#else
#define REC_STAMP() do { } while (0)
#endif
    REC_STAMP();
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    int32_t n1 = g.n1;
    if (n1_dev) {
        const int32_t n1_now = *(PLSLAM_AS_GLOBAL const int32_t*)n1_dev;
        n1 = n1_now >= 0 && n1_now < n1 ? n1_now : n1;
    }
    const int32_t n2 = g.n2;
    const int32_t ncell = g.cols * g.rows;
    const uint32_t fb2 = n2 > 1 ? 32u - (uint32_t)__builtin_clz((uint32_t)n2 - 1u) : 1u;        // bits of a column number (at least 1)
    const uint32_t fb1 = 23u - fb2 > 14u ? 14u : 23u - fb2;
    if (!(g.mutual && fb2 <= 22u && (uint32_t)n1 <= (1u << fb1)) || n1 > REC_ROWS_MAX) return;   // (launcher: never)
    PLSLAM_AS_GLOBAL const int32_t* cs = (PLSLAM_AS_GLOBAL const int32_t*)g.cell_start;
    PLSLAM_AS_GLOBAL const int32_t* items = (PLSLAM_AS_GLOBAL const int32_t*)g.cell_items;
    PLSLAM_AS_GLOBAL const int32_t* centres = (PLSLAM_AS_GLOBAL const int32_t*)g.centres;
    PLSLAM_AS_GLOBAL const u32x4* g_d1 = (PLSLAM_AS_GLOBAL const u32x4*)g.d1;
    PLSLAM_AS_GLOBAL const u32x4* g_d2 = (PLSLAM_AS_GLOBAL const u32x4*)g.d2;
    PLSLAM_AS_GLOBAL const double* dir1 = (PLSLAM_AS_GLOBAL const double*)g.dir1;
    PLSLAM_AS_GLOBAL const double* dir2 = (PLSLAM_AS_GLOBAL const double*)g.dir2;
    const bool dirs = g.dir1 != nullptr && g.dir2 != nullptr;
    PLSLAM_AS_GLOBAL uint32_t* rcnt = (PLSLAM_AS_GLOBAL uint32_t*)g.scratch;       // (the layout k_grid_candidates writes)
    PLSLAM_AS_GLOBAL uint32_t* raw = rcnt + g.n1 + (g.n1 + GRID_THREADS - 1) / GRID_THREADS + (uint32_t)g.pair_cap;
    const int32_t gpc = (g.rows + REC_G - 1) / REC_G;                   // groups per grid column
    const int32_t cx = (int32_t)blockIdx.x / gpc, cy0 = ((int32_t)blockIdx.x - cx * gpc) * REC_G;
    if (cx >= g.cols) return;
    const int32_t ng = g.rows - cy0 < REC_G ? g.rows - cy0 : REC_G, cell0 = cx * g.rows + cy0;
    // the group's slice of the CSR list and this wave's first rows' centres: requested together
    const int32_t it_all = cs[ncell], it_begin = cs[cell0], it_end = cs[cell0 + ng];
    const int32_t my_cs = tid <= ng ? cs[cell0 + tid] : 0;
    const int32_t q = (n1 + NW - 1) / NW, r_begin = wv * q, r_end = r_begin + q < n1 ? r_begin + q : n1;
    const bool one_centre = g.n_centres == 1;
    int32_t cxy[SWEEP_UN][2];
    if (one_centre) {
#pragma unroll
        for (int k = 0; k < SWEEP_UN; ++k) {
            const int32_t r = r_begin + k * 64 + lane;
            cxy[k][0] = cxy[k][1] = 0;
            if (r < r_end) {
                cxy[k][0] = centres[2 * (int64_t)r];
                cxy[k][1] = centres[2 * (int64_t)r + 1];
            }
        }
    }
    if ((uint32_t)it_all > (uint32_t)g.n_items || it_begin >= it_end) return;    // (an inconsistent grid: k_match_grid reports it)
    if (tid <= ng) s_cs[tid] = my_cs;
    REC_STAMP();
    // this wave's first columns: their numbers now (under the sweep below), their descriptors together once those are here --
    // two round trips for IT_UN columns, not two each (a group holds ~3 items, a wave takes every fourth)
    constexpr int IT_UN = 4;
    const int32_t k_first = it_begin + wv;
    int32_t i2_un[IT_UN];
#pragma unroll
    for (int t = 0; t < IT_UN; ++t) i2_un[t] = k_first + t * NW < it_end ? items[k_first + t * NW] : -1;

    // ---- the rows whose windows touch the group, each with the mask of the cells it reaches ----
    // (cell (cx, cy) of the grid lies in a centre's clamped window [min, max) exactly when cx - x is in [-w0, w1] and cy - y in
    // [-w2, w3]: the clamps of window_of only cut what no cell index reaches)
    auto cells_of = [&](int64_t x, int64_t y) -> uint32_t {
        const int64_t dx = (int64_t)cx - x;
        int64_t lo = y - g.w[2], hi = y + g.w[3];
        lo = lo > cy0 ? lo : cy0;
        hi = hi < cy0 + ng - 1 ? hi : cy0 + ng - 1;
        if (dx >= -(int64_t)g.w[0] && dx <= (int64_t)g.w[1] && lo <= hi)
            return ((2u << (uint32_t)(hi - cy0)) - 1u) & ~((1u << (uint32_t)(lo - cy0)) - 1u);
        return 0u;
    };
    const uint64_t below = (1ull << lane) - 1ull;
    uint32_t found = 0;                               // (uniform) this wave's finds so far
    auto keep = [&](int32_t r, uint32_t mask) {
        const uint64_t b = __ballot(mask != 0u);
        if (mask) {
            const uint32_t pos = found + (uint32_t)__popcll(b & below);
            s_rows[wv][pos] = (uint16_t)r;
            s_mask[wv][pos] = (uint8_t)mask;
        }
        found += (uint32_t)__popcll(b);
    };
    if (one_centre) {
#pragma unroll
        for (int k = 0; k < SWEEP_UN; ++k) {
            if (r_begin + k * 64 >= r_end) break;
            const int32_t r = r_begin + k * 64 + lane;
            keep(r, r < r_end ? cells_of(cxy[k][0], cxy[k][1]) : 0u);
        }
    }
    for (int32_t r0 = r_begin + (one_centre ? SWEEP_UN * 64 : 0); r0 < r_end; r0 += 64) {      // (many rows, or several centres a row)
        const int32_t r = r0 + lane;
        uint32_t mask = 0;
        if (r < r_end)
            for (int32_t c = 0; c < g.n_centres; ++c) {
                PLSLAM_AS_GLOBAL const int32_t* p = centres + ((int64_t)r * g.n_centres + c) * 2;
                mask |= cells_of(p[0], p[1]);
            }
        keep(r, mask);
    }
    if (lane == 0) s_wn[wv] = found;
    REC_STAMP();
    __syncthreads();
    REC_STAMP();
    uint32_t first_of[NW + 1];                        // the waves' finds, concatenated: row j of the group
    first_of[0] = 0u;
#pragma unroll
    for (int w = 0; w < NW; ++w) first_of[w + 1] = first_of[w] + s_wn[w];
    const uint32_t n_rows = first_of[NW];
    auto row_at = [&](uint32_t j, uint32_t& row, uint32_t& mk) {
        uint32_t w = 0;
#pragma unroll
        for (int t = 1; t < NW; ++t) w += j >= first_of[t] ? 1u : 0u;
        uint32_t base = 0;
#pragma unroll
        for (int t = 1; t < NW; ++t) base = w == (uint32_t)t ? first_of[t] : base;
        row = j < n_rows ? s_rows[w][j - base] : 0u;
        mk = j < n_rows ? s_mask[w][j - base] : 0u;
    };

    // ---- a wave per column; lane j holds the j-th row (the first 64 rows' descriptors are loaded once) ----
    uint32_t row_0, mask_0;
    row_at((uint32_t)lane, row_0, mask_0);
    const u32x4 qa_0 = g_d1[2 * (int64_t)row_0], qb_0 = g_d1[2 * (int64_t)row_0 + 1];
    auto run_column = [&](int32_t k, int32_t i2, const u32x4& ta, const u32x4& tb, double b0, double b1) {
        PLSLAM_AS_GLOBAL uint32_t* slot = raw + (uint64_t)(uint32_t)k * REC_SLOT;
        const bool room = ((uint64_t)(uint32_t)k + 1u) * REC_SLOT <= (uint64_t)(uint32_t)g.pair_cap;    // (launcher: always)
        uint32_t n_rec = 0;                            // (uniform) records of this run so far
        if ((uint32_t)i2 < (uint32_t)n2) {
            uint32_t cq = 0;                           // the cell of item k: how many of the group's inner boundaries lie at or below k
            for (int32_t t = 1; t < ng; ++t) cq += k >= s_cs[t] ? 1u : 0u;
            const uint32_t bit = 1u << cq;
            uint32_t carry = REC_D_MASK + 1u;          // the smallest distance of the rows before this chunk
            for (uint32_t j0 = 0; j0 < n_rows && carry; j0 += 64) {
                uint32_t row = row_0, mk = mask_0;
                u32x4 qa = qa_0, qb = qb_0;
                if (j0) {
                    row_at(j0 + (uint32_t)lane, row, mk);
                    qa = g_d1[2 * (int64_t)row];
                    qb = g_d1[2 * (int64_t)row + 1];
                }
                bool valid = (mk & bit) != 0u;
                const uint32_t d = (uint32_t)(__popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                                              __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w));
                if (dirs) {
                    const double a0 = dir1[2 * (int64_t)row], a1 = dir1[2 * (int64_t)row + 1];
                    const double dot = a0 * b0 + a1 * b1;
                    if (fabs(dot) < g.sim_th) valid = false;     // NaN (zero-length direction) compares false: kept
                }
                const uint32_t dm = valid ? d : REC_D_MASK + 1u;
                uint32_t incl = dm;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
                    if (lane >= o) incl = t < incl ? t : incl;
                }
                uint32_t excl = (uint32_t)__shfl_up((int)incl, 1);
                if (lane == 0) excl = REC_D_MASK + 1u;
                excl = excl < carry ? excl : carry;
                const bool rec = valid && d < excl;
                const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
                carry = all < carry ? all : carry;
                const uint64_t m = __ballot(rec);
                if (rec) {
                    const uint32_t pos = n_rec + (uint32_t)__popcll(m & below);
                    const uint32_t word = (d << (fb1 + fb2)) | (row << fb2) | (uint32_t)i2;
                    if (pos < REC_SLOT) {
                        if (room) slot[pos] = word;
                    } else {                                      // beyond the run's own words: from the store's end downwards
                        const uint32_t gp = (uint32_t)atomic_add_global(aux, 1);
                        if ((uint64_t)(uint32_t)it_all * REC_SLOT + gp < (uint64_t)(uint32_t)g.pair_cap)
                            raw[(uint32_t)g.pair_cap - 1u - gp] = word;
                    }
                }
                n_rec += (uint32_t)__popcll(m);
            }
        }
        if (room && (uint32_t)lane < REC_SLOT && (uint32_t)lane >= n_rec) slot[lane] = KEY_NONE;
    };
    {
        u32x4 ta[IT_UN], tb[IT_UN];
        double b0[IT_UN], b1[IT_UN];
#pragma unroll
        for (int t = 0; t < IT_UN; ++t) {
            const int32_t i2 = __builtin_amdgcn_readfirstlane(i2_un[t]);
            const int64_t at = (uint32_t)i2 < (uint32_t)n2 ? i2 : 0;
            ta[t] = g_d2[2 * at];
            tb[t] = g_d2[2 * at + 1];
            b0[t] = dirs ? dir2[2 * at] : 0.0;
            b1[t] = dirs ? dir2[2 * at + 1] : 0.0;
        }
#pragma unroll
        for (int t = 0; t < IT_UN; ++t) {
            if (k_first + t * NW >= it_end) break;
            run_column(k_first + t * NW, __builtin_amdgcn_readfirstlane(i2_un[t]), ta[t], tb[t], b0[t], b1[t]);
        }
    }
    for (int32_t k = k_first + IT_UN * NW; k < it_end; k += NW) {       // (a dense group)
        const int32_t i2 = __builtin_amdgcn_readfirstlane(items[k]);
        const int64_t at = (uint32_t)i2 < (uint32_t)n2 ? i2 : 0;
        const u32x4 ta = g_d2[2 * at], tb = g_d2[2 * at + 1];
        run_column(k, i2, ta, tb, dirs ? dir2[2 * at] : 0.0, dirs ? dir2[2 * at + 1] : 0.0);
    }
    REC_STAMP();
#ifdef PLSLAM_GRID_TIMING
    if (tid == 0 && (blockIdx.x % 97) == 5)
        printf("[k_grid_records group %d: %d items, %u rows] start %llu | args+cs %d sweep %d barrier %d columns %d (x10 ns)\n",
               (int)blockIdx.x, it_end - it_begin, n_rows, (unsigned long long)(tr[0] % 100000000ull), (int)(tr[1] - tr[0]), (int)(tr[2] - tr[1]),
               (int)(tr[3] - tr[2]), (int)(tr[4] - tr[3]));
#endif
#undef REC_STAMP
}

}  // namespace

// words of the tables that live in LDS when they fit (cell_start copy + column / row words)
size_t grid_fixed_words(int32_t n1, int32_t n2, int64_t ncell)
{
    return (size_t)(ncell + 1) + 2 * (size_t)n2 + 2 * (size_t)n1;
}
bool grid_fits_lds(int32_t n1, int32_t n2, int64_t ncell)
{
    return grid_fixed_words(n1, n2, ncell) * 4 <= GRID_LDS_FIXED_MAX_BYTES;
}
// global scratch of one problem: [tables when they do not fit LDS |] slot counts | round depths | candidate store x 2
size_t grid_scratch_words(int32_t n1, int32_t n2, int64_t ncell, int32_t pair_cap)
{
    return (grid_fits_lds(n1, n2, ncell) ? 0 : 2 * (size_t)n2 + 2 * (size_t)n1) + (size_t)n1 +
           (size_t)((n1 + GRID_THREADS - 1) / GRID_THREADS) + 2 * (size_t)pair_cap;
}

// LDS bytes of a problem in each mode (MODE 2: tables, items at a 16-byte boundary, desc2 rows)
size_t grid_lds_bytes(int mode, int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs)
{
    if (mode == 0) return 0;
    size_t w = grid_fixed_words(n1, n2, ncell);
    if (mode == 2) {
        w = (w + 3) & ~size_t(3);
        w = (w + (size_t)n_items + 3) & ~size_t(3);
        w += 8 * (size_t)n2;
        if (dirs) w += 4 * (size_t)n2;          // the directions of the desc2 lines (2 doubles each)
        w += (size_t)n2;                        // flat mode: the best (d, row) seen per column while PA runs
    }
    return w * 4;
}
int grid_mode(int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs)
{
    if (grid_lds_bytes(2, n1, n2, ncell, n_items, dirs) <= GRID_LDS_MAX_BYTES) return 2;
    return grid_fits_lds(n1, n2, ncell) ? 1 : 0;
}

// capacity of the candidate store (host-side data): rows go in blocks of 1024, a block needs 1024 slots per grid item
// inside the windows of its fullest row (mutual only; without it nothing is stored).  (The 256-lane workgroups of small
// problems use blocks of 256: never more than this.)
int64_t grid_store_capacity_host(const int32_t* centres, int32_t n1, int32_t n_centres, const int32_t* cell_start,
                                 int32_t cols, int32_t rows, const int32_t window[4], int mutual)
{
    if (!mutual) return 0;
    int64_t total = 0, depth = 0;
    for (int32_t i1 = 0; i1 < n1; ++i1) {
        int64_t cnt = 0;
        for (int32_t c = 0; c < n_centres; ++c) {
            const int64_t k = (int64_t)i1 * n_centres + c;
            const int64_t x = centres[2 * k], y = centres[2 * k + 1];
            const int64_t min_x = x - window[0] > 0 ? x - window[0] : 0;
            const int64_t max_x = x + window[1] + 1 < cols ? x + window[1] + 1 : cols;
            const int64_t min_y = y - window[2] > 0 ? y - window[2] : 0;
            const int64_t max_y = y + window[3] + 1 < rows ? y + window[3] + 1 : rows;
            if (min_y >= max_y) continue;
            for (int64_t x_ = min_x; x_ < max_x; ++x_) cnt += cell_start[x_ * rows + max_y] - cell_start[x_ * rows + min_y];
        }
        if (cnt > depth) depth = cnt;
        if ((i1 & (GRID_THREADS - 1)) == GRID_THREADS - 1 || i1 == n1 - 1) {
            total += depth * GRID_THREADS;
            depth = 0;
        }
    }
    return total;
}

// upper bound of grid_store_capacity_host() from the grid alone: fullest cell x cells of a window, at most every item, per
// window centre; rows in blocks of 1024
int64_t grid_store_capacity_bound(int32_t n1, int32_t n_centres, const int32_t* cell_start, int32_t cols, int32_t rows,
                                  const int32_t window[4], int mutual)
{
    if (!mutual || n1 <= 0) return 0;
    const int64_t ncell = (int64_t)cols * rows;
    int64_t fullest = 0;
    for (int64_t c = 0; c < ncell; ++c) fullest = std::max<int64_t>(fullest, (int64_t)cell_start[c + 1] - cell_start[c]);
    const int64_t wx = std::min<int64_t>((int64_t)window[0] + window[1] + 1, cols);
    const int64_t wy = std::min<int64_t>((int64_t)window[2] + window[3] + 1, rows);
    const int64_t per_row = std::min<int64_t>(fullest * wx * wy, cell_start[ncell]) * n_centres;
    return per_row * GRID_THREADS * ((n1 + GRID_THREADS - 1) / GRID_THREADS);
}

constexpr int GRID_SPLIT_MAX = 16;         // ... with at most this many lanes per row (one per window column)
#ifndef PLSLAM_GRID_SPLIT_MIN_ROWS
#define PLSLAM_GRID_SPLIT_MIN_ROWS 128
#endif
constexpr int GRID_SPLIT_MIN_ROWS = PLSLAM_GRID_SPLIT_MIN_ROWS;   // one problem alone: from this many rows on PA runs as its own many-workgroup launch
constexpr int GRID_SMALL_ROWS = 256;    // problems of at most this many rows run on 256-lane workgroups (MODE 2 only)

// launch groups: 0 = tables in global scratch, 1 = tables in LDS, 2 = everything in LDS / 1024 lanes, 3 = everything in
// LDS / 256 lanes (n1 <= GRID_SMALL_ROWS)
int grid_group(int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs)
{
    const int mode = grid_mode(n1, n2, ncell, n_items, dirs);
    return mode == 2 && n1 <= GRID_SMALL_ROWS ? 3 : mode;
}
// dynamic LDS a problem of the group asks for: group 2 takes everything (one workgroup per CU either way: the spare LDS
// holds the candidates); group 3 adds room for the candidate runs (64 per row) so that several problems share a CU
size_t grid_group_lds_bytes(int group, int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs)
{
    if (group == 0) return 0;
    if (group == 2) return GRID_LDS_MAX_BYTES;
    size_t b = grid_lds_bytes(group == 3 ? 2 : 1, n1, n2, ncell, n_items, dirs);
    if (group == 3) {
        b += 4 * (2 * (size_t)n1 + 1 + 64 * (size_t)n1);
        b = (b + 4095) & ~size_t(4095);
        if (b > GRID_LDS_MAX_BYTES) b = GRID_LDS_MAX_BYTES;
    }
    return b;
}

// one: the descriptor of a lone problem by value (d_probs is not read then); pre / pre_slots: its listed candidates
template <int MODE, int NT>
static int launch_group(const GridDesc* d_probs, int32_t n, size_t lds_bytes, hipStream_t s, const uint32_t* pre = nullptr,
                        uint32_t pre_slots = 0, const GridDesc* one = nullptr, const int32_t* n1_dev = nullptr)
{
    if (n <= 0) return PLSLAM_OK;
    if (MODE > 0) {
        static std::once_flag once;
        static hipError_t attr = hipSuccess;
        std::call_once(once, [] {
            attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_match_grid<MODE, NT, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRID_LDS_MAX_BYTES);
            if (attr == hipSuccess && MODE == 2 && NT == 1024)
                attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_match_grid<2, 1024, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)GRID_LDS_MAX_BYTES);
        });
        PLSLAM_HIP_CHECK(attr);
    }
    static const GridDesc none{};
    if (one && MODE == 2 && NT == 1024)
        hipLaunchKernelGGL((k_match_grid<2, 1024, true>), dim3(1), dim3(1024), lds_bytes, s, nullptr, (uint32_t)(lds_bytes / 4), pre,
                           pre_slots, *one, n1_dev);
    else
        hipLaunchKernelGGL((k_match_grid<MODE, NT, false>), dim3((unsigned)n), dim3(NT), lds_bytes, s, d_probs,
                           (uint32_t)(lds_bytes / 4), pre, pre_slots, none, nullptr);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

// ---- a SMALL lone problem, dense (round 6) ---------------------------------------------------------------------------------
// The line problems of the SLAM loop are 200 x 200 (src/mapHandler.cpp:418, :706; config_kitti.yaml's 200 LSD lines): the
// machinery above -- records found cell by cell on every CU, a list bucketed by column, record passes for items that sit in several
// cells (every line segment does) -- spent 15 + 24 us of kernels on 7 600 candidate pairs.  At this size the problem is a 256 x 256
// bit matrix: ONE workgroup, everything in LDS, no candidate list at all.
//   A  membership: member(i1, i2) = item i2 lies in a cell of a window of row i1 and passes the range and direction tests
//      (GridStructure::get + the two `continue`s of matchGrid) -- four lanes per row walk the window's cell columns, one LDS
//      atomic OR per item
//   B  (mutual) a lane per COLUMN walks the rows in order: the rows that strictly improve the column's running distance are its
//      records -- upstream's `if (d < distances[i2]) ... else continue`, evaluated where it is sequential by definition -- and the
//      last of them is m21
//   C  a lane per ROW folds its live candidates (ascending i2: the defined visiting order) into the best two keys, applies the
//      fp64 ratio test, the mutual check, counts.
// Same results as the kernels above on every problem both accept (tests/test_gpu_match_grid.py runs its cases through both).
constexpr int DENSE_MAX = 256, DENSE_NT = 1024, DENSE_CHUNK = 16;
int g_grid_dense = 1;               // ctx option "grid_dense": 0 = the small lone problem takes the general kernels as before
// LDS words: d1 8 n1 | d2 8 n2 | member 8 n1 | live 8 n1 | any n1 | memberT 8 n2 | m21 n2 | centres 2 nc n1 | R | (dirs: 4 n1 + 4 n2
// doubles' words, 8-byte aligned)
// R is one region with three lives: the grid (cell_start ncell + 1, items) while A runs; the chunk minima of the columns
// (nchunk n2) while B runs; the rows' per-word best pairs (16 n1) while C runs
static size_t dense_region_words(int32_t n1, int32_t n2, int64_t ncell, int32_t n_items)
{
    const size_t nchunk = (size_t)(n1 + DENSE_CHUNK - 1) / DENSE_CHUNK;
    return std::max<size_t>((size_t)ncell + 1 + (size_t)n_items, std::max<size_t>(nchunk * (size_t)n2, 16 * (size_t)n1));
}
size_t grid_dense_lds_bytes(int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs, int32_t n_centres = 2)
{
    size_t w = (size_t)(25 + 2 * n_centres) * (size_t)n1 + (size_t)17 * (size_t)n2 + dense_region_words(n1, n2, ncell, n_items) + 2;
    if (dirs) w += 4 * ((size_t)n1 + (size_t)n2);
    return w * 4;
}
constexpr size_t DENSE_LDS_MAX_BYTES = 128 * 1024;
bool grid_dense_ok(int32_t n1, int32_t n2, int64_t ncell, int32_t n_items, bool dirs, int32_t n_centres)
{
    return g_grid_dense && n1 > 0 && n1 <= DENSE_MAX && n2 > 0 && n2 <= DENSE_MAX && n_centres >= 1 && n_centres <= 4 &&
           grid_dense_lds_bytes(n1, n2, ncell, n_items, dirs, n_centres) <= DENSE_LDS_MAX_BYTES;
}

__global__ void __launch_bounds__(DENSE_NT)
k_match_grid_dense(GridDesc g)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t s_w[];
    const int tid = (int)threadIdx.x, lane = tid & 63;
    const int32_t n1 = g.n1, n2 = g.n2, rows = g.rows, cols = g.cols;
    const int32_t ncell = cols * rows;
    const int32_t nchunk = (n1 + DENSE_CHUNK - 1) / DENSE_CHUNK;
    const bool dirs = g.dir1 != nullptr && g.dir2 != nullptr;
    // (every LDS pointer carries its address space: a generic one makes the compiler emit FLAT accesses)
    typedef PLSLAM_AS_LDS uint32_t* lds_u32;
    typedef PLSLAM_AS_LDS int32_t* lds_i32;
    typedef PLSLAM_AS_LDS u32x4* lds_u32x4;
    const lds_u32 base = (lds_u32)s_w;
    const lds_u32 d1w = base;
    const lds_u32 d2w = d1w + 8 * n1;
    const lds_u32 member = d2w + 8 * n2;
    const lds_u32 live = member + 8 * n1;
    const lds_u32 anyitem = live + 8 * n1;
    const lds_u32 memberT = anyitem + n1;                                // [column][row bits]: a column's 16-row chunk is 16 bits of one word
    const lds_i32 m21 = (lds_i32)(memberT + 8 * n2);
    const lds_i32 scen = m21 + n2;                                       // the window centres
    const lds_u32 region = (lds_u32)(scen + 2 * g.n_centres * n1);
    const lds_u32 cs = region;                                           // life 1: the grid
    const int32_t n_items_decl = g.n_items;
    const lds_i32 items = (lds_i32)(cs + ncell + 1);
    const lds_u32 cmin = region;                                         // life 2: [chunk][column] (d << 8 | row) of the chunk's best row
    const lds_u32 pairs = region;                                        // life 3: [row][word][2] best two keys of the word's candidates
    const int64_t rwords = std::max<int64_t>((int64_t)ncell + 1 + n_items_decl, std::max<int64_t>((int64_t)nchunk * n2, 16 * (int64_t)n1));
    const int64_t dir_off = ((region - base) + rwords + 1) & ~int64_t(1);            // (even word offset from a 16-byte aligned base: 8-byte aligned)
    PLSLAM_AS_LDS double* const sdir = (PLSLAM_AS_LDS double*)(base + dir_off);       // dir1 | dir2
    const lds_u32x4 d1v = (lds_u32x4)d1w, d2v = (lds_u32x4)d2w;
    __shared__ uint32_t s_cnt[DENSE_NT / 64];
#ifdef PLSLAM_DENSE_TIMING
    unsigned long long ts[8]; int nts = 0;
#define DENSE_STAMP() do { __syncthreads(); ts[nts++] = wall_clock64(); } while (0)
#else
#define DENSE_STAMP() do {} while (0)
#endif
    DENSE_STAMP();

    // ---- everything into LDS: the requests of a lane's first pieces of every array go out together (one round trip for the
    // shipped sizes: 64 x 48 cells, a few thousand items); longer arrays continue in loops.  The bit matrices are cleared. ----
    {
        const PLSLAM_AS_GLOBAL u32x4* a = (const PLSLAM_AS_GLOBAL u32x4*)(uintptr_t)g.d1;       // (16-byte aligned: grid_check_problem)
        const PLSLAM_AS_GLOBAL u32x4* b = (const PLSLAM_AS_GLOBAL u32x4*)(uintptr_t)g.d2;
        const PLSLAM_AS_GLOBAL uint32_t* c = (const PLSLAM_AS_GLOBAL uint32_t*)(uintptr_t)g.cell_start;
        const PLSLAM_AS_GLOBAL int32_t* it = (const PLSLAM_AS_GLOBAL int32_t*)(uintptr_t)g.cell_items;
        const PLSLAM_AS_GLOBAL int32_t* cen = (const PLSLAM_AS_GLOBAL int32_t*)(uintptr_t)g.centres;
        const PLSLAM_AS_GLOBAL double* p1 = (const PLSLAM_AS_GLOBAL double*)(uintptr_t)g.dir1;
        const PLSLAM_AS_GLOBAL double* p2 = (const PLSLAM_AS_GLOBAL double*)(uintptr_t)g.dir2;
        constexpr int E = 4;                                              // pieces per lane requested at once
        const int ncen = 2 * g.n_centres * n1;
        u32x4 ra = {0, 0, 0, 0}, rb = {0, 0, 0, 0};
        uint32_t rc[E], ri[E];
        int32_t rce[2] = {0, 0};
        double rd1 = 0.0, rd2 = 0.0;
        if (tid < 2 * n1) ra = a[tid];
        if (tid < 2 * n2) rb = b[tid];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = tid + e * DENSE_NT;
            rc[e] = k <= ncell ? c[k] : 0u;
            ri[e] = k < n_items_decl ? (uint32_t)it[k] : 0u;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) { const int k = tid + e * DENSE_NT; if (k < ncen) rce[e] = cen[k]; }
        if (dirs) {
            if (tid < 2 * n1) rd1 = p1[tid];
            if (tid < 2 * n2) rd2 = p2[tid];
        }
        for (int k = tid; k < 17 * n1; k += DENSE_NT) member[k] = 0u;          // member | live | anyitem
        for (int k = tid; k < 8 * n2; k += DENSE_NT) memberT[k] = 0u;
        if (tid < 2 * n1) d1v[tid] = ra;
        if (tid < 2 * n2) d2v[tid] = rb;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int k = tid + e * DENSE_NT;
            if (k <= ncell) cs[k] = rc[e];
            if (k < n_items_decl) items[k] = (int32_t)ri[e];
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) { const int k = tid + e * DENSE_NT; if (k < ncen) scen[k] = rce[e]; }
        if (dirs) {
            if (tid < 2 * n1) sdir[tid] = rd1;
            if (tid < 2 * n2) sdir[2 * n1 + tid] = rd2;
        }
        for (int k = tid + E * DENSE_NT; k <= ncell; k += DENSE_NT) cs[k] = c[k];
        for (int k = tid + E * DENSE_NT; k < n_items_decl; k += DENSE_NT) items[k] = it[k];
        for (int k = tid + 2 * DENSE_NT; k < ncen; k += DENSE_NT) scen[k] = cen[k];
    }
    __syncthreads();
    // (cell_start is the caller's: an offset beyond the declared item count would read past the copy)
    const uint32_t n_items = cs[ncell] < (uint32_t)n_items_decl ? cs[ncell] : (uint32_t)n_items_decl;

    DENSE_STAMP();
    // ---- A: membership.  A task = (row, centre, cell column of its window): the cells (x, min_y .. max_y - 1) have consecutive
    // ids, i.e. ONE run of the item list.  One LDS atomic OR per hit and matrix (measured: the LDS pipe of the one CU this kernel
    // runs on is what bounds it -- ~2 500 wave-level atomic instructions are 11 of this phase's 12.5 us at 200 x 200 lines.  Built
    // and measured slower: a lane per (row, centre) with masks of its own and no atomics, 25 us -- the serial chain per lane; a wave
    // per row, the lanes' masks OR-ed by shuffles, kernel 22 -> 57 us -- 48 cross-lane exchanges per row through the same LDS pipe;
    // this form with the transposed matrix built afterwards by ballots instead of the second atomic per hit: call 41.8 -> 46.1 us) ----
    {
        const lds_i32 cen = scen;
        const int wx = g.w[0] + g.w[1] + 1;                                 // columns of an unclamped window
        const int per_row = g.n_centres * wx;
        for (int task = tid; task < n1 * per_row; task += DENSE_NT) {
            const int i1 = task / per_row, rem = task - i1 * per_row, c = rem / wx, dx = rem - c * wx;
            const int64_t x = cen[((size_t)i1 * g.n_centres + c) * 2], y = cen[((size_t)i1 * g.n_centres + c) * 2 + 1];
            const int64_t x_ = x - g.w[0] + dx;
            if (x_ < 0 || x_ >= cols) continue;
            const int64_t min_y = y - g.w[2] > 0 ? y - g.w[2] : 0, max_y = y + g.w[3] + 1 < rows ? y + g.w[3] + 1 : rows;
            if (min_y >= max_y) continue;
            uint32_t k0 = cs[x_ * rows + min_y], k1 = cs[x_ * rows + max_y];
            k1 = k1 < n_items ? k1 : n_items;
            if (k0 >= k1) continue;
            anyitem[i1] = 1u;
            double ux = 0.0, uy = 0.0;
            if (dirs) { ux = sdir[2 * i1]; uy = sdir[2 * i1 + 1]; }
            for (uint32_t k = k0; k < k1; ++k) {
                const int32_t i2 = items[k];
                if (i2 < 0 || i2 >= n2) continue;
                if (dirs) {
                    const double dot = ux * sdir[2 * n1 + 2 * i2] + uy * sdir[2 * n1 + 2 * i2 + 1];
                    if (fabs(dot) < g.sim_th) continue;
                }
                const uint32_t bit = 1u << (i2 & 31);
                if (member[8 * i1 + (i2 >> 5)] & bit) continue;          // (seen through another cell: a plain read is cheaper than the atomics)
                atomicOr((uint32_t*)&member[8 * i1 + (i2 >> 5)], bit);
                if (g.mutual) atomicOr((uint32_t*)&memberT[8 * i2 + (i1 >> 5)], 1u << (i1 & 31));
            }
        }
    }
    __syncthreads();

    DENSE_STAMP();
    // ---- B: the columns' records (mutual problems).  A task = (chunk of 16 rows, column): the chunk's member rows are 16 bits of
    // ONE word of the transposed matrix (a fifth of the pairs are members: only those distances are evaluated); the chunk's best
    // (d, row) is published, then -- behind one barrier -- the rows that beat everything in front of them are the column's
    // records: upstream's `if (d < distances[i2]) ... else continue`, evaluated in row order where it is sequential by definition ----
    if (g.mutual) {
        auto dist = [&](int i1, const u32x4& b0, const u32x4& b1) -> uint32_t {
            const u32x4 a0 = d1v[2 * i1], a1 = d1v[2 * i1 + 1];
            return (uint32_t)(__popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                              __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w));
        };
        for (int task = tid; task < nchunk * n2; task += DENSE_NT) {
            const int ch = task / n2, j = task - ch * n2;
            uint32_t bits = (memberT[8 * j + (ch >> 1)] >> (16 * (ch & 1))) & 0xFFFFu;
            uint32_t best = 0xFFFFFFFFu;
            if (bits) {
                const u32x4 b0 = d2v[2 * j], b1 = d2v[2 * j + 1];
                while (bits) {
                    const int i1 = DENSE_CHUNK * ch + __builtin_ctz(bits);
                    bits &= bits - 1u;
                    const uint32_t key = (dist(i1, b0, b1) << 8) | (uint32_t)i1;      // (d, row): the earliest row among equals
                    best = key < best ? key : best;
                }
            }
            cmin[task] = best;
        }
        __syncthreads();
        for (int task = tid; task < nchunk * n2; task += DENSE_NT) {
            const int ch = task / n2, j = task - ch * n2;
            uint32_t bits = (memberT[8 * j + (ch >> 1)] >> (16 * (ch & 1))) & 0xFFFFu;
            if (bits) {
                uint32_t run = 0xFFFFu;                                      // the column's distance in front of this chunk
                for (int c2 = 0; c2 < ch; ++c2) { const uint32_t v = cmin[c2 * n2 + j] >> 8; run = v < run ? v : run; }
                const u32x4 b0 = d2v[2 * j], b1 = d2v[2 * j + 1];
                const uint32_t bit = 1u << (j & 31);
                while (bits) {
                    const int i1 = DENSE_CHUNK * ch + __builtin_ctz(bits);
                    bits &= bits - 1u;
                    const uint32_t d = dist(i1, b0, b1);
                    if (d < run) {                                           // upstream: `if (d < distances[i2])`
                        run = d;
                        atomicOr((uint32_t*)&live[8 * i1 + (j >> 5)], bit);
                    }
                }
            }
            if (ch == 0) {                                                   // m21: the row of the column's smallest (d, row)
                uint32_t bk = 0xFFFFFFFFu;
                for (int c2 = 0; c2 < nchunk; ++c2) { const uint32_t v = cmin[c2 * n2 + j]; bk = v < bk ? v : bk; }
                m21[j] = bk == 0xFFFFFFFFu ? -1 : (int32_t)(bk & 255u);
            }
        }
        __syncthreads();
    }

    DENSE_STAMP();
    // ---- C: rows.  A task = (row, word of its candidate mask): the word's candidates folded into the best two keys (ascending
    // i2 inside the word, words merged in ascending order: upstream's strict `<` updates); then a lane per row ----
    const lds_u32 mask = g.mutual ? live : member;
    for (int task = tid; task < 8 * n1; task += DENSE_NT) {
        const int i1 = task >> 3, w = task & 7;
        uint32_t bits = mask[task];
        uint32_t k1 = KEY_NONE, k2 = KEY_NONE;
        if (bits) {
            const u32x4 a0 = d1v[2 * i1], a1 = d1v[2 * i1 + 1];
            while (bits) {
                const int b = __builtin_ctz(bits);
                bits &= bits - 1u;
                const int j = 32 * w + b;
                const u32x4 b0 = d2v[2 * j], b1 = d2v[2 * j + 1];
                const uint32_t d = (uint32_t)(__popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                                              __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w));
                const uint32_t key = (d << KEY_IDX_BITS) | (uint32_t)j;
                if (key < k1) { k2 = k1; k1 = key; }
                else if (key < k2) k2 = key;
            }
        }
        pairs[2 * task] = k1;
        pairs[2 * task + 1] = k2;
    }
    __syncthreads();
    uint32_t cnt = 0;
    PLSLAM_AS_GLOBAL int32_t* const out = (PLSLAM_AS_GLOBAL int32_t*)(uintptr_t)g.matches_12;
    for (int i1 = tid; i1 < n1; i1 += DENSE_NT) {
        uint32_t k1 = KEY_NONE, k2 = KEY_NONE;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const uint32_t p1 = pairs[16 * i1 + 2 * w], p2 = pairs[16 * i1 + 2 * w + 1];      // (distinct keys: a column appears once)
            if (p1 < k1) { k2 = k1 < p2 ? k1 : p2; k1 = p1; }
            else if (p1 < k2) k2 = p1;
        }
        int32_t m = -1;
        if (k1 != KEY_NONE) {
            const double best_d = (double)(int32_t)(k1 >> KEY_IDX_BITS);
            const double best_d2 = k2 == KEY_NONE ? 2147483647.0 : (double)(int32_t)(k2 >> KEY_IDX_BITS);
            if (best_d < best_d2 * g.nnr) {
                const int32_t i2 = (int32_t)(k1 & KEY_IDX_MASK);
                if (!g.mutual || m21[i2] == i1) m = i2;
            }
        } else if (2147483647.0 < 2147483647.0 * g.nnr && anyitem[i1]) {
            cnt += 1;       // upstream, nnr > 1 only: a row whose candidates all fail passes `best_d < best_d2 * nnr` with best_idx = -1 and is COUNTED
        }
        out[i1] = m;
        cnt += m >= 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, o);
    if (lane == 0) s_cnt[tid >> 6] = cnt;
    __syncthreads();
    if (tid == 0 && g.n_matches) {
        uint32_t all = 0;
        for (int w = 0; w < DENSE_NT / 64; ++w) all += s_cnt[w];
        *(PLSLAM_AS_GLOBAL int32_t*)(uintptr_t)g.n_matches = (int32_t)all;
    }
#ifdef PLSLAM_DENSE_TIMING
    DENSE_STAMP();
    if (tid == 0) printf("[k_match_grid_dense n1=%d n2=%d] load %d A %d B %d C %d (x10 ns)\n", n1, n2, (int)(ts[1] - ts[0]), (int)(ts[2] - ts[1]),
                         (int)(ts[3] - ts[2]), (int)(ts[4] - ts[3]));
#endif
#undef DENSE_STAMP
}

// ONE problem on `s`: a mutual problem that runs LDS-resident with packed candidate words (what k_match_grid decides for
// itself: row and column numbers of 23 bits together) and has enough rows to be worth a second launch gets its distances
// from k_grid_candidates on many workgroups, then k_match_grid<2, 1024> with pre = 1; everything else is one launch.
// aux = grid_aux_words(n2) device words the two launches share -- [0] the candidate list's length -- holding zero when the
// launches reach them: callers upload an image anyway and put them there (grid_aux_fill).  Without it (nullptr) the problem is
// one launch.
size_t grid_aux_words(int32_t) { return 4; }
void grid_aux_fill(void* host_image, int32_t n2) { memset(host_image, 0, grid_aux_words(n2) * 4); }
// n1_upper_bound: q.n1 is an upper bound (the descriptor's n1 is patched on the device): both launches go out whenever the
// problem runs LDS-resident at the bound -- the kernels decide for themselves whether the row count admits the packed words.
int grid_launch_single(const plslam_grid_problem& q, const GridDesc* d_desc, hipStream_t s, uint32_t* aux, bool n1_upper_bound,
                       const GridDesc* h_desc)
{
    const int64_t ncell = (int64_t)q.grid_cols * q.grid_rows;
    const bool dirs = q.dir1 != nullptr && q.dir2 != nullptr;
    // a small lone problem whose row count the host knows: one workgroup, dense (k_match_grid_dense)
    if (h_desc && !n1_upper_bound && grid_dense_ok(q.n1, q.n2, ncell, q.n_items, dirs, q.n_centres)) {
        static std::once_flag once;
        static hipError_t attr = hipSuccess;
        std::call_once(once, [] {
            attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_match_grid_dense), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)DENSE_LDS_MAX_BYTES);
        });
        PLSLAM_HIP_CHECK(attr);
        hipLaunchKernelGGL(k_match_grid_dense, dim3(1), dim3(DENSE_NT), grid_dense_lds_bytes(q.n1, q.n2, ncell, q.n_items, dirs, q.n_centres), s, *h_desc);
        PLSLAM_HIP_CHECK(hipGetLastError());
        return PLSLAM_OK;
    }
    int group = grid_group(q.n1, q.n2, ncell, q.n_items, dirs);
    // ONE problem: nothing shares the CU, and a mutual problem of <= 256 rows still has up to 1024 (row, window part) tasks
    if (group == 3 && q.mutual && q.n1 * GRID_SPLIT > GRID_SMALL_ROWS) group = 2;
    uint32_t fb2 = 1;
    while (fb2 < 22 && (1u << fb2) < (uint32_t)q.n2) ++fb2;
    const uint32_t fb1 = 23u - fb2 > 14u ? 14u : 23u - fb2;
    const bool flat = q.mutual && (uint32_t)q.n2 <= (1u << fb2) && (uint32_t)q.n1 <= (1u << fb1);
    if (group == 2 && (flat || (n1_upper_bound && q.mutual)) && aux && q.n1 >= GRID_SPLIT_MIN_ROWS && q.n2 > 0 && q.pair_capacity > 0) {
        const int64_t wx = std::min<int64_t>((int64_t)q.window[0] + q.window[1] + 1, q.grid_cols);
        const int split = (int)std::max<int64_t>(1, std::min<int64_t>(wx, GRID_SPLIT_MAX));
        const unsigned nwg = (unsigned)(((int64_t)q.n1 * split + 255) / 256);
        // the records of each column, found cell by cell (k_grid_records: the descriptor goes by value, so the row count must
        // be the host's), or every candidate pair (k_grid_candidates)
        const int64_t n_groups = (int64_t)q.grid_cols * ((q.grid_rows + REC_G - 1) / REC_G);
        const size_t lds = grid_group_lds_bytes(2, q.n1, q.n2, ncell, q.n_items, dirs);
        if (PLSLAM_GRID_RECORDS && h_desc && n_groups <= REC_GROUPS_MAX && q.n1 <= REC_ROWS_MAX &&
            (int64_t)q.n_items * REC_SLOT <= (int64_t)q.pair_capacity) {
            // (n1_upper_bound: the row count is the device descriptor's, patched by the caller's kernels)
            const int32_t* n1_dev = n1_upper_bound ? &d_desc->n1 : nullptr;
            hipLaunchKernelGGL(k_grid_records, dim3((unsigned)n_groups), dim3(REC_NT), 0, s, *h_desc, aux, n1_dev);
            PLSLAM_HIP_CHECK(hipGetLastError());
            return launch_group<2, 1024>(d_desc, 1, lds, s, aux, REC_SLOT, h_desc, n1_dev);
        }
        hipLaunchKernelGGL(k_grid_candidates, dim3(nwg), dim3(256), 0, s, d_desc, aux, split);
        PLSLAM_HIP_CHECK(hipGetLastError());
        return launch_group<2, 1024>(d_desc, 1, lds, s, aux);
    }
    int32_t n_mode[4] = {0, 0, 0, 0};
    size_t lds_bytes[4] = {0, 0, 0, 0};
    n_mode[group] = 1;
    lds_bytes[group] = grid_group_lds_bytes(group, q.n1, q.n2, ncell, q.n_items, dirs);
    return launch_match_grid(d_desc, n_mode, lds_bytes, s);
}

// d_probs: the problems of group 3 first, then group 2, 1, 0; lds_bytes[g] = the largest grid_group_lds_bytes() in group g
int launch_match_grid(const GridDesc* d_probs, const int32_t n[4], const size_t lds_bytes[4], hipStream_t s)
{
    int rc;
    if ((rc = launch_group<2, 256>(d_probs, n[3], lds_bytes[3], s))) return rc;
    if ((rc = launch_group<2, 1024>(d_probs + n[3], n[2], lds_bytes[2], s))) return rc;
    if ((rc = launch_group<1, 1024>(d_probs + n[3] + n[2], n[1], lds_bytes[1], s))) return rc;
    return launch_group<0, 1024>(d_probs + n[3] + n[2] + n[1], n[0], 0, s);
}

}  // namespace plslam

// ---------------------------------------------------------------------------------------------
// C ABI (include/plslam_hip.h)
// ---------------------------------------------------------------------------------------------
using namespace plslam;

struct plslam_grid_plan {
    plslam_ctx* ctx = nullptr;
    int32_t nprob = 0;
    int32_t n_mode[4] = {0, 0, 0, 0};  // the table holds the problems of launch group 3 first, then 2, 1, 0
    size_t lds_bytes[4] = {0, 0, 0, 0};   // largest LDS request of a problem of each group
    DevBuf table, scratch, status;
};

static size_t grid_prob_scratch(const plslam_grid_problem& q)
{
    return (grid_scratch_words(q.n1, q.n2, (int64_t)q.grid_cols * q.grid_rows, q.pair_capacity) + 63) & ~size_t(63);
}

// device_rows: d1 / d2 are the pointers the kernels will read (16-byte vector loads); host rows are staged into aligned
// device memory first and may sit anywhere
static int grid_check_problem(const plslam_grid_problem& q, bool device_rows = true)
{
    PLSLAM_REQUIRE(q.n1 >= 0 && q.n2 >= 0 && q.n_centres >= 1 && q.grid_cols >= 1 && q.grid_rows >= 1,
                   PLSLAM_EINVAL);
    PLSLAM_REQUIRE((int64_t)q.grid_cols * q.grid_rows < (int64_t(1) << 31) - 1, PLSLAM_ERANGE);
    PLSLAM_REQUIRE(q.n1 < PLSLAM_MAX_GRID_ROWS && q.n2 <= PLSLAM_MAX_TRAIN_ROWS, PLSLAM_ERANGE);
    PLSLAM_REQUIRE(q.window[0] >= 0 && q.window[1] >= 0 && q.window[2] >= 0 && q.window[3] >= 0,
                   PLSLAM_EINVAL);
    PLSLAM_REQUIRE(q.pair_capacity >= 0 && q.n_items >= 0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(q.cell_start != nullptr && (q.n_items == 0 || q.cell_items != nullptr), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(q.n1 == 0 || (q.d1 && q.centres1 && q.matches_12), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(q.n2 == 0 || q.d2, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(!device_rows || (((uintptr_t)q.d1 & 15) == 0 && ((uintptr_t)q.d2 & 15) == 0), PLSLAM_EINVAL);
    PLSLAM_REQUIRE((q.dir1 == nullptr) == (q.dir2 == nullptr) || q.n1 == 0 || q.n2 == 0, PLSLAM_EINVAL);
    return PLSLAM_OK;
}

static void grid_fill_desc(const plslam_grid_problem& q, uint32_t* scratch, int32_t* status, GridDesc* d)
{
    d->d1 = q.d1; d->d2 = q.d2; d->centres = q.centres1;
    d->cell_start = q.cell_start; d->cell_items = q.cell_items;
    d->dir1 = q.dir1; d->dir2 = q.dir2;
    d->matches_12 = q.matches_12; d->n_matches = q.n_matches;
    d->scratch = scratch; d->status = status;
    d->sim_th = q.sim_th; d->nnr = q.nnr;
    d->n1 = q.n1; d->n2 = q.n2; d->n_centres = q.n_centres; d->cols = q.grid_cols; d->rows = q.grid_rows;
    d->mutual = q.mutual ? 1 : 0;
    for (int k = 0; k < 4; ++k) d->w[k] = q.window[k];
    d->pair_cap = q.pair_capacity;
    d->n_items = q.n_items;
}

namespace plslam {
// One problem with DEVICE pointers, in two steps so that the descriptor can travel inside a larger upload of the caller:
// grid_prepare_one checks the problem and writes its GridDesc to h_desc_slot (host); grid_launch_prepared launches it once
// that descriptor is at d_desc_slot on the device.
int grid_prepare_one(const plslam_grid_problem& q, uint32_t* scratch, int32_t* status, GridDesc* h_desc_slot)
{
    int rc;
    if ((rc = grid_check_problem(q))) return rc;
    grid_fill_desc(q, scratch, status, h_desc_slot);
    return PLSLAM_OK;
}
int grid_launch_prepared(const plslam_grid_problem& q, const GridDesc* d_desc_slot, hipStream_t s)
{
    return grid_launch_single(q, d_desc_slot, s, nullptr, false, nullptr);       // (no shared words: one launch)
}
// h_desc_slot must stay valid until the copy is done (pinned or synchronised by the caller)
int launch_match_grid_one(const plslam_grid_problem& q, uint32_t* scratch, int32_t* status, GridDesc* d_desc_slot,
                          GridDesc* h_desc_slot, hipStream_t s)
{
    int rc;
    if ((rc = grid_prepare_one(q, scratch, status, h_desc_slot))) return rc;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d_desc_slot, h_desc_slot, sizeof(GridDesc), hipMemcpyHostToDevice, s));
    return grid_launch_single(q, d_desc_slot, s, nullptr, false, nullptr);
}
}  // namespace plslam

extern "C" {

int plslam_grid_plan_create(plslam_ctx* ctx, const plslam_grid_problem* probs, int32_t nprob,
                            plslam_grid_plan** out)
{
    PLSLAM_REQUIRE(ctx && out && nprob >= 0 && (nprob == 0 || probs), PLSLAM_EINVAL);
    *out = nullptr;
    int rc;
    size_t words = 0;
    for (int32_t b = 0; b < nprob; ++b) {
        if ((rc = grid_check_problem(probs[b]))) return rc;
        words += grid_prob_scratch(probs[b]);
    }
    DeviceGuard g(ctx->device);
    plslam_grid_plan* P = new (std::nothrow) plslam_grid_plan();
    PLSLAM_REQUIRE(P != nullptr, PLSLAM_ENOMEM);
    P->ctx = ctx;
    P->nprob = nprob;
    auto fail = [&](int code) { plslam_grid_plan_destroy(P); return code; };
    if ((rc = P->table.reserve(sizeof(GridDesc) * (size_t)(nprob ? nprob : 1)))) return fail(rc);
    if ((rc = P->scratch.reserve(words * 4 + 256))) return fail(rc);
    if ((rc = P->status.reserve(256))) return fail(rc);
    std::vector<GridDesc> tab((size_t)nprob);
    size_t off = 0;
    int32_t slot = 0;
    for (int mode = 3; mode >= 0; --mode)
        for (int32_t b = 0; b < nprob; ++b) {
            const int64_t ncell = (int64_t)probs[b].grid_cols * probs[b].grid_rows;
            const bool dirs = probs[b].dir1 != nullptr && probs[b].dir2 != nullptr;
            if (grid_group(probs[b].n1, probs[b].n2, ncell, probs[b].n_items, dirs) != mode) continue;
            const size_t lb = grid_group_lds_bytes(mode, probs[b].n1, probs[b].n2, ncell, probs[b].n_items, dirs);
            if (lb > P->lds_bytes[mode]) P->lds_bytes[mode] = lb;
            ++P->n_mode[mode];
            grid_fill_desc(probs[b], P->scratch.as<uint32_t>() + off, P->status.as<int32_t>(), &tab[slot++]);
            off += grid_prob_scratch(probs[b]);
        }
    if (hipMemset(P->status.p, 0, 256) != hipSuccess ||
        (nprob && hipMemcpy(P->table.p, tab.data(), sizeof(GridDesc) * (size_t)nprob, hipMemcpyHostToDevice) !=
                      hipSuccess)) {
        set_last_error("%s:%d: upload of the grid problem table failed", __FILE__, __LINE__);
        return fail(PLSLAM_EHIP);
    }
    *out = P;
    return PLSLAM_OK;
}

int plslam_grid_plan_run(plslam_grid_plan* plan, void* stream)
{
    PLSLAM_REQUIRE(plan != nullptr, PLSLAM_EINVAL);
    DeviceGuard g(plan->ctx->device);
    return launch_match_grid(plan->table.as<GridDesc>(), plan->n_mode, plan->lds_bytes,
                             stream ? static_cast<hipStream_t>(stream) : plan->ctx->stream);
}

int plslam_grid_plan_overflows(plslam_grid_plan* plan, void* stream, int32_t* n_overflows)
{
    PLSLAM_REQUIRE(plan && n_overflows, PLSLAM_EINVAL);
    DeviceGuard g(plan->ctx->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : plan->ctx->stream;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(n_overflows, plan->status.p, 4, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipMemsetAsync(plan->status.p, 0, 4, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return PLSLAM_OK;
}

void plslam_grid_plan_destroy(plslam_grid_plan* plan)
{
    if (!plan) return;
    DeviceGuard g(plan->ctx->device);
    plan->table.release();
    plan->scratch.release();
    plan->status.release();
    delete plan;
}

int plslam_match_grid(plslam_ctx* ctx, const int32_t* centres1, int32_t n_centres, const uint8_t* d1,
                      int32_t n1, const int32_t* cell_start, const int32_t* cell_items, int32_t grid_cols,
                      int32_t grid_rows, const uint8_t* d2, int32_t n2, const double* dir1,
                      const double* dir2, double sim_th, const int32_t window[4], double nnr, int mutual,
                      int32_t* matches_12, int32_t* n_matches)
{
    PLSLAM_REQUIRE(ctx && window, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(n1 >= 0, PLSLAM_EINVAL);
    if (n_matches) *n_matches = 0;
    if (n1 == 0) return PLSLAM_OK;
    plslam_grid_problem q{};
    q.d1 = d1; q.d2 = d2; q.centres1 = centres1; q.cell_start = cell_start; q.cell_items = cell_items;
    q.dir1 = dir1; q.dir2 = dir2;
    q.n1 = n1; q.n2 = n2; q.n_centres = n_centres; q.grid_cols = grid_cols; q.grid_rows = grid_rows;
    q.n_items = 0;   // validated and set below
    for (int k = 0; k < 4; ++k) q.window[k] = window[k];
    q.sim_th = sim_th; q.nnr = nnr; q.mutual = mutual;
    q.matches_12 = matches_12;
    int rc;
    if ((rc = grid_check_problem(q, false))) return rc;
    // the grid is host data here: validate the offsets and count the (row, candidate) pairs exactly.  The ENTRIES are
    // not validated: an entry outside [0, n2) is skipped by the kernel before any read, as upstream's loop skips it
    // (`if (i2 < 0 || i2 >= desc2.rows) continue;`) -- tests/test_gpu_match_grid.py::test_empty_and_out_of_range_inputs
    const int64_t ncell = (int64_t)grid_cols * grid_rows;
    PLSLAM_REQUIRE(cell_start[0] == 0, PLSLAM_EINVAL);
    for (int64_t c = 0; c < ncell; ++c) PLSLAM_REQUIRE(cell_start[c + 1] >= cell_start[c], PLSLAM_EINVAL);
    const int32_t n_items = cell_start[ncell];
    PLSLAM_REQUIRE(n_items == 0 || cell_items, PLSLAM_EINVAL);
    q.n_items = n_items;
    // capacity of the candidate store.  A bound from the grid alone (fullest cell x cells of a window, at most every item,
    // per window centre; rows in blocks of 1024) costs one pass over cell_start; only when that bound is large is the
    // exact figure worth a walk over every row's window.
    int64_t pairs = grid_store_capacity_bound(n1, n_centres, cell_start, grid_cols, grid_rows, window, mutual);
    if (pairs > (int64_t(1) << 21))
        pairs = grid_store_capacity_host(centres1, n1, n_centres, cell_start, grid_cols, grid_rows, window, mutual);
    PLSLAM_REQUIRE(pairs < (int64_t(1) << 31) - 1, PLSLAM_ERANGE);
    q.pair_capacity = (int32_t)pairs;

    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    // ONE pinned staging block -> one H2D copy: [GridDesc | centres | cell_start | cell_items | d1 | d2 | dir1 | dir2]
    Carver ci;
    const bool dirs = dir1 && dir2 && n2 > 0;
    const size_t oT = ci.take(sizeof(GridDesc)), oX = ci.take(grid_aux_words(n2) * 4), oC = ci.take((size_t)n1 * n_centres * 8),
                 oS = ci.take((size_t)(ncell + 1) * 4), oI = ci.take((size_t)n_items * 4),
                 oA = ci.take((size_t)n1 * 32), oB = ci.take((size_t)n2 * 32),
                 oD1 = ci.take(dirs ? (size_t)n1 * 16 : 0), oD2 = ci.take(dirs ? (size_t)n2 * 16 : 0);
    Carver co;
    const size_t oM = co.take((size_t)n1 * 4), oN = co.take(8);   // n_matches, status
    if ((rc = ctx->pin_in.reserve(ci.off))) return rc;
    if ((rc = ctx->in_a.reserve(ci.off))) return rc;
    if ((rc = ctx->pin_out.reserve(co.off))) return rc;
    if ((rc = ctx->out_a.reserve(co.off))) return rc;
    if ((rc = ctx->misc_a.reserve(grid_scratch_words(n1, n2, ncell, q.pair_capacity) * 4 + 256))) return rc;
    char* h = ctx->pin_in.as<char>();
    char* d = ctx->in_a.as<char>();
    char* dout = ctx->out_a.as<char>();
    // (option "zero_copy_kb": a small upload image is read by the kernels where it lies in page-locked host memory -- the copy
    // command in front of them, with its completion signal, is the larger part of such a call's device-side time)
    // Taken where it was measured to pay (profiles/r6_r_grid_latency_dense_zero_copy.txt): a problem the dense one-workgroup kernel
    // takes -- it reads every input word ONCE, into LDS: 200 x 200 lines 57.3 -> 50.9 us per call; the general kernels walk the
    // cells and the descriptors again and again (neutral to 64 kB, slower beyond) and keep the copy unless the option is negative
    // (-kb: every problem whose image fits |kb|).
    bool zero_copy = false;
    const bool dense = grid_dense_ok(n1, n2, ncell, n_items, dirs, n_centres);
    const size_t zc_limit = (size_t)(ctx->zero_copy_kb < 0 ? -ctx->zero_copy_kb : ctx->zero_copy_kb) * 1024;
    if (zc_limit > 0 && ci.off <= zc_limit && (dense || ctx->zero_copy_kb < 0))
        if (char* m = static_cast<char*>(ctx->pin_in.dev)) { d = m; zero_copy = true; }
    memcpy(h + oC, centres1, (size_t)n1 * n_centres * 8);
    memcpy(h + oS, cell_start, (size_t)(ncell + 1) * 4);
    if (n_items) memcpy(h + oI, cell_items, (size_t)n_items * 4);
    memcpy(h + oA, d1, (size_t)n1 * 32);
    if (n2) memcpy(h + oB, d2, (size_t)n2 * 32);
    if (dirs) {
        memcpy(h + oD1, dir1, (size_t)n1 * 16);
        memcpy(h + oD2, dir2, (size_t)n2 * 16);
    }
    plslam_grid_problem dq = q;
    dq.centres1 = (const int32_t*)(d + oC);
    dq.cell_start = (const int32_t*)(d + oS);
    dq.cell_items = (const int32_t*)(d + oI);
    dq.d1 = (const uint8_t*)(d + oA);
    dq.d2 = (const uint8_t*)(d + oB);
    dq.dir1 = dirs ? (const double*)(d + oD1) : nullptr;
    dq.dir2 = dirs ? (const double*)(d + oD2) : nullptr;
    // results: the kernel writes the table, the count and the status word straight into the page-locked block when the
    // device can address it (one copy-engine command less on the call's critical path)
    char* hout_dev = static_cast<char*>(ctx->pin_out.dev);
    int32_t* hres = (int32_t*)(ctx->pin_out.as<char>() + oN);
    if (hout_dev) {
        hres[0] = hres[1] = 0;
        dout = hout_dev;
    }
    dq.matches_12 = (int32_t*)(dout + oM);
    dq.n_matches = (int32_t*)(dout + oN);
    if ((rc = grid_check_problem(dq))) return rc;                 // what the kernel reads: the staged, aligned rows
    // (no status word over PCIe -- it is bumped with an atomic; an overflow also shows as a count of -1)
    grid_fill_desc(dq, ctx->misc_a.as<uint32_t>(), hout_dev ? nullptr : (int32_t*)(dout + oN) + 1, (GridDesc*)(h + oT));
    grid_aux_fill(h + oX, n2);
    hipStream_t s = ctx->stream;
    StreamSyncOnError sg(s);
    if (!zero_copy) PLSLAM_HIP_CHECK(hipMemcpyAsync(d, h, ci.off, hipMemcpyHostToDevice, s));
    if (!hout_dev) PLSLAM_HIP_CHECK(hipMemsetAsync(dout + oN, 0, 8, s));
    if ((rc = grid_launch_single(dq, (const GridDesc*)(d + oT), s, (uint32_t*)(d + oX), false, (const GridDesc*)(h + oT)))) return rc;
    if (!hout_dev) PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->pin_out.p, dout, co.off, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    const int32_t* res = (const int32_t*)(ctx->pin_out.as<char>() + oN);
    if (res[1] != 0 || res[0] < 0) {   // cannot happen: the capacity above is an upper bound
        set_last_error("%s:%d: matchGrid candidate store overflow (%d slots provided)", __FILE__, __LINE__, (int)pairs);
        return PLSLAM_ERANGE;
    }
    memcpy(matches_12, ctx->pin_out.as<char>() + oM, (size_t)n1 * 4);
    if (n_matches) *n_matches = res[0];
    return PLSLAM_OK;
}

int64_t plslam_grid_pair_capacity(const int32_t* centres1, int32_t n1, int32_t n_centres, const int32_t* cell_start,
                                  int32_t grid_cols, int32_t grid_rows, const int32_t window[4], int mutual)
{
    if (!centres1 || !cell_start || !window || n1 < 0 || n_centres < 1 || grid_cols < 1 || grid_rows < 1) return -1;
    return plslam::grid_store_capacity_host(centres1, n1, n_centres, cell_start, grid_cols, grid_rows, window, mutual);
}

int64_t plslam_grid_pair_capacity_bound(int32_t n1, int32_t n_centres, const int32_t* cell_start, int32_t grid_cols,
                                        int32_t grid_rows, const int32_t window[4], int mutual)
{
    if (!cell_start || !window || n1 < 0 || n_centres < 1 || grid_cols < 1 || grid_rows < 1) return -1;
    return plslam::grid_store_capacity_bound(n1, n_centres, cell_start, grid_cols, grid_rows, window, mutual);
}

}  // extern "C"
