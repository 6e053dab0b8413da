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
//     (i1, i2) takes part in row i1's best / second best   <=>   i1 == min{ i1' : i2 in C(i1'), d(i1',i2) <= d(i1,i2) }
//     matches_21[i2] = the row of the lexicographic minimum (d, i1) of column i2
// (the tests check this form against a literal restatement of the sequential loop).  Ties between equally distant best candidates go to the lowest i2 (upstream: the
// iteration order of a std::unordered_set<int>, implementation-defined).
//
// One workgroup of 1024 lanes per problem, phases separated by workgroup barriers; every cross-lane combination is
// a min of composite keys or an integer count, so the result does not depend on scheduling:
//   P1 (lane per row)     count the candidates of every column i2 (the direction test of the line overload applied)
//   P2                    exclusive scan of the counts -> column lists in CSR form (workgroup scan through LDS)
//   P3 (lane per row)     d = popcount(desc1[i1] ^ desc2[i2]); append (d << 22 | i1) to column i2's list (unordered;
//                         nothing below depends on the order); atomicMin the column's (d, i1) key
//   P4 (wave per column)  first[d] = min i1 of the entries at distance d (257 bins in LDS, ds_min_u32), prefix-min over
//                         d; an entry is live iff that prefix minimum is its own row; live entries fold (d << 23 | i2)
//                         into the row's best key with atomicMin and keep a flag bit in the list
//   P5 (wave per column)  live entries other than the row's best fold into the row's second-best key
//   P6 (lane per row)     ratio test `best_d < best_d2 * nnr` in fp64 (int * double upstream; best_d2 = INT_MAX when
//                         absent, so a single live candidate passes), mutual check, count
// A row's window is 7x7 cells at the shipped configuration (matching_f2f_ws = 3), a few dozen candidates, so a
// problem is ~10^4..10^5 distances: latency-bound.  Duplicated candidates (an item sitting in several cells of the
// window, the two windows of a line overlapping) appear several times in the lists; every use is idempotent.
#include <cstring>
#include <new>

#include "common.hpp"

namespace plslam {
namespace {

constexpr int GRID_THREADS = 1024, GRID_WAVES = GRID_THREADS / 64;
constexpr uint32_t ROW_BITS = 22, ROW_MASK = (1u << ROW_BITS) - 1u;   // pair entry: live << 31 | d << 22 | i1
constexpr uint32_t LIVE_BIT = 0x80000000u;
constexpr int NBINS = 320;                                             // 257 distance values, 5 per lane

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t ld_coherent(const uint32_t* p)
{   // values other lanes produced with atomics earlier in this kernel
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// GridStructure::get over every window centre of row i1; f(i2) for each candidate that survives the range check
// (`if (i2 < 0 || i2 >= desc2.rows) continue;`) and the direction test of the line overload
template <class F>
__device__ __forceinline__ void for_candidates(const GridDesc& g, int32_t i1, F&& f)
{
    double a0 = 0.0, a1 = 0.0;
    const bool dirs = g.dir1 != nullptr && g.dir2 != nullptr;
    if (dirs) {
        a0 = g.dir1[2 * (int64_t)i1];
        a1 = g.dir1[2 * (int64_t)i1 + 1];
    }
    for (int32_t c = 0; c < g.n_centres; ++c) {
        const int32_t* p = g.centres + ((int64_t)i1 * g.n_centres + c) * 2;
        const int64_t x = p[0], y = p[1];
        const int64_t min_x = x - g.w[0] > 0 ? x - g.w[0] : 0;
        const int64_t max_x = x + g.w[1] + 1 < g.cols ? x + g.w[1] + 1 : g.cols;
        const int64_t min_y = y - g.w[2] > 0 ? y - g.w[2] : 0;
        const int64_t max_y = y + g.w[3] + 1 < g.rows ? y + g.w[3] + 1 : g.rows;
        if (min_y >= max_y) continue;
        for (int64_t x_ = min_x; x_ < max_x; ++x_) {
            // cells (x_, min_y .. max_y-1) are adjacent in the CSR order (id = x*rows + y)
            const int32_t s = g.cell_start[x_ * g.rows + min_y], e = g.cell_start[x_ * g.rows + max_y];
            for (int32_t k = s; k < e; ++k) {
                const int32_t i2 = g.cell_items[k];
                if ((uint32_t)i2 >= (uint32_t)g.n2) continue;
                if (dirs) {
                    const double dot = a0 * g.dir2[2 * (int64_t)i2] + a1 * g.dir2[2 * (int64_t)i2 + 1];
                    if (fabs(dot) < g.sim_th) continue;       // NaN (zero-length direction) compares false: kept
                }
                f(i2);
            }
        }
    }
}

__global__ __launch_bounds__(GRID_THREADS) void k_match_grid(const GridDesc* __restrict__ probs)
{
    __shared__ uint32_t s_part[GRID_THREADS];
    __shared__ uint32_t s_wave[GRID_WAVES];
    __shared__ uint32_t s_bins[GRID_WAVES][NBINS];
    __shared__ uint32_t s_total;

    const GridDesc g = probs[blockIdx.x];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t n1 = g.n1, n2 = g.n2;
    uint32_t* col_start = g.scratch;                     // n2 + 1 (counts during P1)
    uint32_t* col_fill = col_start + (n2 + 1);           // n2
    uint32_t* col_best = col_fill + n2;                  // n2   (d << 22 | i1) lexicographic column minimum
    uint32_t* row_k1 = col_best + n2;                    // n1   (d << 23 | i2) best live candidate
    uint32_t* row_k2 = row_k1 + n1;                      // n1   second best
    uint32_t* pairs = row_k2 + n1;                       // pair_cap

    // ---- P0 ----
    for (int32_t j = tid; j <= n2; j += GRID_THREADS) col_start[j] = 0u;
    for (int32_t j = tid; j < n2; j += GRID_THREADS) col_best[j] = KEY_NONE;
    for (int32_t i = tid; i < n1; i += GRID_THREADS) {
        row_k1[i] = KEY_NONE;
        row_k2[i] = KEY_NONE;
    }
    __threadfence();
    __syncthreads();

    // ---- P1: column counts ----
    for (int32_t i1 = tid; i1 < n1; i1 += GRID_THREADS)
        for_candidates(g, i1, [&](int32_t i2) { atomicAdd(&col_start[i2], 1u); });
    __threadfence();
    __syncthreads();

    // ---- P2: exclusive scan of col_start[0..n2) in place; col_start[n2] = total ----
    {
        const int32_t per = (n2 + GRID_THREADS - 1) / GRID_THREADS;
        const int32_t b = tid * per < n2 ? tid * per : n2, e = b + per < n2 ? b + per : n2;
        uint32_t sum = 0;
        for (int32_t j = b; j < e; ++j) sum += ld_coherent(&col_start[j]);
        uint32_t incl = sum;                                        // wave inclusive scan
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t v = (uint32_t)__shfl_up((int)incl, off);
            if (lane >= off) incl += v;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        uint32_t base = 0;
        for (int w = 0; w < wave; ++w) base += s_wave[w];
        uint32_t run = base + incl - sum;
        for (int32_t j = b; j < e; ++j) {
            const uint32_t c = ld_coherent(&col_start[j]);
            col_start[j] = run;
            col_fill[j] = run;
            run += c;
        }
        if (tid == GRID_THREADS - 1) {
            col_start[n2] = run;
            s_total = run;
        }
        __threadfence();
        __syncthreads();
    }
    if (s_total > (uint32_t)g.pair_cap) {                           // list does not fit: report, match nothing
        for (int32_t i = tid; i < n1; i += GRID_THREADS) g.matches_12[i] = -1;
        if (tid == 0) {
            if (g.n_matches) *g.n_matches = -1;
            if (g.status) atomicAdd(g.status, 1);
        }
        return;
    }

    // ---- P3: distances; column lists; column minima ----
    for (int32_t i1 = tid; i1 < n1; i1 += GRID_THREADS) {
        uint32_t q[8];
        const uint32_t* qa = reinterpret_cast<const uint32_t*>(g.d1) + (int64_t)i1 * 8;
#pragma unroll
        for (int w = 0; w < 8; ++w) q[w] = qa[w];
        for_candidates(g, i1, [&](int32_t i2) {
            const uint32_t* t = reinterpret_cast<const uint32_t*>(g.d2) + (int64_t)i2 * 8;
            uint32_t d = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) d += (uint32_t)__popc(q[w] ^ t[w]);
            const uint32_t key = (d << ROW_BITS) | (uint32_t)i1;
            const uint32_t pos = atomicAdd(&col_fill[i2], 1u);
            pairs[pos] = key;
            if (g.mutual) atomicMin(&col_best[i2], key);
        });
    }
    __threadfence();
    __syncthreads();

    // ---- P4: live entries -> row best ----
    uint32_t* bins = s_bins[wave];
    for (int32_t i2 = wave; i2 < n2; i2 += GRID_WAVES) {
        const uint32_t s = col_start[i2], e = col_start[i2 + 1];
        if (s == e) continue;
        const bool need_bins = g.mutual && e - s > 1;               // a single entry is its own prefix minimum
        if (need_bins) {
#pragma unroll
            for (int k = 0; k < NBINS / 64; ++k) bins[lane + 64 * k] = KEY_NONE;
            wave_sync();
            for (uint32_t p = s + lane; p < e; p += 64) {
                const uint32_t key = pairs[p];
                atomicMin(&bins[key >> ROW_BITS], key & ROW_MASK);
            }
            wave_sync();
            // prefix minimum over the distance value: lane l owns bins 5l .. 5l+4
            uint32_t v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = bins[5 * lane + k];
#pragma unroll
            for (int k = 1; k < 5; ++k) v[k] = umin32(v[k], v[k - 1]);
            uint32_t incl = v[4];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)incl, off);
                if (lane >= off) incl = umin32(incl, o);
            }
            uint32_t excl = (uint32_t)__shfl_up((int)incl, 1);
            if (lane == 0) excl = KEY_NONE;
#pragma unroll
            for (int k = 0; k < 5; ++k) bins[5 * lane + k] = umin32(v[k], excl);
            wave_sync();
        }
        for (uint32_t p = s + lane; p < e; p += 64) {
            const uint32_t key = pairs[p];
            const uint32_t d = key >> ROW_BITS, i1 = key & ROW_MASK;
            const bool live = !need_bins || bins[d] == i1;
            if (live) {
                pairs[p] = key | LIVE_BIT;
                atomicMin(&row_k1[i1], (d << KEY_IDX_BITS) | (uint32_t)i2);
            }
        }
        wave_sync();                                                // bins are reused by the next column
    }
    __threadfence();
    __syncthreads();

    // ---- P5: live entries other than the row's best -> row second best ----
    for (int32_t i2 = wave; i2 < n2; i2 += GRID_WAVES) {
        const uint32_t s = col_start[i2], e = col_start[i2 + 1];
        for (uint32_t p = s + lane; p < e; p += 64) {
            const uint32_t key = pairs[p];
            if (!(key & LIVE_BIT)) continue;
            const uint32_t d = (key & ~LIVE_BIT) >> ROW_BITS, i1 = key & ROW_MASK;
            const uint32_t rk = (d << KEY_IDX_BITS) | (uint32_t)i2;
            if (rk != ld_coherent(&row_k1[i1])) atomicMin(&row_k2[i1], rk);
        }
    }
    __threadfence();
    __syncthreads();

    // ---- P6: ratio test, mutual check, count ----
    uint32_t cnt = 0;
    for (int32_t i1 = tid; i1 < n1; i1 += GRID_THREADS) {
        const uint32_t k1 = ld_coherent(&row_k1[i1]), k2 = ld_coherent(&row_k2[i1]);
        int32_t m = -1;
        if (k1 != KEY_NONE) {
            const double best_d = (double)(int32_t)(k1 >> KEY_IDX_BITS);
            const double best_d2 = k2 == KEY_NONE ? 2147483647.0 : (double)(int32_t)(k2 >> KEY_IDX_BITS);
            if (best_d < best_d2 * g.nnr) {
                const int32_t i2 = (int32_t)(k1 & KEY_IDX_MASK);
                if (!g.mutual || (ld_coherent(&col_best[i2]) & ROW_MASK) == (uint32_t)i1) m = i2;
            }
        }
        g.matches_12[i1] = m;
        cnt += m >= 0;
    }
    s_part[tid] = cnt;
    __syncthreads();
    for (int st = GRID_THREADS / 2; st > 0; st >>= 1) {
        if (tid < st) s_part[tid] += s_part[tid + st];
        __syncthreads();
    }
    if (tid == 0 && g.n_matches) *g.n_matches = (int32_t)s_part[0];
}

}  // namespace

size_t grid_scratch_words(int32_t n1, int32_t n2, int32_t pair_cap)
{
    return (size_t)(n2 + 1) + 2 * (size_t)n2 + 2 * (size_t)n1 + (size_t)pair_cap;
}

int launch_match_grid(const GridDesc* d_probs, int32_t nprob, hipStream_t s)
{
    if (nprob <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_match_grid, dim3((unsigned)nprob), dim3(GRID_THREADS), 0, s, d_probs);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

}  // namespace plslam

// ---------------------------------------------------------------------------------------------
// C ABI (include/plslam_hip.h)
// ---------------------------------------------------------------------------------------------
using namespace plslam;

struct plslam_grid_plan {
    plslam_ctx* ctx = nullptr;
    int32_t nprob = 0;
    DevBuf table, scratch, status;
};

static int grid_check_problem(const plslam_grid_problem& q)
{
    PLSLAM_REQUIRE(q.n1 >= 0 && q.n2 >= 0 && q.n_centres >= 1 && q.grid_cols >= 1 && q.grid_rows >= 1,
                   PLSLAM_EINVAL);
    PLSLAM_REQUIRE((int64_t)q.grid_cols * q.grid_rows < (int64_t(1) << 31) - 1, PLSLAM_ERANGE);
    PLSLAM_REQUIRE(q.n1 < PLSLAM_MAX_GRID_ROWS && q.n2 <= PLSLAM_MAX_TRAIN_ROWS, PLSLAM_ERANGE);
    PLSLAM_REQUIRE(q.window[0] >= 0 && q.window[1] >= 0 && q.window[2] >= 0 && q.window[3] >= 0,
                   PLSLAM_EINVAL);
    PLSLAM_REQUIRE(q.pair_capacity >= 0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(q.cell_start != nullptr, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(q.n1 == 0 || (q.d1 && q.centres1 && q.matches_12), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(q.n2 == 0 || q.d2, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(((uintptr_t)q.d1 & 3) == 0 && ((uintptr_t)q.d2 & 3) == 0, PLSLAM_EINVAL);
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
}

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
        words += (grid_scratch_words(probs[b].n1, probs[b].n2, probs[b].pair_capacity) + 63) & ~size_t(63);
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
    for (int32_t b = 0; b < nprob; ++b) {
        grid_fill_desc(probs[b], P->scratch.as<uint32_t>() + off, P->status.as<int32_t>(), &tab[b]);
        off += (grid_scratch_words(probs[b].n1, probs[b].n2, probs[b].pair_capacity) + 63) & ~size_t(63);
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
    return launch_match_grid(plan->table.as<GridDesc>(), plan->nprob,
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
    for (int k = 0; k < 4; ++k) q.window[k] = window[k];
    q.sim_th = sim_th; q.nnr = nnr; q.mutual = mutual;
    q.matches_12 = matches_12;
    int rc;
    if ((rc = grid_check_problem(q))) return rc;
    // the grid is host data here: validate it and count the (row, candidate) pairs exactly
    const int64_t ncell = (int64_t)grid_cols * grid_rows;
    PLSLAM_REQUIRE(cell_start[0] == 0, PLSLAM_EINVAL);
    for (int64_t c = 0; c < ncell; ++c) PLSLAM_REQUIRE(cell_start[c + 1] >= cell_start[c], PLSLAM_EINVAL);
    const int32_t n_items = cell_start[ncell];
    PLSLAM_REQUIRE(n_items == 0 || cell_items, PLSLAM_EINVAL);
    int64_t pairs = 0;
    for (int64_t k = 0; k < (int64_t)n1 * n_centres; ++k) {
        const int64_t x = centres1[2 * k], y = centres1[2 * k + 1];
        const int64_t min_x = x - window[0] > 0 ? x - window[0] : 0;
        const int64_t max_x = x + window[1] + 1 < grid_cols ? x + window[1] + 1 : grid_cols;
        const int64_t min_y = y - window[2] > 0 ? y - window[2] : 0;
        const int64_t max_y = y + window[3] + 1 < grid_rows ? y + window[3] + 1 : grid_rows;
        if (min_y >= max_y) continue;
        for (int64_t x_ = min_x; x_ < max_x; ++x_)
            pairs += cell_start[x_ * grid_rows + max_y] - cell_start[x_ * grid_rows + min_y];
    }
    PLSLAM_REQUIRE(pairs < (int64_t(1) << 31) - 1, PLSLAM_ERANGE);
    q.pair_capacity = (int32_t)pairs;

    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard g(ctx->device);
    // ONE pinned staging block -> one H2D copy: [GridDesc | centres | cell_start | cell_items | d1 | d2 | dir1 | dir2]
    Carver ci;
    const bool dirs = dir1 && dir2 && n2 > 0;
    const size_t oT = ci.take(sizeof(GridDesc)), oC = ci.take((size_t)n1 * n_centres * 8),
                 oS = ci.take((size_t)(ncell + 1) * 4), oI = ci.take((size_t)n_items * 4),
                 oA = ci.take((size_t)n1 * 32), oB = ci.take((size_t)n2 * 32),
                 oD1 = ci.take(dirs ? (size_t)n1 * 16 : 0), oD2 = ci.take(dirs ? (size_t)n2 * 16 : 0);
    Carver co;
    const size_t oM = co.take((size_t)n1 * 4), oN = co.take(8);   // n_matches, status
    if ((rc = ctx->pin_in.reserve(ci.off))) return rc;
    if ((rc = ctx->in_a.reserve(ci.off))) return rc;
    if ((rc = ctx->pin_out.reserve(co.off))) return rc;
    if ((rc = ctx->out_a.reserve(co.off))) return rc;
    if ((rc = ctx->misc_a.reserve(grid_scratch_words(n1, n2, q.pair_capacity) * 4))) return rc;
    char* h = ctx->pin_in.as<char>();
    char* d = ctx->in_a.as<char>();
    char* dout = ctx->out_a.as<char>();
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
    dq.matches_12 = (int32_t*)(dout + oM);
    dq.n_matches = (int32_t*)(dout + oN);
    grid_fill_desc(dq, ctx->misc_a.as<uint32_t>(), (int32_t*)(dout + oN) + 1, (GridDesc*)(h + oT));
    hipStream_t s = ctx->stream;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d, h, ci.off, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemsetAsync(dout + oN, 0, 8, s));
    if ((rc = launch_match_grid((const GridDesc*)(d + oT), 1, s))) return rc;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->pin_out.p, dout, co.off, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    const int32_t* res = (const int32_t*)(ctx->pin_out.as<char>() + oN);
    if (res[1] != 0) {   // cannot happen: the capacity above is exact
        set_last_error("%s:%d: matchGrid pair list overflow (%d pairs counted)", __FILE__, __LINE__, (int)pairs);
        return PLSLAM_ERANGE;
    }
    memcpy(matches_12, ctx->pin_out.as<char>() + oM, (size_t)n1 * 4);
    if (n_matches) *n_matches = res[0];
    return PLSLAM_OK;
}

}  // extern "C"
