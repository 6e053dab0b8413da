// hamming.hip -- K1 (256-bit Hamming kNN-2 scans) and K2 (ratio + mutual finalize) for gfx950.
//
// Replaces cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) + stvo-pl matchNNR()/match()
// (reached from src/mapHandler.cpp:277,424,597,712,3223,3249 of the reference).
//
// The XOR + popcount forms of the scan: a 256-bit distance costs 8 v_xor_b32 + 8 v_bcnt_u32_b32 (the
// bcnt accumulates for free); these kernels are VALU-issue bound, HBM sees each descriptor once.
// (The default scan for mutual problems is the matrix-core form K1e in hamming_mfma.hip; the kernels
// here serve directed problems, single small problems, n2 > 2048, and forced scan variants.)
//
// Result order == OpenCV batchDistance(K=2): lexicographic (distance, trainIdx).  It is encoded
// as an unsigned min over composite keys  key = (distance << 23) | trainIdx  so that every
// merge (per lane, across lanes, across workgroups) is associative and tie-exact.
#include "common.hpp"
#include "stereo_gates_dev.hpp"

namespace plslam {

typedef const __attribute__((address_space(4))) uint32_t* sptr_t;  // scalar (SMEM) loads
typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));

// ---- tiny VALU helpers the compiler would otherwise re-associate / not select -----------------
__device__ __forceinline__ uint32_t bcnt0(uint32_t x)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc)
{
    uint32_t r;  // r = popcount(x) + acc in ONE VALU op
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ uint32_t med3_u32(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// key = (d << 23) | j with j wave-uniform (SGPR): one v_lshl_or_b32.  Written as asm because the
// compiler otherwise proves j % 4 == 0 and splits (j + k) into a shift plus a v_or3.
__device__ __forceinline__ uint32_t make_key_s(uint32_t d, uint32_t j_uniform)
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, 23, %2" : "=v"(r) : "v"(d), "s"(j_uniform));
    return r;
}
static_assert(KEY_IDX_BITS == 23, "make_key_s hard-codes the shift");
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// keep the two smallest keys; invariant b0 <= b1
__device__ __forceinline__ void best2_push(uint32_t& b0, uint32_t& b1, uint32_t key)
{
    const uint32_t nb1 = med3_u32(b0, b1, key);
    b0 = umin(b0, key);
    b1 = nb1;
}
// merge two sorted pairs
__device__ __forceinline__ void best2_merge(uint32_t& a0, uint32_t& a1, uint32_t c0, uint32_t c1)
{
    const uint32_t lo = umin(a0, c0);
    const uint32_t hi = umin(umax(a0, c0), umin(a1, c1));
    a0 = lo;
    a1 = hi;
}

// 8 XOR + 8 BCNT: q in VGPRs, train row in SGPRs (wave-uniform)
#define PLSLAM_DIST8(q, tp)                                                                      \
    bcnt_acc(q[7] ^ (tp)[7],                                                                     \
      bcnt_acc(q[6] ^ (tp)[6],                                                                   \
        bcnt_acc(q[5] ^ (tp)[5],                                                                 \
          bcnt_acc(q[4] ^ (tp)[4],                                                               \
            bcnt_acc(q[3] ^ (tp)[3],                                                             \
              bcnt_acc(q[2] ^ (tp)[2],                                                           \
                bcnt_acc(q[1] ^ (tp)[1], bcnt0(q[0] ^ (tp)[0]))))))))

// same with the popcount chain started from `bias` (a VGPR) instead of 0: d + bias at no extra cost
#define PLSLAM_DIST8B(q, tp, bias)                                                               \
    bcnt_acc(q[7] ^ (tp)[7],                                                                     \
      bcnt_acc(q[6] ^ (tp)[6],                                                                   \
        bcnt_acc(q[5] ^ (tp)[5],                                                                 \
          bcnt_acc(q[4] ^ (tp)[4],                                                               \
            bcnt_acc(q[3] ^ (tp)[3],                                                             \
              bcnt_acc(q[2] ^ (tp)[2],                                                           \
                bcnt_acc(q[1] ^ (tp)[1], bcnt_acc(q[0] ^ (tp)[0], bias))))))))

// XCD-striped block tables: hardware places workgroup b on XCD b % 8 and dispatches in increasing
// b; the host lays the table out as 8 rows of L = gridDim.x / 8 entries, row x = the work of XCD x in
// dispatch order (capi.hip, `stripe`), so the blocks of one problem -- which stream the same
// descriptor sets -- sit on one XCD's L2 at the same time.  Rows are padded with item = -1.
__device__ __forceinline__ int xcd_remap(int orig, int nwg)
{
    return (orig & 7) * (nwg >> 3) + (orig >> 3);
}

// ---------------------------------------------------------------------------------------------
// K1a  lane-per-query scan.  One lane owns one query row (8 VGPRs) for the whole scan and keeps
// its private best-2; train rows are wave-uniform and arrive through the scalar data cache as
// s_load_dwordx8/x16 into SGPRs, so a distance is exactly 16 VALU ops + 3 for the best-2 update,
// with no LDS traffic and no cross-lane reduction.  All waves of a workgroup stream the same
// train rows (K$/L2 hits).
// ---------------------------------------------------------------------------------------------
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
k_scan_lane_per_query(const ScanDesc* __restrict__ scans, const BlockDesc* __restrict__ blocks,
                      int32_t* __restrict__ zero, int nzero)
{
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nzero; i += BLOCK) g_(zero)[i] = 0;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const BlockDesc bd = blocks[wg];
    if (bd.item < 0) return;                       // padding entry of the XCD-striped table
    const ScanDesc sc = scans[bd.item];
    const int nq = sc.nq, nt = sc.nt;
    const int row = bd.row0 + (int)threadIdx.x;
    const int rrow = row < nq ? row : nq - 1;

    uint32_t q[8];
    {
        const auto qp = g_(reinterpret_cast<const u32x4*>(sc.q + (size_t)rrow * 32));
        const u32x4 a = qp[0], b = qp[1];
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w;
        q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    }

    uint32_t b0 = KEY_NONE, b1 = KEY_NONE;
    sptr_t tp = (sptr_t)(uintptr_t)sc.t;

    int j = 0;
    const int nt4 = nt & ~3;
    for (; j < nt4; j += 4) {
        sptr_t t = tp + (size_t)j * 8;
        const uint32_t d0 = PLSLAM_DIST8(q, t);
        const uint32_t d1 = PLSLAM_DIST8(q, t + 8);
        const uint32_t d2 = PLSLAM_DIST8(q, t + 16);
        const uint32_t d3 = PLSLAM_DIST8(q, t + 24);
        best2_push(b0, b1, make_key_s(d0, (uint32_t)j));
        best2_push(b0, b1, make_key_s(d1, (uint32_t)(j + 1)));
        best2_push(b0, b1, make_key_s(d2, (uint32_t)(j + 2)));
        best2_push(b0, b1, make_key_s(d3, (uint32_t)(j + 3)));
    }
    for (; j < nt; ++j) {
        sptr_t t = tp + (size_t)j * 8;
        const uint32_t d = PLSLAM_DIST8(q, t);
        best2_push(b0, b1, make_key_s(d, (uint32_t)j));
    }

    if (row < nq) g_(reinterpret_cast<gvec2_t*>(sc.keys))[row] = gvec2_t{b0, b1};
}

// ---------------------------------------------------------------------------------------------
// K1d  wave-per-query scan: the LATENCY variant for plans too small to give every SIMD a wave under
// K1a/K1b (one StVO::match call in the SLAM loop: 1500 queries are 6 workgroups there, 94 here).
// A 256-thread workgroup stages a tile of train rows in LDS (16-byte slots XOR-swizzled with bit 3
// of the row so that the per-lane 32-byte row reads are conflict-free ds_read_b128); each wave
// takes 4 queries into SGPRs (s_load_dwordx8), every lane scans the tile rows lane, lane+64, ...
// (19 VALU ops per distance, the train row is read once for the 4 queries), and a 6-step
// cross-lane butterfly merges the lanes' best-2 keys per query.  Same keys, same order as K1a.
// ---------------------------------------------------------------------------------------------
constexpr int WPQ_TILE_ROWS = 1536;                // 48 KB of LDS: 3 workgroups per CU
constexpr int WPQ_QUERIES_PER_WAVE = 4;
constexpr int WPQ_QUERIES_PER_BLOCK = 4 * WPQ_QUERIES_PER_WAVE;

__device__ __forceinline__ uint32_t wpq_slot(uint32_t row, uint32_t half)
{
    return ((2u * row + half) ^ ((row >> 3) & 1u));  // index of a 16-byte slot
}

__global__ void __launch_bounds__(256)
k_scan_wave_per_query(const ScanDesc* __restrict__ scans, const BlockDesc* __restrict__ blocks,
                      int32_t* __restrict__ zero, int nzero)
{
    __shared__ __attribute__((aligned(16))) uint4 tile[WPQ_TILE_ROWS * 2];

    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nzero; i += 256) g_(zero)[i] = 0;

    const BlockDesc bd = blocks[blockIdx.x];
    const ScanDesc sc = scans[bd.item];
    const int nq = sc.nq, nt = sc.nt;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q0 = bd.row0 + WPQ_QUERIES_PER_WAVE * wave;

    // this wave's 4 queries, wave-uniform -> SGPRs
    sptr_t qp = (sptr_t)(uintptr_t)sc.q;
    uint32_t q[WPQ_QUERIES_PER_WAVE][8];
#pragma unroll
    for (int k = 0; k < WPQ_QUERIES_PER_WAVE; ++k) {
        const int qi = q0 + k < nq ? q0 + k : nq - 1;
        sptr_t p = qp + (size_t)qi * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) q[k][i] = p[i];
    }
    uint32_t b[WPQ_QUERIES_PER_WAVE][2];
#pragma unroll
    for (int k = 0; k < WPQ_QUERIES_PER_WAVE; ++k) b[k][0] = b[k][1] = KEY_NONE;

    const auto tg = g_(reinterpret_cast<const u32x4*>(sc.t));
    for (int t0 = 0; t0 < nt; t0 += WPQ_TILE_ROWS) {
        const int rows = nt - t0 < WPQ_TILE_ROWS ? nt - t0 : WPQ_TILE_ROWS;
        if (t0) __syncthreads();                       // previous tile fully consumed
        for (int c = threadIdx.x; c < 2 * rows; c += 256)   // 16-byte chunks, coalesced
            { const u32x4 v = tg[(size_t)2 * t0 + c]; tile[wpq_slot((uint32_t)c >> 1, (uint32_t)c & 1u)] = make_uint4(v.x, v.y, v.z, v.w); }
        __syncthreads();
        for (int r = lane; r < rows; r += 64) {
            const uint4 ta = tile[wpq_slot((uint32_t)r, 0)], tb = tile[wpq_slot((uint32_t)r, 1)];
            const uint32_t t[8] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
            const uint32_t j = (uint32_t)(t0 + r);
#pragma unroll
            for (int k = 0; k < WPQ_QUERIES_PER_WAVE; ++k) {
                const uint32_t d = PLSLAM_DIST8(t, q[k]);
                best2_push(b[k][0], b[k][1], (d << KEY_IDX_BITS) | j);
            }
        }
    }
    // wave-level best-2 reduce (keys are unique per lane: distinct train indices)
#pragma unroll
    for (int k = 0; k < WPQ_QUERIES_PER_WAVE; ++k) {
        uint32_t k0 = b[k][0], k1 = b[k][1];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
            best2_merge(k0, k1, (uint32_t)__shfl_xor((int)k0, m), (uint32_t)__shfl_xor((int)k1, m));
        if (lane == 0 && q0 + k < nq) g_(reinterpret_cast<gvec2_t*>(sc.keys))[q0 + k] = gvec2_t{k0, k1};
    }
}

// ---------------------------------------------------------------------------------------------
// K1b  symmetric scan for mutual problems: ONE distance d(i,j) serves both directions.
//
// A wave owns 64 rows of `a` (one per lane, 8 VGPRs) and streams every row of `b` (SGPRs).
//   row side  (a -> b): lane-private best-2 on 32-bit keys (d<<23 | j), as in K1a.
//   column side (b -> a): the wave also drops d as u16 into a wave-private, transposed LDS tile
//     tile[jj][lane]; after 64 rows of b it re-reads the tile column-wise (lane = column jj, 128
//     contiguous bytes = the 64 distances of that column) and reduces them with PACKED 16-bit keys
//     (d << 6) | i_local  -- 10 distance bits + 6 index bits order exactly like (d, i) -- so one
//     v_pk_min/v_pk_max handles two candidates.  The result is the column's best-2 over this
//     wave's 64 rows: a partial, written to part21[iblk][j] and merged over iblk by K1c.
// Cost per (i,j): 8 xor + 8 bcnt + 3 (row update) + 2 (column update) VALU ops for TWO directed
// distances, against 2 x 19 for two directed scans.  LDS rows are 144 B (128 + 16 pad) so that
// the column-mode ds_read_b128 of 16 consecutive lanes hit 16 distinct 16-byte slots.
// No inter-wave communication: the four waves of a workgroup are independent (no barriers).
// ---------------------------------------------------------------------------------------------
constexpr int SYM_TILE_ROW_U16 = 72;                          // 144 bytes
constexpr int SYM_TILE_U16 = 64 * SYM_TILE_ROW_U16;           // 9216 bytes per wave

typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// packed keys of two candidates: ((d_hi << 6) | i_hi) << 16 | ((d_lo << 6) | i_lo);  w = d_hi<<16|d_lo.
// Rows past the end of `a` (ragged last block) carry d + 512 (see SYM_INVALID_BIAS), i.e. key16 >=
// 0x8000: they sort after every real candidate (real keys are <= (256 << 6) | 63 = 0x403F).
constexpr uint32_t SYM_INVALID_BIAS = 512;
__device__ __forceinline__ uint32_t pk_keys(uint32_t w, uint32_t ipair_uniform)
{
    uint32_t r;
    asm("v_lshl_or_b32 %0, %1, 6, %2" : "=v"(r) : "v"(w), "s"(ipair_uniform));
    return r;
}
__device__ __forceinline__ uint32_t key16_to_32(uint32_t c, uint32_t i_base)
{
    return c >= 0x8000u ? KEY_NONE : (((c >> 6) << KEY_IDX_BITS) | (i_base + (c & 63u)));
}

__device__ __forceinline__ void sym_column_reduce(const uint16_t* __restrict__ col, uint32_t& b0,
                                                  uint32_t& b1)
{
    // col: this lane's 64 u16 distances d(i, j), i = 0..63 (16-byte aligned)
    const uint4* cp = reinterpret_cast<const uint4*>(col);
    b0 = 0xFFFFFFFFu;
    b1 = 0xFFFFFFFFu;
#pragma unroll
    for (int v = 0; v < 8; ++v) {
        const uint4 w4 = cp[v];
        const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i_lo = 8 * v + 2 * u;
            const uint32_t key = pk_keys(w[u], (uint32_t)(((i_lo + 1) << 16) | i_lo));
            b1 = pk_min_u16(b1, pk_max_u16(b0, key));
            b0 = pk_min_u16(b0, key);
        }
    }
}

__global__ void __launch_bounds__(256)
k_scan_symmetric(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks,
                 int32_t* __restrict__ zero, int nzero)
{
    __shared__ __attribute__((aligned(16))) uint16_t lds[4 * SYM_TILE_U16];

    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nzero; i += 256) g_(zero)[i] = 0;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const BlockDesc bd = blocks[wg];
    if (bd.item < 0) return;                       // padding entry of the XCD-striped table
    const SymDesc sd = syms[bd.item];
    const int n1 = sd.n1, n2 = sd.n2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int i0 = bd.row0 + 64 * wave;           // first a-row of this wave
    if (i0 >= n1) return;                          // (no barriers in this kernel)
    const int iblk = i0 >> 6;
    const int row = i0 + lane;
    const uint32_t bias = row < n1 ? 0u : SYM_INVALID_BIAS;   // start of the popcount chain
    const int rrow = row < n1 ? row : n1 - 1;

    uint32_t q[8];
    {
        const auto qp = g_(reinterpret_cast<const u32x4*>(sd.a + (size_t)rrow * 32));
        const u32x4 a = qp[0], b = qp[1];
        q[0] = a.x; q[1] = a.y; q[2] = a.z; q[3] = a.w;
        q[4] = b.x; q[5] = b.y; q[6] = b.z; q[7] = b.w;
    }
    uint16_t* tile = lds + wave * SYM_TILE_U16;
    uint16_t* wr = tile + lane;                                  // tile[jj][lane]
    const uint16_t* col = tile + lane * SYM_TILE_ROW_U16;        // tile[lane][0..63]
    const auto part = g_(reinterpret_cast<gvec2_t*>(sd.part21)) + (size_t)iblk * n2;
    sptr_t tp = (sptr_t)(uintptr_t)sd.b;

    uint32_t rb0 = KEY_NONE, rb1 = KEY_NONE;
    for (int j0 = 0; j0 < n2; j0 += 64) {
        const int jc = n2 - j0 < 64 ? n2 - j0 : 64;
        // ---- row mode: stream jc rows of b ------------------------------------------------
        int jj = 0;
        for (; jj + 4 <= jc; jj += 4) {
            sptr_t t = tp + (size_t)(j0 + jj) * 8;
            const uint32_t d0 = PLSLAM_DIST8B(q, t, bias);
            const uint32_t d1 = PLSLAM_DIST8B(q, t + 8, bias);
            const uint32_t d2 = PLSLAM_DIST8B(q, t + 16, bias);
            const uint32_t d3 = PLSLAM_DIST8B(q, t + 24, bias);
            wr[(jj + 0) * SYM_TILE_ROW_U16] = (uint16_t)d0;
            wr[(jj + 1) * SYM_TILE_ROW_U16] = (uint16_t)d1;
            wr[(jj + 2) * SYM_TILE_ROW_U16] = (uint16_t)d2;
            wr[(jj + 3) * SYM_TILE_ROW_U16] = (uint16_t)d3;
            best2_push(rb0, rb1, make_key_s(d0, (uint32_t)(j0 + jj)));
            best2_push(rb0, rb1, make_key_s(d1, (uint32_t)(j0 + jj + 1)));
            best2_push(rb0, rb1, make_key_s(d2, (uint32_t)(j0 + jj + 2)));
            best2_push(rb0, rb1, make_key_s(d3, (uint32_t)(j0 + jj + 3)));
        }
        for (; jj < jc; ++jj) {
            sptr_t t = tp + (size_t)(j0 + jj) * 8;
            const uint32_t d = PLSLAM_DIST8B(q, t, bias);
            wr[jj * SYM_TILE_ROW_U16] = (uint16_t)d;
            best2_push(rb0, rb1, make_key_s(d, (uint32_t)(j0 + jj)));
        }
        // LDS operations of one wave execute in order; only the compiler must not reorder them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- column mode: lane = column j0 + lane -----------------------------------------
        uint32_t c0p, c1p;
        sym_column_reduce(col, c0p, c1p);
        {
            // two sorted streams (even i in the low halves, odd i in the high halves) -> best 2
            const uint32_t e0 = c0p & 0xFFFFu, o0 = c0p >> 16, e1 = c1p & 0xFFFFu, o1 = c1p >> 16;
            const uint32_t m0 = umin(e0, o0);
            const uint32_t m1 = umin(umax(e0, o0), umin(e1, o1));
            if (lane < jc)
                part[j0 + lane] = gvec2_t{key16_to_32(m0, (uint32_t)i0), key16_to_32(m1, (uint32_t)i0)};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (row < n1) g_(reinterpret_cast<gvec2_t*>(sd.keys12))[row] = gvec2_t{rb0, rb1};
}

// ---------------------------------------------------------------------------------------------
// K1b'  symmetric scan, FOUR rows of `a` per lane (sym_rows = 4; chosen automatically for large plans).  Any VALU op with an SGPR source
// issues at the slow rate on gfx950 (4.1 vs 2.4 cycles, profiles/r1_valu_microbench.txt), and the
// train row lives in SGPRs.  With rows q0..q3 in one lane,  q_r ^ t = (q0 ^ t) ^ (q0 ^ q_r):  only
// the first XOR touches the SGPRs, the other three use a per-lane constant c_r = q0 ^ q_r and run
// at the fast VGPR-only rate.  A wave (= a 64-thread workgroup, no barriers) owns 256 rows of `a`:
// lane l holds rows i0 + 64 r + l.  The transposed LDS tile keeps its 9216 bytes: 16 rows of `b`
// x 4 row-blocks x 144 B.  In column mode lane l reduces row-block (l >> 4) of column (l & 15)
// with the packed 16-bit keys, the four row-blocks are combined across lanes, and ONE partial per
// 256 rows of `a` is written (4x fewer partials than K1b).
// MEASURED (profiles/r1_valu_microbench.txt, "alt xor(v,v)/bcnt"): a fast op alternating with a slow
// one issues at nearly the slow rate, so the VALU gain is small; what pays is the 4x smaller partial
// table and merge: 364k vs 347k pairs/s at 4096 pairs/step, 347k vs 344k at 2048 -- but 266k vs 320k
// at 512, where the 4x coarser work units lose to tail quantisation.  The plan picks (capi.hip).
// ---------------------------------------------------------------------------------------------
constexpr int SYM4_SUBTILE_U16 = 16 * SYM_TILE_ROW_U16;      // 16 b-rows x 144 B = 2304 B per row-block

__global__ void __launch_bounds__(64)
k_scan_symmetric_r4(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks,
                    int32_t* __restrict__ zero, int nzero)
{
    __shared__ __attribute__((aligned(16))) uint16_t tile[4 * SYM4_SUBTILE_U16];

    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nzero; i += 64) g_(zero)[i] = 0;

    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const BlockDesc bd = blocks[wg];
    if (bd.item < 0) return;                       // padding entry of the XCD-striped table
    const SymDesc sd = syms[bd.item];
    const int n1 = sd.n1, n2 = sd.n2;
    const int lane = threadIdx.x;
    const int i0 = bd.row0;                        // first of this wave's 256 a-rows
    const int iblk = i0 >> 8;

    uint32_t q0[8], c1[8], c2[8], c3[8];
    uint32_t bias[4];
    {
        uint32_t qr[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + 64 * r + lane;
            bias[r] = row < n1 ? 0u : SYM_INVALID_BIAS;
            const int rrow = row < n1 ? row : n1 - 1;
            const auto qp = g_(reinterpret_cast<const u32x4*>(sd.a + (size_t)rrow * 32));
            const u32x4 a = qp[0], b = qp[1];
            qr[r][0] = a.x; qr[r][1] = a.y; qr[r][2] = a.z; qr[r][3] = a.w;
            qr[r][4] = b.x; qr[r][5] = b.y; qr[r][6] = b.z; qr[r][7] = b.w;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            q0[k] = qr[0][k];
            c1[k] = qr[0][k] ^ qr[1][k];
            c2[k] = qr[0][k] ^ qr[2][k];
            c3[k] = qr[0][k] ^ qr[3][k];
        }
    }
    uint16_t* wr = tile + lane;                                          // tile[r][jj][lane]
    const uint16_t* col = tile + (lane >> 4) * SYM4_SUBTILE_U16 + (lane & 15) * SYM_TILE_ROW_U16;
    const uint32_t col_i_base = (uint32_t)(i0 + 64 * (lane >> 4));
    const auto part = g_(reinterpret_cast<gvec2_t*>(sd.part21)) + (size_t)iblk * n2;
    sptr_t tp = (sptr_t)(uintptr_t)sd.b;

    uint32_t rb[4][2];
#pragma unroll
    for (int r = 0; r < 4; ++r) rb[r][0] = rb[r][1] = KEY_NONE;

#define PLSLAM_SYM4_ROW(T, JJ)                                                                    \
    {                                                                                             \
        uint32_t x[8];                                                                            \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) x[k] = q0[k] ^ (T)[k];                      \
        const uint32_t e0 = bcnt_acc(x[7], bcnt_acc(x[6], bcnt_acc(x[5], bcnt_acc(x[4],           \
                            bcnt_acc(x[3], bcnt_acc(x[2], bcnt_acc(x[1], bcnt_acc(x[0], bias[0]))))))));  \
        const uint32_t e1 = bcnt_acc(x[7] ^ c1[7], bcnt_acc(x[6] ^ c1[6], bcnt_acc(x[5] ^ c1[5],  \
                            bcnt_acc(x[4] ^ c1[4], bcnt_acc(x[3] ^ c1[3], bcnt_acc(x[2] ^ c1[2],  \
                            bcnt_acc(x[1] ^ c1[1], bcnt_acc(x[0] ^ c1[0], bias[1]))))))));        \
        const uint32_t e2 = bcnt_acc(x[7] ^ c2[7], bcnt_acc(x[6] ^ c2[6], bcnt_acc(x[5] ^ c2[5],  \
                            bcnt_acc(x[4] ^ c2[4], bcnt_acc(x[3] ^ c2[3], bcnt_acc(x[2] ^ c2[2],  \
                            bcnt_acc(x[1] ^ c2[1], bcnt_acc(x[0] ^ c2[0], bias[2]))))))));        \
        const uint32_t e3 = bcnt_acc(x[7] ^ c3[7], bcnt_acc(x[6] ^ c3[6], bcnt_acc(x[5] ^ c3[5],  \
                            bcnt_acc(x[4] ^ c3[4], bcnt_acc(x[3] ^ c3[3], bcnt_acc(x[2] ^ c3[2],  \
                            bcnt_acc(x[1] ^ c3[1], bcnt_acc(x[0] ^ c3[0], bias[3]))))))));        \
        wr[0 * SYM4_SUBTILE_U16 + (JJ) * SYM_TILE_ROW_U16] = (uint16_t)e0;                        \
        wr[1 * SYM4_SUBTILE_U16 + (JJ) * SYM_TILE_ROW_U16] = (uint16_t)e1;                        \
        wr[2 * SYM4_SUBTILE_U16 + (JJ) * SYM_TILE_ROW_U16] = (uint16_t)e2;                        \
        wr[3 * SYM4_SUBTILE_U16 + (JJ) * SYM_TILE_ROW_U16] = (uint16_t)e3;                        \
        const uint32_t jkey = (uint32_t)(j0 + (JJ));                                              \
        best2_push(rb[0][0], rb[0][1], make_key_s(e0, jkey));                                     \
        best2_push(rb[1][0], rb[1][1], make_key_s(e1, jkey));                                     \
        best2_push(rb[2][0], rb[2][1], make_key_s(e2, jkey));                                     \
        best2_push(rb[3][0], rb[3][1], make_key_s(e3, jkey));                                     \
    }

    for (int j0 = 0; j0 < n2; j0 += 16) {
        const int jc = n2 - j0 < 16 ? n2 - j0 : 16;
        // ---- row mode --------------------------------------------------------------------------
        int jj = 0;
        for (; jj + 4 <= jc; jj += 4) {
            sptr_t t = tp + (size_t)(j0 + jj) * 8;
            PLSLAM_SYM4_ROW(t, jj + 0)
            PLSLAM_SYM4_ROW(t + 8, jj + 1)
            PLSLAM_SYM4_ROW(t + 16, jj + 2)
            PLSLAM_SYM4_ROW(t + 24, jj + 3)
        }
        for (; jj < jc; ++jj) {
            sptr_t t = tp + (size_t)(j0 + jj) * 8;
            PLSLAM_SYM4_ROW(t, jj)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- column mode: lane = (column j0 + (lane & 15), row-block lane >> 4) -----------------
        uint32_t c0p, c1p;
        sym_column_reduce(col, c0p, c1p);
        const uint32_t ev0 = c0p & 0xFFFFu, od0 = c0p >> 16, ev1 = c1p & 0xFFFFu, od1 = c1p >> 16;
        uint32_t k0 = key16_to_32(umin(ev0, od0), col_i_base);
        uint32_t k1 = key16_to_32(umin(umax(ev0, od0), umin(ev1, od1)), col_i_base);
        // combine the four row-blocks of a column: lanes l, l^16, l^32, l^48
        best2_merge(k0, k1, (uint32_t)__shfl_xor((int)k0, 16), (uint32_t)__shfl_xor((int)k1, 16));
        best2_merge(k0, k1, (uint32_t)__shfl_xor((int)k0, 32), (uint32_t)__shfl_xor((int)k1, 32));
        if (lane < jc) part[j0 + lane] = gvec2_t{k0, k1};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
#undef PLSLAM_SYM4_ROW
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + 64 * r + lane;
        if (row < n1) g_(reinterpret_cast<gvec2_t*>(sd.keys12))[row] = gvec2_t{rb[r][0], rb[r][1]};
    }
}

// K1c  merge the per-block column partials of K1b: keys21[j] = best-2 over iblk of part21[iblk][j]
__global__ void __launch_bounds__(256)
k_merge_partials(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks)
{
    const BlockDesc bd = blocks[blockIdx.x];
    const SymDesc sd = syms[bd.item];
    const int j = bd.row0 + (int)threadIdx.x;
    if (j >= sd.n2) return;
    const auto part = g_(reinterpret_cast<const gvec2_t*>(sd.part21));
    uint32_t b0 = KEY_NONE, b1 = KEY_NONE;
    for (int ib = 0; ib < sd.n_iblk; ++ib) {
        const gvec2_t p = part[(size_t)ib * sd.n2 + j];
        best2_merge(b0, b1, p.x, p.y);
    }
    g_(reinterpret_cast<gvec2_t*>(sd.keys21))[j] = gvec2_t{b0, b1};
}

// ---------------------------------------------------------------------------------------------
// K2  finalize: ratio test (fp32, one multiply) + mutual consistency -> matches_12, #matches.
// stvo-pl matchNNR: accept iff (float)d0 < (float)d1 * nnr; match(): keep i1->i2 iff m21[i2]==i1.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int ratio_pick(uint32_t k0, uint32_t k1, float nnr)
{
    if (k1 == KEY_NONE) return -1;  // fewer than two neighbours: defined as "no match"
    const float d0 = (float)(k0 >> KEY_IDX_BITS);
    const float d1n = __fmul_rn((float)(k1 >> KEY_IDX_BITS), nnr);
    return d0 < d1n ? (int)(k0 & KEY_IDX_MASK) : -1;
}

#ifndef PLSLAM_NT_FINALIZE
#define PLSLAM_NT_FINALIZE 1
#endif
// One row of a problem: ratio test on its scan result, the mutual check against the candidate column's pair kb = fetch_kb(m)
// (K1h / K1i plans: completed lazily, see below), the table entry, the count and the stereo gate.  EVERY lane of the workgroup
// calls it (DPP row rotations inside); the workgroup's 256 lanes are 256 consecutive rows starting at a multiple of 16.
// `pre`: the row's pair of keys12 if the caller has loaded it already (problems that are not column-split), else nullptr.
template <class FetchKb>
__device__ __forceinline__ void finalize_row(const ProblemDesc& p, const int i1, const plslam_stereo_gate_problem* __restrict__ gates,
                                             FetchKb fetch_kb, const gvec2_t* pre = nullptr)
{
    int m = -1;
    bool accepted = false, cleared = false;
    uint2 k = make_uint2(KEY_NONE, KEY_NONE);            // this row's scan result (rows past n1: none)
    uint2 kb = make_uint2(KEY_NONE, KEY_NONE);           // the candidate column's keys21 pair
    bool check = false;                                  // this row has a candidate whose consistency must be checked
    if (i1 < p.n1) {
        if (p.nsplit > 1) {
            // column-split problem: best-2 over the ranges' row results; keys are (d << 23 | j) with j relative to the
            // range, so its first column is added first -- (d, j) order holds within and across ranges
            const auto tmp = g_(reinterpret_cast<const gvec2_t*>(p.split_tmp));
            uint32_t b0 = KEY_NONE, b1 = KEY_NONE;
            for (int s = 0; s < p.nsplit; ++s) {
                const gvec2_t q = tmp[(size_t)s * p.n1 + i1];
                const uint32_t off = (uint32_t)(s * p.cstep);
                best2_merge(b0, b1, q.x == KEY_NONE ? KEY_NONE : q.x + off, q.y == KEY_NONE ? KEY_NONE : q.y + off);
            }
            k = make_uint2(b0, b1);
            g_(reinterpret_cast<gvec2_t*>(p.keys12_out))[i1] = gvec2_t{k.x, k.y};          // diagnostics (plslam_match_plan_dump)
        } else if (pre) {
            k = make_uint2(pre->x, pre->y);
        } else {
            // (read once: non-temporal, like the table written below -- they must not push the scan's rows out of L2)
            const gvec2_t kv = PLSLAM_NT_FINALIZE ? __builtin_nontemporal_load(g_(reinterpret_cast<const gvec2_t*>(p.keys12)) + i1)
                                                  : g_(reinterpret_cast<const gvec2_t*>(p.keys12))[i1];
            k = make_uint2(kv.x, kv.y);
        }
        m = ratio_pick(k.x, k.y, p.nnr);
        accepted = m >= 0;
        // [RECALL] stvo-pl matchNNR resize()s the table it is given: an entry already there survives a rejected row, then
        // goes through the consistency loop like any other (an entry outside [0, n2) -- an out-of-bounds read upstream --
        // is defined as failing it)
        if (!accepted && p.keep_prior) m = g_(p.matches_12)[i1];
        if (m >= 0 && p.mutual) {
            check = m < p.n2;
            if (check) kb = fetch_kb(m);
            else { m = -1; cleared = true; }
        }
    }
    bool ok = true;
    if (!p.lazy21) {
        if (check) ok = ratio_pick(kb.x, kb.y, p.nnr) == i1;
    } else {
        // K1h plans.  kb.x = column m's best row (exact); kb.y = its best row OUTSIDE kb.x's aligned group of 16 rows of d1: an
        // upper bound of the true second best.  The consistency check can only hold if the best row is THIS row; then the
        // other 15 rows of this row's group are what kb.y has not seen.  Row r's own scan result bounds its distance to
        // column m from below: it IS r's best distance if m is r's best column, and it is at least r's second-best distance
        // otherwise.  Those results sit in the neighbouring lanes (a group = 16 aligned lanes: DPP row rotation).  With the
        // lower bound lo = min(kb.y's distance, the 15 bounds) and the upper bound hi = kb.y's distance on the same side of
        // the ratio threshold the test is decided; only between them are the 15 distances recomputed (XOR + popcount).
        const uint32_t mm = check ? (uint32_t)m : 0xFFFFFFFFu;
        uint32_t lb = 0xFFFFFFFFu;                       // min over the other rows of the group of their bound (a distance)
#define PLSLAM_FIN_ROT(S)                                                                           \
        {                                                                                          \
            const uint32_t px = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k.x, 0x120 + (S), 0xf, 0xf, false); \
            const uint32_t py = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)k.y, 0x120 + (S), 0xf, 0xf, false); \
            const uint32_t b = (px != KEY_NONE && (px & KEY_IDX_MASK) == mm) ? (px >> KEY_IDX_BITS)  \
                                                                           : (py == KEY_NONE ? 0xFFFFFFFFu : (py >> KEY_IDX_BITS)); \
            lb = lb < b ? lb : b;                                                                  \
        }
        PLSLAM_FIN_ROT(1) PLSLAM_FIN_ROT(2) PLSLAM_FIN_ROT(3) PLSLAM_FIN_ROT(4) PLSLAM_FIN_ROT(5)
        PLSLAM_FIN_ROT(6) PLSLAM_FIN_ROT(7) PLSLAM_FIN_ROT(8) PLSLAM_FIN_ROT(9) PLSLAM_FIN_ROT(10)
        PLSLAM_FIN_ROT(11) PLSLAM_FIN_ROT(12) PLSLAM_FIN_ROT(13) PLSLAM_FIN_ROT(14) PLSLAM_FIN_ROT(15)
#undef PLSLAM_FIN_ROT
        if (check) {
            ok = kb.x != KEY_NONE && (int)(kb.x & KEY_IDX_MASK) == i1;
            if (ok) {
                const float d0 = (float)(kb.x >> KEY_IDX_BITS);
                const uint32_t hi = kb.y == KEY_NONE ? 0xFFFFFFFFu : (kb.y >> KEY_IDX_BITS);
                const uint32_t lo = hi < lb ? hi : lb;
                if (hi != 0xFFFFFFFFu && !(d0 < __fmul_rn((float)hi, p.nnr))) {
                    ok = false;                                      // fails even against the upper bound
                } else if (lo != 0xFFFFFFFFu && d0 < __fmul_rn((float)lo, p.nnr)) {
                    ok = true;                                       // holds even against the lower bound (a second row exists)
                } else {
                    uint32_t second = kb.y;
                    const auto b = g_(reinterpret_cast<const u32x4*>(p.d2 + (size_t)m * 32));
                    const u32x4 b_lo = b[0], b_hi = b[1];
                    const int base = i1 & ~15;
                    for (int q = 0; q < 16; ++q) {
                        const int i = base + q;
                        const bool use = i != i1 && i < p.n1;
                        const auto a = g_(reinterpret_cast<const u32x4*>(p.d1 + (size_t)(use ? i : i1) * 32));
                        const u32x4 a_lo = a[0], a_hi = a[1];
                        const uint32_t d = __popc(a_lo.x ^ b_lo.x) + __popc(a_lo.y ^ b_lo.y) + __popc(a_lo.z ^ b_lo.z) +
                                           __popc(a_lo.w ^ b_lo.w) + __popc(a_hi.x ^ b_hi.x) + __popc(a_hi.y ^ b_hi.y) +
                                           __popc(a_hi.z ^ b_hi.z) + __popc(a_hi.w ^ b_hi.w);
                        const uint32_t cand = use ? ((d << KEY_IDX_BITS) | (uint32_t)i) : KEY_NONE;
                        second = second < cand ? second : cand;
                    }
                    ok = ratio_pick(kb.x, second, p.nnr) == i1;
                }
            }
        }
    }
    if (check && !ok) { m = -1; cleared = true; }
    if (i1 < p.n1) {
        if (PLSLAM_NT_FINALIZE) __builtin_nontemporal_store(m, g_(p.matches_12) + i1);
        else g_(p.matches_12)[i1] = m;
        // the gather's wire table (plslam_match_plan_set_wire16): the same entry as int16 (m < n2 <= 32768)
        if (p.matches_16) g_(p.matches_16)[i1] = (int16_t)m;
    }
    if (p.n_matches) {
        // the reference's arithmetic: +1 per row the ratio test accepts, -1 per entry the consistency loop clears (equal
        // to the number of entries >= 0 unless kept entries are involved)
        const int delta = (int)__popcll(__ballot(accepted)) - (int)__popcll(__ballot(cleared));
        if ((threadIdx.x & 63) == 0 && delta) (void)atomic_add_global(p.n_matches, delta);
    }
    if (gates && p.gate >= 0) {           // (a plan without a gate stage passes no table: never read through a stale index)
        // StereoFrame's gate over this L<->R table (stereo_gates_dev.hpp), on the entry just decided.  (Requesting the gate's
        // descriptor and the row's left feature ahead of the keys -- two links off the chain of dependent accesses -- measured
        // 2.5 % SLOWER for the stages behind the scan, 0.298 against 0.290 ms per 4096-pair step.)
        const plslam_stereo_gate_problem q = gates[p.gate];
        const int kept = i1 < p.n1 ? stereo_gate_row(q, i1, m) : 0;
        const unsigned long long bal = __ballot(kept);
        if (q.n_stereo && (threadIdx.x & 63) == 0 && bal) (void)atomic_add_global(q.n_stereo, (int)__popcll(bal));
    }
}

__global__ void __launch_bounds__(256)
k_finalize(const ProblemDesc* __restrict__ probs, const BlockDesc* __restrict__ blocks,
           const plslam_stereo_gate_problem* __restrict__ gates, int nblocks, int per_xcd)
{
  // per_xcd > 0: workgroup b runs on XCD b % 8 and takes table entry (b % 8) * per_xcd + b / 8, so that the (<= 6) row blocks of
  // one problem gather their column pairs keys21[m] through ONE L2 -- in table order they sit on six XCDs and each fetches
  // the problem's whole column table (round-4 counters: FETCH_SIZE 488 k KiB per 4096-pair step against 174 k; tools/
  // fetch_gather_calib.hip reproduces the pattern and the factor).  Option "post_xcd" 2 (default): plan_build deals the table
  // to the XCDs problem by problem (rows of per_xcd entries, padding item = -1) -- consecutive problems on different XCDs, the
  // eight sweep memory together: step time equal to 1 % better, the stage alone 2 % slower.  Option 1: per_xcd = ceil(nblocks /
  // 8) over the table in problem order, i.e. a contiguous eighth per XCD: the same bytes but 8 % SLOWER for the stage (0.302
  // against 0.280 ms per step) -- eight distant regions of memory at a time.
  // (a capped grid walks the block table: plslam_ctx option "post_workgroups")
  const int nslots = per_xcd > 0 ? 8 * per_xcd : nblocks;
  for (int b = blockIdx.x; b < nslots; b += gridDim.x) {
    const int blk = per_xcd > 0 ? (b & 7) * per_xcd + (b >> 3) : b;
    if (blk >= nblocks) continue;                      // (the whole workgroup: finalize_row's DPP rotations see all lanes or none)
    const BlockDesc bd = blocks[blk];
    if (bd.item < 0) continue;                         // padding entry of a table dealt to the XCDs (option "post_xcd" 2)
    const ProblemDesc p = probs[bd.item];
    finalize_row(p, bd.row0 + (int)threadIdx.x, gates, [&](int m) {
        const gvec2_t kv = g_(reinterpret_cast<const gvec2_t*>(p.keys21))[m];
        return make_uint2(kv.x, kv.y);
    });
  }
}

// K2'  everything behind a K1h / K1i scan in ONE kernel, one workgroup per problem: the column partials of the problem's <= 16
// row blocks are merged into LDS (what k_merge_fix16<1, false> writes to keys21: best row, distance of the best row outside
// its group of 16), then the problem's rows are finalized from there (+ gates).  No merged column table in HBM: the separate
// kernels write it once (8 B per column) and gather it back through up to six L2s with 8-byte reads (round 3 PMC: 1.0 GB
// fetched to write 0.16 GB).  Throughput plans only (plan_build: every problem mutual, on K1h / K1i, n2 <= POST_FUSED_MAX_N2,
// lazy keys): a plan of a few large problems has too few workgroups for this.
template <int NT>
__global__ void __launch_bounds__(NT)
k_post_fused(const ProblemDesc* __restrict__ probs, const plslam_stereo_gate_problem* __restrict__ gates, int nprob)
{
    extern __shared__ __attribute__((aligned(16))) uint2 kb_lds[];           // [n2]: the merged pair of every column
    for (int pi = blockIdx.x; pi < nprob; pi += gridDim.x) {
        const ProblemDesc p = probs[pi];
        const MhLayout L(p.n2);
        const int nwb = (p.n1 + 255) >> 8, n2p = L.slots_padded(), nslots = 32 * L.ntiles;
        const auto part = g_(p.part21);
        if (pi != (int)blockIdx.x) __syncthreads();        // the problem before: every lane is past its reads of kb_lds
        // the first 256 rows of keys12 now: their latency passes under the merge
        const auto k12 = g_(reinterpret_cast<const gvec2_t*>(p.keys12));
        gvec2_t kcur = gvec2_t{KEY_NONE, KEY_NONE};
        if ((int)threadIdx.x < p.n1) kcur = __builtin_nontemporal_load(k12 + threadIdx.x);
        // merge: a lane takes slots 256 apart, PF at a time (their nwb words each are requested together: one round trip)
        constexpr int PF = NT >= 1024 ? 2 : 4;
        for (int s0_ = (int)threadIdx.x; s0_ < nslots; s0_ += NT * PF) {
            uint32_t b0[PF], b1[PF], sx[PF];
            int j[PF];
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                const int slot = s0_ + NT * q;
                j[q] = slot < nslots ? L.row_of(slot >> 5, slot & 31) : p.n2;
                b0[q] = b1[q] = KEY_NONE;
                sx[q] = 0xFFFFFFFFu;
            }
            for (int wb = 0; wb < nwb; ++wb) {
                uint32_t e[PF];
#pragma unroll
                for (int q = 0; q < PF; ++q) e[q] = j[q] < p.n2 ? __builtin_nontemporal_load(&part[(size_t)wb * n2p + s0_ + NT * q]) : 0xFFFFFFFFu;
#pragma unroll
                for (int q = 0; q < PF; ++q) {
                    const uint32_t e0 = e[q] >> 9, e1 = ((e[q] & 511u) << 8) | 255u;     // (d0 << 8 | row0), the second entry's distance
                    const uint32_t k = ((e0 >> 8) << KEY_IDX_BITS) | ((e0 & 255u) + 256u * (uint32_t)wb);
                    sx[q] = k < b0[q] ? e1 : sx[q];
                    b1[q] = umin(b1[q], umax(b0[q], k));
                    b0[q] = umin(b0[q], k);
                }
            }
#pragma unroll
            for (int q = 0; q < PF; ++q) {
                if (j[q] >= p.n2) continue;                // the slot holds no column
                if (b0[q] < (257u << KEY_IDX_BITS)) {
                    b1[q] = umin(b1[q], ((sx[q] >> 8) << KEY_IDX_BITS) | KEY_IDX_MASK);
                    if (b1[q] >= (257u << KEY_IDX_BITS)) b1[q] = KEY_NONE;
                } else {
                    b0[q] = b1[q] = KEY_NONE;
                }
                kb_lds[j[q]] = make_uint2(b0[q], b1[q]);
            }
        }
        __syncthreads();
        // rows: the next block's keys are requested before this block's are consumed
#pragma unroll 1
        for (int r0 = 0; r0 < p.n1; r0 += NT) {
            const int inext = r0 + NT + (int)threadIdx.x;
            gvec2_t knext = gvec2_t{KEY_NONE, KEY_NONE};
            if (inext < p.n1) knext = __builtin_nontemporal_load(k12 + inext);
            finalize_row(p, r0 + (int)threadIdx.x, gates, [&](int m) { return kb_lds[m]; }, &kcur);
            kcur = knext;
        }
    }
}

// copies the per-problem counters to caller pointers that are not one contiguous array
__global__ void __launch_bounds__(256)
k_scatter_counts(const int32_t* __restrict__ src, int32_t* const* __restrict__ dst, int32_t n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n && g_(dst)[i]) *g_(g_(dst)[i]) = g_(src)[i];
}

// keys -> (idx, dist) pairs of the knnMatch ABI
__global__ void __launch_bounds__(256)
k_unpack_keys(const uint32_t* __restrict__ keys, int32_t n, int32_t* __restrict__ idx,
              int32_t* __restrict__ dist)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = g_(keys)[i];
    idx[i] = k == KEY_NONE ? -1 : (int32_t)(k & KEY_IDX_MASK);
    dist[i] = k == KEY_NONE ? INT32_MAX : (int32_t)(k >> KEY_IDX_BITS);
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
int scan_rows_per_block(int variant, int block_threads)
{
    return variant == PLSLAM_SCAN_WAVE_PER_QUERY ? WPQ_QUERIES_PER_BLOCK : block_threads;
}

int launch_scan(const plslam_ctx* ctx, int variant, int block_threads, const ScanDesc* d_scans,
                const BlockDesc* d_blocks, int nblocks, int32_t* d_zero, int nzero, hipStream_t s)
{
    (void)ctx;
    if (nblocks <= 0) {
        if (nzero > 0) PLSLAM_HIP_CHECK(hipMemsetAsync(d_zero, 0, sizeof(int32_t) * nzero, s));
        return PLSLAM_OK;
    }
    if (variant == PLSLAM_SCAN_WAVE_PER_QUERY) {
        hipLaunchKernelGGL(k_scan_wave_per_query, dim3(nblocks), dim3(256), 0, s, d_scans, d_blocks, d_zero,
                           nzero);
        PLSLAM_HIP_CHECK(hipGetLastError());
        return PLSLAM_OK;
    }
    PLSLAM_REQUIRE(variant == PLSLAM_SCAN_LANE_PER_QUERY, PLSLAM_EINVAL);
    switch (block_threads) {
        case 256:
            hipLaunchKernelGGL(k_scan_lane_per_query<256>, dim3(nblocks), dim3(256), 0, s, d_scans,
                               d_blocks, d_zero, nzero);
            break;
        case 512:
            hipLaunchKernelGGL(k_scan_lane_per_query<512>, dim3(nblocks), dim3(512), 0, s, d_scans,
                               d_blocks, d_zero, nzero);
            break;
        case 1024:
            hipLaunchKernelGGL(k_scan_lane_per_query<1024>, dim3(nblocks), dim3(1024), 0, s, d_scans,
                               d_blocks, d_zero, nzero);
            break;
        default:
            PLSLAM_REQUIRE(!"scan_block must be 256, 512 or 1024", PLSLAM_EINVAL);
    }
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int sym_rows_per_block(int rows_per_lane) { return 256; (void)rows_per_lane; }
int sym_rows_per_partial(int rows_per_lane) { return rows_per_lane == 4 ? 256 : 64; }

int launch_scan_sym(int rows_per_lane, const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks,
                    int32_t* d_zero, int nzero, hipStream_t s)
{
    if (nblocks <= 0) return PLSLAM_OK;
    if (rows_per_lane == 4)
        hipLaunchKernelGGL(k_scan_symmetric_r4, dim3(nblocks), dim3(64), 0, s, d_sym, d_blocks, d_zero, nzero);
    else
        hipLaunchKernelGGL(k_scan_symmetric, dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks, d_zero, nzero);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int launch_merge_partials(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, hipStream_t s)
{
    if (nblocks <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_merge_partials, dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

#ifndef PLSLAM_POST_FUSED_THREADS
#define PLSLAM_POST_FUSED_THREADS 256
#endif
constexpr int POST_FUSED_THREADS = PLSLAM_POST_FUSED_THREADS;
int launch_post_fused(const ProblemDesc* d_probs, int nprob, const plslam_stereo_gate_problem* d_gates, size_t lds_bytes,
                      hipStream_t s)
{
    if (nprob <= 0) return PLSLAM_OK;
    // (measured per 16 384-problem step at C2, exclusive: this kernel with 256 lanes per problem 0.335 ms, with 1024 lanes
    // 0.65 ms, the separate merge + finalize kernels 0.276 ms -- see capi.hip plan_build, option "post_fuse")
    hipLaunchKernelGGL(k_post_fused<POST_FUSED_THREADS>, dim3(nprob), dim3(POST_FUSED_THREADS), lds_bytes, s, d_probs, d_gates, nprob);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int launch_finalize(const ProblemDesc* d_probs, const BlockDesc* d_blocks, int nblocks,
                    const plslam_stereo_gate_problem* d_gates, hipStream_t s, int grid_cap, bool xcd_chunks, int dealt_row)
{
    if (nblocks <= 0) return PLSLAM_OK;
    // xcd_chunks: contiguous table entries per XCD (k_finalize; option post_xcd 1, measured slower); pointless below a few entries
    // per XCD.  dealt_row > 0: the table is already dealt to the XCDs in 8 rows of that length (plan_build, option post_xcd 2).
    const int per_xcd = dealt_row > 0 ? dealt_row : (xcd_chunks && nblocks >= 64 ? (nblocks + 7) / 8 : 0);
    const int nslots = per_xcd > 0 ? 8 * per_xcd : nblocks;
    int grid = grid_cap > 0 && grid_cap < nslots ? grid_cap : nslots;
    if (per_xcd > 0 && grid < nslots) grid = grid >= 8 ? grid & ~7 : 8;   // a walking workgroup stays on its XCD's chunk
    hipLaunchKernelGGL(k_finalize, dim3(grid), dim3(256), 0, s, d_probs, d_blocks, d_gates, nblocks, per_xcd);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int launch_scatter_counts(const int32_t* d_src, int32_t* const* d_dst, int32_t n, hipStream_t s)
{
    if (n <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_scatter_counts, dim3((n + 255) / 256), dim3(256), 0, s, d_src, d_dst, n);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int launch_unpack_keys(const uint32_t* d_keys, int32_t n, int32_t* d_idx, int32_t* d_dist,
                       hipStream_t s)
{
    if (n <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_unpack_keys, dim3((n + 255) / 256), dim3(256), 0, s, d_keys, n, d_idx,
                       d_dist);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

}  // namespace plslam
