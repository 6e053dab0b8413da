// hamming_mfma_g.hip -- K1f: the matrix-core symmetric Hamming kNN-2 scan with GROUPED row bookkeeping (gfx950).
//
// Contract, work decomposition, block tables, partial table, merge + finalize kernels: exactly those of K1e
// (hamming_mfma.hip) -- keys12[i] = best-2 over j, part21[i-block][j] = best-2 over the block's i, keys =
// (distance << 23) | index = cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) order in both directions (reference call
// sites src/mapHandler.cpp:277,424,597,712,3223,3249).  The distances come out of the same four
// v_mfma_scale_f32_32x32x64_f8f6f4 per 32x32 tile (fp4 codes of +-1, accumulator = 2^23 + 128 d + tag, exact).
//
// What is different: K1e sits on the VALU issue limit of its best-2 bookkeeping (round 1: 455 VALU instructions
// per two tiles, matrix pipe 24 % busy), and three things take instructions out of that stream here.
//
//  1. Row direction: minimum now, second best later.  K1e pushes every packed key pair into a sorted pair
//     (best, second) per (row, column class): 3 packed ops per 2 distances.  Here a lane keeps only the running
//     MINIMUM of a group of 16 consecutive tiles per (row, class) -- ONE v_pk_min_u16 per 2 distances -- and pushes
//     the group minimum into the sorted pair once per 16 tiles.  Best-2 over group minima gives the exact best key
//     B0, and B1 = the best key outside B0's group; the true second best is min(B1, best key among the OTHER
//     members of B0's group).  Those are 15 known columns of the same class (j0 +- 32 k), so after the scan one lane
//     per row recomputes 15 distances with XOR + popcount from the raw rows (L2-resident: the workgroup has just
//     streamed them) and takes the minimum.  Exact, tie order included: within a class the 16-bit key order
//     (d, tile) is the (d, j) order.  Per tile 16 + 64/16 = 20 instead of 48 packed ops; per scan ~350 extra.
//  2. The tile number of a key rides in the accumulator start value (scalar adds on the seeds) instead of one
//     vector add per packed pair; the column keys of a tile then share the offset and finish_columns takes it
//     off again (round 1's tile-in-seed experiment).
//  3. The byte -> fp4 expansion is arithmetic: nibble k of output dword s = 0x2 | bit(4k + s) << 3, i.e.
//     ((x << (3 - s)) & 0x88888888) | 0x22222222: 3 adds + 4 v_and_or per raw dword, no 256-entry table, no
//     data-dependent LDS gathers (round 1 measured 1.9e8 bank-conflict cycles per launch in those), one barrier
//     less at kernel start.  Only the lane -> k map shared by the A and the B operand matters for a contraction
//     over all k, so the changed bit order changes nothing.
//
// The column direction is unchanged (exact best-2 per tile): a column's candidates are complete within the
// tile, so a deferred second best would have to be recomputed per tile and row block, which costs what it saves.
#include "common.hpp"

#include <type_traits>

// build-time experiments for tools/scan_time.py / tools/build_exp.py (results are WRONG with any of them on), a bit mask:
//   1 no workgroup barrier   2 no bookkeeping rows   4 no MFMA   8 no column store   16 no second-best fix-up
//   32 no finish_columns     64 no group push        128 no expansion of the b tile (no global load, no LDS write)
//   256 no operand reads from LDS                    512 no pack
#ifndef PLSLAM_MG_EXPERIMENT
#define PLSLAM_MG_EXPERIMENT 0
#endif
#define PLSLAM_MG_X(bit) ((PLSLAM_MG_EXPERIMENT & (bit)) != 0)

namespace plslam {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4), aligned(4)));   // descriptor rows are only 4-byte aligned
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Pointers read from the launch tables are GENERIC to the compiler, and a generic access is a FLAT instruction, which
// counts on lgkmcnt as well as vmcnt: the `s_waitcnt lgkmcnt(0)` in front of every workgroup barrier then waits for
// the raw-row PREFETCH of two tiles ahead, i.e. every tile pays a full memory latency.  With the address space spelled
// out the loads are global_load (vmcnt only) and stay in flight across the barrier.
#define PLSLAM_GLOBAL __attribute__((address_space(1)))
typedef const PLSLAM_GLOBAL uint32_t* gcu32_t;
typedef const PLSLAM_GLOBAL u32x4_t* gcu32x4_t;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef PLSLAM_GLOBAL u32x2_t* gu2_t;
typedef PLSLAM_GLOBAL uint32_t* gu32_t;

namespace {

constexpr int MF_TILE_N = 32;                 // b rows per tile
constexpr int MF_KSTEPS = 4;                  // 256 bits = 4 x K 64
constexpr int MF_ROW_STRIDE = 144;            // bytes per expanded b row in LDS (128 + 16: 4-bank skew)
constexpr int MF_TILE_BYTES = MF_TILE_N * MF_ROW_STRIDE;
// Streaming accesses (the column partials: written once by the scan, read once by the merge) carry the non-temporal hint, so
// that they do not evict the b rows the scan re-reads from L2 -- of this scan or, in the split stepping, of the next one
#ifndef PLSLAM_NT_STREAMS
#define PLSLAM_NT_STREAMS 1
#endif
#ifndef PLSLAM_MG_GROUP
#define PLSLAM_MG_GROUP 16
#endif
// tiles per row-direction group (a window of 64 tiles = 4 groups of 16).  A larger group halves the parked-pair pushes
// (64 VALU + 16 LDS reads + 16 LDS writes each; measured 8 % of the scan at 8 tiles per group) and costs
// MF_GROUP - 1 recomputed distances per row and window.
constexpr int MF_GROUP = PLSLAM_MG_GROUP;
constexpr int MF_CGROUP = 8;                  // tiles whose column results are staged in LDS and stored together (256 columns)
// fp4 (e2m1) codes: +1.0 = 0x2, -1.0 = 0xA.  b side: bit 0 -> +1, bit 1 -> -1 = s(b); the a side is the b code
// XOR 0x8 per nibble (= -s(a)) and carries the block scale 2^6 (E8M0 133), the b side 2^0 (E8M0 127).
constexpr uint32_t FP4_NEG = 0x88888888u;
constexpr uint32_t FP4_ONE = 0x22222222u;
constexpr int SCALE_A = 133, SCALE_B = 127;
constexpr uint32_t ACC_BITS = 0x4B000000u + 16384u;   // float bits of 2^23 + 16384 (+ small integers: + the integer)

__device__ __forceinline__ uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax_(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ void merge2(uint32_t& a0, uint32_t& a1, uint32_t c0, uint32_t c1)
{
    const uint32_t lo = umin_(a0, c0);
    const uint32_t hi = umin_(umax_(a0, c0), umin_(a1, c1));
    a0 = lo;
    a1 = hi;
}
// packed 16-bit min / max (inline asm: see hamming_mfma.hip -- the vector builtins get sunk out of the MFMA block)
__device__ __forceinline__ uint32_t pk_min16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_max16(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t pk_add16_sat(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_pk_add_u16 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void pk_push2(uint32_t& b0, uint32_t& b1, uint32_t key)
{
    b1 = pk_min16(b1, pk_max16(b0, key));
    b0 = pk_min16(b0, key);
}
// accumulators of the two M-tiles (2^23 + 128 d + tag, tag <= 127) side by side: hi.lo16 << 16 | lo.lo16.
// The BUILTIN, never inline asm: this is the one instruction that reads MFMA results, and the wait states between an
// MFMA and a VALU access to its destination are the compiler's job ("K1e determinism").
__device__ __forceinline__ uint32_t pack_acc(float lo, float hi, uint32_t sel_uniform /* 0x05040100 */)
{
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), sel_uniform);
}
// 16-bit keys are (d << 7) | tag7 (<= (256 << 7) + 127 = 0x807F); anything above is "none"
constexpr uint32_t KEY16_MAX = 0x807Fu;
__device__ __forceinline__ uint32_t key16_to_key32(uint32_t k16, uint32_t tag_bias, uint32_t idx_base,
                                                   uint32_t idx_scale)
{
    return k16 > KEY16_MAX ? KEY_NONE
                           : (((k16 >> 7) << KEY_IDX_BITS) | (idx_base + ((k16 & 127u) - tag_bias) * idx_scale));
}
// 32 bits of a descriptor -> 32 fp4 codes of s(bit): dword s holds bits 4k + s, nibble k = 0x2 | bit << 3
// 7 VALU ops of the fast class (measured ~2.5 cycles per wave instruction against ~4.2 for shifts and v_and_or): three adds
// for x << 1, 2, 3 and four v_bitop3 (a & b) | c.  Written with asm / the builtin because the compiler turns x + x back
// into a shift and (x & m) | c into v_and + v_or.
template <bool A_SIDE>
__device__ __forceinline__ i32x4 expand_dword_fp4(uint32_t x)
{
    uint32_t x1, x2, x3;
    asm("v_add_u32 %0, %1, %1" : "=v"(x1) : "v"(x));
    asm("v_add_u32 %0, %1, %1" : "=v"(x2) : "v"(x1));
    asm("v_add_u32 %0, %1, %1" : "=v"(x3) : "v"(x2));
    constexpr uint32_t base = A_SIDE ? (FP4_ONE ^ FP4_NEG) : FP4_ONE;       // a side: sign nibble-bit flipped
    constexpr unsigned TT = A_SIDE ? 0x6Au : 0xEAu;                         // (a & b) ^ c  |  (a & b) | c
    i32x4 v;
    v.x = (int)__builtin_amdgcn_bitop3_b32(x3, FP4_NEG, base, TT);
    v.y = (int)__builtin_amdgcn_bitop3_b32(x2, FP4_NEG, base, TT);
    v.z = (int)__builtin_amdgcn_bitop3_b32(x1, FP4_NEG, base, TT);
    v.w = (int)__builtin_amdgcn_bitop3_b32(x, FP4_NEG, base, TT);
    return v;
}
__device__ __forceinline__ uint32_t bcnt_acc_(uint32_t x, uint32_t acc)
{
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}
__device__ __forceinline__ int xcd_remap_(int orig, int nwg) { return (orig & 7) * (nwg >> 3) + (orig >> 3); }

}  // namespace

// MULTI = false: every problem of the launch has n2 <= 2048 (one window of 64 tiles; the window bounds are
// compile-time facts).  MULTI = true: any n2.  DIRECTED = true: only keys12 (row direction) is produced.
// FUSED = true: one workgroup per PROBLEM.  It walks the problem's row blocks of 256 itself and then finishes the problem:
// column partials (still in L2 / the Infinity Cache) -> keys21 in LDS -> ratio test + mutual check -> matches_12 and the
// match count.  No merge kernel, no finalize kernel, no counter zeroing, no keys21 round trip through HBM.
template <bool MULTI, bool DIRECTED, bool FUSED>
__global__ void __launch_bounds__(256, 3)      // 3 waves per SIMD: <= 168 unified VGPRs
k_scan_sym_mfma_g(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks,
                  int32_t* __restrict__ zero, int nzero)
{
    // one buffer, two lives: during the scan the double-buffered b tile (9 216 B) followed by the PARKED sorted pairs of the
    // row direction ([wave][reg][lane] x 8 B = 32 768 B: they are touched once per group of tiles, so they live here and not in
    // 32 VGPRs); after the scan the row-result transpose [wave][row 0..63][33] (33 792 B) over both
    constexpr int ROWX_STRIDE = 33;               // dwords per row: lane = row reads are conflict-free
    constexpr int PARK_OFF = 2 * MF_TILE_BYTES;
    constexpr int SMEM_BYTES = PARK_OFF + 4 * 16 * 64 * 8;
    static_assert(SMEM_BYTES >= 4 * 64 * ROWX_STRIDE * 4, "the transpose must fit");
    __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM_BYTES];
    // column results of the current group of 8 tiles, per wave [tile in group][column] (1 KB per wave): they leave for
    // HBM once per group as one 16-byte store per lane.  A global store per tile would sit in the same counter (vmcnt) as
    // the raw-row prefetch, and with loads AND stores pending the counter is out of order: every wait becomes
    // vmcnt(0), the prefetch distance collapses to one tile, and a tile then costs a full memory latency (round 2
    // finding: 2.7 ms of the 3.4 ms scan were there with NO bookkeeping at all in the kernel).
    __shared__ __attribute__((aligned(16))) uint32_t colstage[4][MF_CGROUP * MF_TILE_N];
    uint8_t* const btile = smem;
    u32x2_t* const park = reinterpret_cast<u32x2_t*>(smem + PARK_OFF) + (threadIdx.x >> 6) * (16 * 64) + (threadIdx.x & 63);

    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nzero; i += 256) zero[i] = 0;

    const int wg = xcd_remap_(blockIdx.x, gridDim.x);
    const BlockDesc bd = blocks[wg];
    if (bd.item < 0) return;                       // padding entry of the XCD-striped table
    const SymDesc sd = syms[bd.item];
    // (the row count may live on the device: the launch and the tables are sized for sd.n1, the rows that exist are the first
    // *n1_dev; a workgroup whose rows all lie behind them has nothing to do -- its partials are never read)
    const int n1 = sd.n1_dev ? min(sd.n1, *(const PLSLAM_GLOBAL int32_t*) sd.n1_dev) : sd.n1;
    const int n2 = sd.n2;
    if (!FUSED && bd.row0 >= n1) return;
    const int n2p = (n2 + 255) & ~255;             // rows of the partial table are padded to 256 columns: whole groups are stored
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, g = lane >> 5;
    const gcu32_t araw = (gcu32_t) reinterpret_cast<const uint32_t*>(sd.a);
    const gcu32_t braw = (gcu32_t) reinterpret_cast<const uint32_t*>(sd.b);
    for (int i0 = FUSED ? 0 : bd.row0;; i0 += 256) {     // first of the 256 a-rows at hand (not FUSED: the one block of the entry)
    const int iw = i0 + 64 * w;                    // first of this wave's 64

    // ---- A operands: rows iw + 32 mt + c, raw dword 2 ks + g of each, as fp4 codes of -s(a) (the b code with the
    // sign nibble-bit flipped); the factor 64 is the block scale ----
    i32x4 afrag[2][MF_KSTEPS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = iw + 32 * mt + c;
        const int rrow = row < n1 ? row : n1 - 1;                 // clamped; masked in the epilogue
        const gcu32_t p = araw + (size_t)rrow * 8 + g;           // this lane's dword of each K-step: 2 ks + g
#pragma unroll
        for (int ks = 0; ks < MF_KSTEPS; ++ks) afrag[mt][ks] = expand_dword_fp4<true>(p[2 * ks]);
    }
    const int scale_a = SCALE_A, scale_b = SCALE_B;
    const uint32_t pack_sel = 0x05040100u;

    // row-direction state per accumulator register r (M-tile 0 in the low halves, M-tile 1 in the high halves):
    //   gm[r]    running minimum of the 16-bit keys (d << 7 | tile + LOC) of the current group of MF_GROUP tiles
    //   park[r]  (LDS) the best two GROUP minima of the lane's column class
    uint32_t gm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) gm[r] = 0xFFFFFFFFu;

    const bool rows_ragged = iw + 64 > n1;         // wave-uniform: some of this wave's rows do not exist
    const uint32_t ibase = (uint32_t)(iw + 4 * g); // + local index = a-row of an accumulator
    const uint32_t ghtag = (uint32_t)(4 * g) | ((uint32_t)(4 * g + 32) << 16);   // see finish_columns
    // column partials of THIS WAVE's 64 rows: part16[(row block of 64)][column] = best | second << 16 as 16-bit keys
    // (d << 7 | row within the block); the merge kernel (k_merge_partials16) widens and combines them.  Every wave writes
    // its own results straight from registers: no LDS exchange between the waves, no flush step on the tile's critical path.
    const gu32_t part = DIRECTED ? (gu32_t) nullptr : (gu32_t) sd.part21 + (size_t)(iw >> 6) * n2p;
    const bool wave_has_rows = iw < n1;
    uint32_t* const cstage = colstage[w];
    *reinterpret_cast<i32x4*>(cstage + 4 * lane) = i32x4{-1, -1, -1, -1};     // wave-private

    // expansion duty of this lane: b row (tid >> 3) of the tile, dword (tid & 7) of it
    const int ej = tid >> 3, ewd = tid & 7;
    const int ntiles = (n2 + MF_TILE_N - 1) / MF_TILE_N;
    auto load_raw = [&](int t) __attribute__((always_inline)) -> uint32_t {
        int j = t * MF_TILE_N + ej;
        j = j < n2 ? j : n2 - 1;
        return braw[(size_t)j * 8 + ewd];
    };
    auto expand_store = [&](uint32_t raw, int buf) __attribute__((always_inline)) {
        uint8_t* dst = btile + buf * MF_TILE_BYTES + ej * MF_ROW_STRIDE + ewd * 16;
        *reinterpret_cast<i32x4*>(dst) = expand_dword_fp4<false>(raw);
    };
    // Bookkeeping of one packed key pair (rows LOC and LOC + 32 of the wave, column j0 + c).  kc[R] was packed from the
    // accumulators right after the tile's MFMAs (pack_tile) and holds the keys of BOTH directions (the tile number came
    // in through the accumulator seed): ONE packed min into the group minimum of the row direction, 3 packed ops for the
    // column best-2: 4 VALU ops per 2 distances (+ the pack).  MASKED = false is the steady state (every row of this
    // wave and every column of the tile exists).
#define PLSLAM_MG_EPI_ROW(R)                                                                       \
    {                                                                                              \
        constexpr uint32_t LOC = ((R) & 3) + 8 * ((R) >> 2);                                       \
        uint32_t kcv = kc[R];                                                                      \
        uint32_t kr = kcv;                                                                         \
        if (MASKED) {                                                                              \
            kr = col_ok ? kr : 0xFFFFFFFFu;                                                        \
            kcv |= ((int)(ibase + LOC) < n1 ? 0u : 0x0000FFFFu) |                                  \
                   ((int)(ibase + LOC + 32u) < n1 ? 0u : 0xFFFF0000u);                             \
        }                                                                                          \
        gm[R] = pk_min16(gm[R], kr);                                                               \
        if (!DIRECTED) pk_push2(cb0, cb1, kcv);                                                    \
    }
    {
#define WT0 (MULTI ? wt0v : 0)
#define WT1 (MULTI ? wt1v : ntiles)
    // The 16-bit row keys hold 64 tile numbers, so the scan runs in WINDOWS of 64 tiles (2048 columns): after
    // each window the row state is reduced, completed (second best) and merged into keys12, then restarted.
    int wt0v = 0, wt1v = ntiles < 64 ? ntiles : 64;
    uint32_t raw1 = 0u, raw2 = 0u, raw3 = 0u;              // raw b dwords of tiles t+1, t+2, t+3 of the coming step
    // the 8 tiles that end with tile `tl` are over: their column results go to the partial table (256 columns: 16 bytes
    // per lane)
    auto store_columns = [&](int tl) __attribute__((always_inline)) {
        if (!DIRECTED && wave_has_rows && !PLSLAM_MG_X(8)) {
            const int j0 = (tl & ~(MF_CGROUP - 1)) * MF_TILE_N + 4 * lane;      // WT0 is a multiple of 8
            const i32x4 v = *reinterpret_cast<const i32x4*>(cstage + 4 * lane);
            if (j0 < n2p) {
                if (PLSLAM_NT_STREAMS) __builtin_nontemporal_store(v, reinterpret_cast<PLSLAM_GLOBAL i32x4*>(part + j0));
                else *reinterpret_cast<PLSLAM_GLOBAL i32x4*>(part + j0) = v;
            }
            // tiles of a partial last group that never ran leave "none" in the padding columns (never read; keeps the
            // partial table a pure function of the inputs, which tools/determinism_check.py compares word for word)
            *reinterpret_cast<i32x4*>(cstage + 4 * lane) = i32x4{-1, -1, -1, -1};
        }
    };
    // a row group is over: its minima go into the sorted pairs, the minima restart
    auto push_groups = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const u32x2_t v = park[r * 64];
            uint32_t b0 = v.x, b1 = v.y;
            pk_push2(b0, b1, gm[r]);
            park[r * 64] = u32x2_t{b0, b1};
            gm[r] = 0xFFFFFFFFu;
        }
    };
    // column best-2 of a finished tile: the two halves of (cb0, cb1) are sorted streams over disjoint rows
    // of the same column -> best 2 of the lane, then of the wave (lanes l and l + 32 hold the same column), as 16-bit
    // keys.  The keys still carry "+ tile within the window" from the accumulator seed; every key of a tile carries
    // the same offset, so all comparisons here are unaffected and the merge step takes it off (a "none" half 0xFFFF
    // stays above KEY16_MAX after the subtraction).
    auto finish_columns = [&](int t, uint32_t cb0, uint32_t cb1) __attribute__((always_inline)) {
        if (DIRECTED || PLSLAM_MG_X(32)) { asm volatile("" ::"v"(cb0), "v"(cb1)); return; }
        // The tag of a column key is LOC (bits 0,1,3,4 of the row within the wave) [+ tile]; OR-ing in bit 2 (= g) and
        // bit 5 (= M-tile, the high halves) makes it the full row within the wave [+ tile: LOC + tile < 128 and the
        // OR-ed bits are clear in LOC but NOT in LOC + tile, so they are ADDED: no carry leaves the 7-bit tag because
        // row + tile <= 63 + 63; the packed add saturates, so a "none" half (0xFFFF: masked rows) stays 0xFFFF and never
        // carries into its neighbour], so keys of the two halves and of lane + 32 compare directly.
        cb0 = pk_add16_sat(cb0, ghtag);
        cb1 = pk_add16_sat(cb1, ghtag);
        const uint32_t e0 = cb0 & 0xFFFFu, o0 = cb0 >> 16, e1 = cb1 & 0xFFFFu, o1 = cb1 >> 16;
        uint32_t m0 = umin_(e0, o0), m1 = umin_(umax_(e0, o0), umin_(e1, o1));
        // lanes l < 32 fetch lane l + 32's pair with one VALU swap (gfx950: v_permlane32_swap), no LDS round trip;
        // lanes >= 32 compute a value nobody stores
        const uint32_t mine = m0 | (m1 << 16);
        const auto sw = __builtin_amdgcn_permlane32_swap(mine, mine, false, false);
        const uint32_t other = sw[1];
        merge2(m0, m1, other & 0xFFFFu, other >> 16);
        if (lane < MF_TILE_N) cstage[((t - WT0) & (MF_CGROUP - 1)) * MF_TILE_N + lane] = m0 | (m1 << 16);
    };
    // Software pipeline, ONE accumulator set.  A tile's life:  M(t): 8 MFMAs -> P(t): 16 v_perm pack the 32 accumulators
    // into 16 key pairs kc[] (the accumulator registers are free again) -> E(t): bookkeeping from kc[], issued BETWEEN the
    // MFMAs of M(t+1).  So a wave has matrix work and VALU work in flight at the same time with 32 + 16 live registers
    // instead of two accumulator sets (round 1's two-set form collapsed into [8 MFMA][wait][bookkeeping] under the
    // 168-register budget: nothing overlapped inside a wave, and once the bookkeeping shrank the scan stopped following
    // the VALU instruction count).  What sits between the last MFMA of M(t) and P(t) -- the expansion of tile t+1, the
    // column results of tile t-1 -- covers the matrix pipe's latency.
    //   step(t) = barrier | operand reads | seeds | M(t) x E(t-1) | expand(t+1) | finish_columns(t-1) | P(t)
    uint32_t kc[16];
    auto tile_step = [&](int t, bool with_prev, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        if (!PLSLAM_MG_X(1)) __syncthreads();  // tile t expanded; every wave is past its reads of the other buffer
        const uint8_t* bt = btile + (t & 1) * MF_TILE_BYTES + c * MF_ROW_STRIDE + 16 * g;
        const bool col_ok = (t - 1) * MF_TILE_N + c < n2;
        uint32_t cb0 = 0xFFFFFFFFu, cb1 = 0xFFFFFFFFu;
        // all four operand reads of the tile go out at once, right behind the barrier; what follows until the first MFMA
        // needs one of them (the expansion of tile t+1, whose buffer every wave left before this barrier; the seeds; the
        // first bookkeeping rows) covers the LDS latency
        i32x4 bfr[MF_KSTEPS];
#pragma unroll
        for (int ks = 0; ks < MF_KSTEPS; ++ks)
            bfr[ks] = PLSLAM_MG_X(256) ? i32x4{(int)FP4_ONE + t, (int)FP4_ONE, (int)FP4_ONE + ks, (int)FP4_ONE}
                                       : *reinterpret_cast<const i32x4*>(bt + 32 * ks);
        if (!PLSLAM_MG_X(128)) expand_store(raw1, (t + 1) & 1);   // past the last tile: a harmless rewrite of the idle buffer
        // raw-row prefetch, three tiles deep: the request for tile t+4 goes out now, its dword is expanded in step t+3.
        // Always issued (load_raw clamps the row), so the number of loads in flight is the same on every path and the
        // waits are exact counts.
        raw1 = raw2;
        raw2 = raw3;
        if (!PLSLAM_MG_X(128)) raw3 = load_raw(t + 4);
        // accumulator start: 2^23 + 16384 + LOC(reg) + tile within the window: the sum is 2^23 + 128 d + LOC + tile,
        // every partial sum an integer below 2^24, so fp32 accumulation is exact and the float's low 16 bits ARE
        // the key (d << 7 | LOC + tile).  Wave-uniform integers (scalar adds); built from integers through a scalar
        // because __builtin_bit_cast applied to a vector ELEMENT is miscompiled by this toolchain (ROCm 7.2).
        f32x16 cseed, m0, m1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint32_t bits = ACC_BITS + (uint32_t)((r & 3) + 8 * (r >> 2)) + (uint32_t)(t - WT0);
            const float f = __builtin_bit_cast(float, bits);
            cseed[r] = f;
        }
#define PLSLAM_MG_MMA(ACC, MT, KS, CIN)                                                            \
        {                                                                                          \
            const i32x8 a8 = {afrag[MT][KS].x, afrag[MT][KS].y, afrag[MT][KS].z, afrag[MT][KS].w, 0, 0, 0, 0}; \
            const i32x8 b8 = {bfr[KS].x, bfr[KS].y, bfr[KS].z, bfr[KS].w, 0, 0, 0, 0};             \
            if (!PLSLAM_MG_X(4))                                                                   \
                ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, CIN, 4, 4, 0, scale_a, 0, scale_b); \
            else { const f32x16 cin_ = CIN; ACC = cin_; ACC[KS] = __builtin_bit_cast(float, bfr[KS].x ^ a8[0]); } \
            /* an EMPTY asm (no instruction): pins the MFMA here -- without a use in this block the optimizer sinks all */ \
            /* eight MFMAs of a tile down to the pack, i.e. behind the bookkeeping they are meant to overlap with */ \
            asm volatile("" : "+v"(ACC));                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
#define PLSLAM_MG_EPI2(R)                                                                          \
        {                                                                                          \
            if (with_prev && !PLSLAM_MG_X(2)) { PLSLAM_MG_EPI_ROW(R) PLSLAM_MG_EPI_ROW((R) + 1) }  \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
        // program order, fenced: [2 rows] MFMA [2 rows] MFMA ... : single MFMAs, evenly spaced, each followed by VALU work
        // that does not depend on it
        __builtin_amdgcn_sched_barrier(0);
        PLSLAM_MG_EPI2(0)  PLSLAM_MG_MMA(m0, 0, 0, cseed)
        PLSLAM_MG_EPI2(2)  PLSLAM_MG_MMA(m1, 1, 0, cseed)
        PLSLAM_MG_EPI2(4)  PLSLAM_MG_MMA(m0, 0, 1, m0)
        PLSLAM_MG_EPI2(6)  PLSLAM_MG_MMA(m1, 1, 1, m1)
        PLSLAM_MG_EPI2(8)  PLSLAM_MG_MMA(m0, 0, 2, m0)
        PLSLAM_MG_EPI2(10) PLSLAM_MG_MMA(m1, 1, 2, m1)
        PLSLAM_MG_EPI2(12) PLSLAM_MG_MMA(m0, 0, 3, m0)
        PLSLAM_MG_EPI2(14) PLSLAM_MG_MMA(m1, 1, 3, m1)
#undef PLSLAM_MG_EPI2
#undef PLSLAM_MG_MMA
        if (with_prev) {
            finish_columns(t - 1, cb0, cb1);
            // wave-uniform: tile t-1 closed a block of 8 tiles / a row group
            if (((t - 1 - WT0) & (MF_CGROUP - 1)) == MF_CGROUP - 1) store_columns(t - 1);
            if (((t - 1 - WT0) & (MF_GROUP - 1)) == MF_GROUP - 1 && !PLSLAM_MG_X(64)) push_groups();
        }
        // P(t): the key pairs of tile t; the accumulators are dead from here on
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float f0 = m0[r], f1 = m1[r];
            if (!PLSLAM_MG_X(512)) kc[r] = pack_acc(f0, f1, pack_sel);
        }
        if (PLSLAM_MG_X(512)) { asm volatile("" ::"v"(m0), "v"(m1)); kc[0] = __builtin_bit_cast(uint32_t, (float)m0[0]); }
    };
    // E(t) on its own (the last tile of a window has no following M step to hide under)
    auto epilogue = [&](int t, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const bool col_ok = t * MF_TILE_N + c < n2;
        uint32_t cb0 = 0xFFFFFFFFu, cb1 = 0xFFFFFFFFu;
        PLSLAM_MG_EPI_ROW(0) PLSLAM_MG_EPI_ROW(1) PLSLAM_MG_EPI_ROW(2) PLSLAM_MG_EPI_ROW(3)
        PLSLAM_MG_EPI_ROW(4) PLSLAM_MG_EPI_ROW(5) PLSLAM_MG_EPI_ROW(6) PLSLAM_MG_EPI_ROW(7)
        PLSLAM_MG_EPI_ROW(8) PLSLAM_MG_EPI_ROW(9) PLSLAM_MG_EPI_ROW(10) PLSLAM_MG_EPI_ROW(11)
        PLSLAM_MG_EPI_ROW(12) PLSLAM_MG_EPI_ROW(13) PLSLAM_MG_EPI_ROW(14) PLSLAM_MG_EPI_ROW(15)
        finish_columns(t, cb0, cb1);
    };
    // One window: M(WT0) P | M(WT0+1) x E(WT0) P | ... | M(WT1-1) x E(WT1-2) P | E(WT1-1).  Only the last tile of the
    // scan can lack columns.
    auto pipeline = [&](auto steady_tag) __attribute__((always_inline)) {
        const bool last_partial = WT1 == ntiles && (n2 % MF_TILE_N) != 0;
        tile_step(WT0, false, steady_tag);
        for (int t = WT0 + 1; t < WT1; ++t) tile_step(t, true, steady_tag);
        if (last_partial) epilogue(WT1 - 1, std::true_type{}); else epilogue(WT1 - 1, steady_tag);
        store_columns(WT1 - 1);                    // the (possibly partial) last block of columns
        push_groups();                             // ... and row group
    };
    // Row results of a window.  Every lane holds, per accumulator register, the best two GROUP minima (16-bit keys
    // (d, tile + LOC)) of ITS column class for two rows.  Transpose through LDS so that one lane owns one row: lane l
    // reads the 32 class entries of row l in class order, widens them to (key16 << 16 | class) -- which orders like
    // (d, j = 32 tile + class) because every entry of a row carries the same LOC -- and keeps the best two.  The
    // first is the row's best key; the second is the best key OUTSIDE the winner's group, so the lane then visits
    // the other columns of the winner's group (same class, the other tiles of the group) and recomputes their
    // distances from the raw rows: second best = min of the two.  Callers guarantee that all waves are past their
    // last operand read of `smem`; the region used here is private to the wave.
    auto finish_rows = [&]() __attribute__((always_inline)) {
        uint32_t* rowx = reinterpret_cast<uint32_t*>(smem) + w * (64 * ROWX_STRIDE);
        u32x2_t rb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rb[r] = park[r * 64];
        __syncthreads();                           // every wave holds its pairs: the transpose may overwrite the parking area
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = (r & 3) + 8 * (r >> 2) + 4 * g;
            // (best | second << 16) of M-tile 0 (low halves) and of M-tile 1 (high halves)
            rowx[lrow * ROWX_STRIDE + c] = (rb[r].x & 0xFFFFu) | (rb[r].y << 16);
            rowx[(32 + lrow) * ROWX_STRIDE + c] = (rb[r].x >> 16) | (rb[r].y & 0xFFFF0000u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
        const uint32_t* mine = rowx + lane * ROWX_STRIDE;
#pragma unroll 8
        for (int cls = 0; cls < 32; ++cls) {
            const uint32_t e = mine[cls];
            merge2(k0, k1, (e << 16) | (uint32_t)cls, (e & 0xFFFF0000u) | (uint32_t)cls);
        }
        // (key16 << 16 | class) -> (d << 23 | 32 (WT0 + tag - LOC) + class); LOC of local row l: l without bit 2 (= g)
        const uint32_t loc = (uint32_t)(lane & 31 & ~4);
        auto widen = [&](uint32_t k) -> uint32_t {
            const uint32_t k16 = k >> 16, cls = k & 0xFFFFu;
            return key16_to_key32(k16, loc, cls + (uint32_t)(WT0 * MF_TILE_N), (uint32_t)MF_TILE_N);
        };
        const int row = iw + lane;
        if (row < n1) {
            const gu2_t out = (gu2_t) reinterpret_cast<u32x2_t*>(sd.keys12) + row;
            uint32_t r0 = widen(k0), r1 = widen(k1);
            if ((k0 >> 16) <= KEY16_MAX && !PLSLAM_MG_X(16)) {
                // the other members of the winner's group: tiles g0 .. g0 + 7 of the window, same class
                const uint32_t cls0 = k0 & 0xFFFFu;
                const uint32_t tw = ((k0 >> 16) & 127u) - loc;              // winner's tile within the window
                const uint32_t g0 = (uint32_t)WT0 + (tw & ~(uint32_t)(MF_GROUP - 1));
                const gcu32x4_t ap = (gcu32x4_t)(araw + (size_t)row * 8);
                const u32x4_t a_lo = ap[0], a_hi = ap[1];
#pragma unroll
                for (int k = 0; k < MF_GROUP; ++k) {
                    const uint32_t tile = g0 + (uint32_t)k;
                    const uint32_t j = tile * MF_TILE_N + cls0;
                    const bool ok = (uint32_t)k != (tw & (uint32_t)(MF_GROUP - 1)) && tile < (uint32_t)WT1 && j < (uint32_t)n2;
                    const gcu32x4_t bp = (gcu32x4_t)(braw + (size_t)(ok ? j : 0u) * 8);
                    const u32x4_t b_lo = bp[0], b_hi = bp[1];
                    uint32_t d = bcnt_acc_(a_lo.x ^ b_lo.x, 0u);
                    d = bcnt_acc_(a_lo.y ^ b_lo.y, d);
                    d = bcnt_acc_(a_lo.z ^ b_lo.z, d);
                    d = bcnt_acc_(a_lo.w ^ b_lo.w, d);
                    d = bcnt_acc_(a_hi.x ^ b_hi.x, d);
                    d = bcnt_acc_(a_hi.y ^ b_hi.y, d);
                    d = bcnt_acc_(a_hi.z ^ b_hi.z, d);
                    d = bcnt_acc_(a_hi.w ^ b_hi.w, d);
                    const uint32_t cand = ok ? ((d << KEY_IDX_BITS) | j) : KEY_NONE;
                    r1 = umin_(r1, cand);
                }
            }
            if (WT0 > 0) {                                  // later windows: merge with the windows before
                const u32x2_t prev = *out;
                merge2(r0, r1, prev.x, prev.y);
            }
            *out = u32x2_t{r0, r1};
            // two-launch column-split plans (k_split_post): the sub-problem of a problem's FIRST column range names the match
            // table, and its workgroups -- one per row block -- clear the rows; the kernel behind the scan writes the matches
            if (!FUSED && sd.matches_12) ((PLSLAM_GLOBAL int32_t*) sd.matches_12)[row] = -1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    for (;;) {
#pragma unroll
        for (int r = 0; r < 16; ++r) park[r * 64] = u32x2_t{0xFFFFFFFFu, 0xFFFFFFFFu};   // wave-private: no barrier needed
        expand_store(load_raw(WT0), 0);            // WT0 is a multiple of 64: buffer parity restarts at 0
        raw1 = load_raw(WT0 + 1);
        raw2 = load_raw(WT0 + 2);
        raw3 = load_raw(WT0 + 3);
        if (!rows_ragged) pipeline(std::false_type{}); else pipeline(std::true_type{});
        __syncthreads();                           // every wave is past its last operand read of the b tile
        finish_rows();
        if (!MULTI || wt1v == ntiles) break;
        __syncthreads();                           // smem becomes the b tile (+ parking area) again
        wt0v = wt1v;
        wt1v = ntiles < wt0v + 64 ? ntiles : wt0v + 64;
    }
#undef WT0
#undef WT1
    }
    if (!FUSED || i0 + 256 >= n1) break;
    __syncthreads();                               // the transpose area becomes b tile + parking area again
    }   // row blocks
#undef PLSLAM_MG_EPI_ROW
    if (!FUSED) return;

    // ---- the problem's tail: merge the column partials, ratio test + mutual check (K1c' + K2 of the unfused path) ----
    // Every wave waits for its own stores (keys12 rows, partials), then the workgroup meets.  All waves of a workgroup
    // share one L1 (the kernel is not built for threadgroup-split mode), so after the barrier plain loads see them.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t* const k21 = reinterpret_cast<uint32_t*>(smem);           // n2 x (best, second): <= 32 KB (host checks n2)
    if (!DIRECTED) {
        const gcu32_t part = (gcu32_t) sd.part21;
        const int nwb = (n1 + 63) >> 6;
        const gu2_t k21g = (gu2_t) reinterpret_cast<u32x2_t*>(sd.keys21);
        for (int j = tid; j < n2; j += 256) {
            uint32_t b0 = KEY_NONE, b1 = KEY_NONE;
            const uint32_t tw = ((uint32_t)j >> 5) & 63u;  // the keys still carry + tile within the 64-tile window
#pragma unroll 8
            for (int wb = 0; wb < nwb; ++wb) {
                const uint32_t e = part[(size_t)wb * n2p + j];
                merge2(b0, b1, key16_to_key32((e & 0xFFFFu) - tw, 0u, (uint32_t)(64 * wb), 1u),
                       key16_to_key32((e >> 16) - tw, 0u, (uint32_t)(64 * wb), 1u));
            }
            k21[2 * j] = b0;
            k21[2 * j + 1] = b1;
            if (k21g) k21g[j] = u32x2_t{b0, b1};                      // diagnostics (plslam_match_plan_dump) only
        }
        __syncthreads();
    }
    // stvo-pl matchNNR: accept iff (float)d0 < (float)d1 * nnr (one fp32 multiply); match(): keep i1 -> i2 iff m21[i2] == i1
    auto ratio_pick = [&](uint32_t q0, uint32_t q1) -> int {
        if (q1 == KEY_NONE) return -1;                                 // fewer than two neighbours: "no match"
        const float d0 = (float)(q0 >> KEY_IDX_BITS);
        const float d1n = __fmul_rn((float)(q1 >> KEY_IDX_BITS), sd.nnr);
        return d0 < d1n ? (int)(q0 & KEY_IDX_MASK) : -1;
    };
    int found = 0;
    for (int ib = 0; ib < n1; ib += 256) {
        const int i1 = ib + tid;
        int m = -1;
        if (i1 < n1) {
            const u32x2_t q = ((gu2_t) reinterpret_cast<u32x2_t*>(sd.keys12))[i1];
            m = ratio_pick(q.x, q.y);
            if (!DIRECTED && m >= 0 && sd.mutual && ratio_pick(k21[2 * m], k21[2 * m + 1]) != i1) m = -1;
            ((PLSLAM_GLOBAL int32_t*) sd.matches_12)[i1] = m;
        }
        found += (int)__popcll(__ballot(m >= 0));                      // wave-uniform
    }
    if (sd.n_matches) {
        __syncthreads();                                               // k21 is dead: its first words become the counter
        int* const cnt = reinterpret_cast<int*>(smem);
        if (tid == 0) *cnt = 0;
        __syncthreads();
        if (lane == 0) atomicAdd(cnt, found);
        __syncthreads();
        if (tid == 0) *((PLSLAM_GLOBAL int32_t*) sd.n_matches) = *cnt;
    }
}

// K1c'  merge of K1f's column partials: keys21[j] = best-2 over the 64-row blocks of part16[block][j] (16-bit keys
// (d << 7 | row within the block), best | second << 16), widened to (d << 23 | row).  PARTS lanes share a column (each
// takes every PARTS-th row block, then LDS): a batch of 1500-row problems has 24 row blocks per column and thousands of
// columns (PARTS = 1: HBM-bound), ONE 10 000-row local map has 157 row blocks and 1500 columns -- a serial chain of 157
// loads per lane on 6 workgroups took longer (13.6 us) than the scan itself.
template <int PARTS>
__global__ void __launch_bounds__(256)
k_merge_partials16(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks)
{
    constexpr int COLS = 256 / PARTS;
    __shared__ uint32_t red[PARTS > 1 ? 512 : 2];
    const BlockDesc bd = blocks[blockIdx.x];
    const SymDesc sd = syms[bd.item];
    const int jl = (int)threadIdx.x % COLS, part_id = (int)threadIdx.x / COLS;
    const int j = bd.row0 + jl;
    const gcu32_t part = (gcu32_t) sd.part21;
    const int nwb = (sd.n1 + 63) >> 6;
    const int n2p = (sd.n2 + 255) & ~255;                  // padded row of the partial table
    uint32_t b0 = KEY_NONE, b1 = KEY_NONE;
    const uint32_t tw = ((uint32_t)j >> 5) & 63u;          // the keys still carry + tile within the 64-tile window
    if (j < sd.n2) {
        // Only a block's BEST key is widened and merged; the second best overall is either the best of another block (b1)
        // or the second key of the block that holds the overall best (kept raw in s0, widened once at the end): half the
        // instructions per entry -- they matter because this kernel runs under the next step's instruction-bound scan.
        // A "none" half (0xFFFF - tw) widens to a key with d >= 257: it loses every comparison and is mapped to KEY_NONE
        // at the end.
        auto wide = [](uint32_t k16, uint32_t wb) -> uint32_t {         // (d << 7 | row in block) -> (d << 23 | row)
            return ((k16 << 16) & 0xFF800000u) | ((k16 & 63u) + 64u * wb);
        };
        const uint32_t tw2 = tw | (tw << 16);
        uint32_t s0 = 0xFFFFFFFFu;
#pragma unroll 8
        for (int wb = part_id; wb < nwb; wb += PARTS) {
            const uint32_t e = (PLSLAM_NT_STREAMS ? __builtin_nontemporal_load(&part[(size_t)wb * n2p + j])
                                                  : part[(size_t)wb * n2p + j]) - tw2;    // both halves >= tw: no borrow between them
            const uint32_t k = wide(e & 0xFFFFu, (uint32_t)wb);
            s0 = k < b0 ? e : s0;
            b1 = umin_(b1, umax_(b0, k));
            b0 = umin_(b0, k);
        }
        if (b0 < (257u << KEY_IDX_BITS)) {
            b1 = umin_(b1, wide(s0 >> 16, (b0 & KEY_IDX_MASK) >> 6));
            if (b1 >= (257u << KEY_IDX_BITS)) b1 = KEY_NONE;
        } else {
            b0 = b1 = KEY_NONE;
        }
    }
    if (PARTS > 1) {
        red[2 * threadIdx.x] = b0;
        red[2 * threadIdx.x + 1] = b1;
        __syncthreads();
        if (part_id == 0) {
#pragma unroll
            for (int q = 1; q < PARTS; ++q) merge2(b0, b1, red[2 * (q * COLS + jl)], red[2 * (q * COLS + jl) + 1]);
        }
    }
    if (part_id == 0 && j < sd.n2) ((gu2_t) reinterpret_cast<u32x2_t*>(sd.keys21))[j] = u32x2_t{b0, b1};
}

// K1c'' + K2 in ONE kernel behind a column-split K1f scan (C3: one local map against one frame; mapHandler.cpp:532-752): the
// plan run is two launches.  The column side decides.  A workgroup merges the partials of its columns exactly as
// k_merge_partials16 does; the lane that then holds column j's pair applies the column's ratio test -> i* (the only row that can
// be consistent with j), merges row i*'s per-range results (the finalize kernel's best2 over `nsplit` tables, column indices
// relative to ranges of `cstep` columns), applies the row's ratio test -> m, and writes matches_12[i*] = j iff m == j: the set
// {(i, m) : m21[m] == i} of stvo-pl's match() read from the other side (every row has at most one m, every column at most
// one i*).  Rows without a match hold the -1 the scan's first column range left there; the count is the number of pairs.
// Only for mutual problems without keep_prior and without a stereo gate (plan_build / add_stereo_gates decide).
// SymDesc::mutual - 1 = the problem's index; the range's first column = (sd.b - p.d2) / 32.
template <int PARTS>
__global__ void __launch_bounds__(256)
k_split_post(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks, const ProblemDesc* __restrict__ probs)
{
    constexpr int COLS = 256 / PARTS;
    __shared__ uint32_t red[PARTS > 1 ? 512 : 2];
    const BlockDesc bd = blocks[blockIdx.x];
    const SymDesc sd = syms[bd.item];
    const ProblemDesc p = probs[sd.mutual - 1];
    const int jl = (int)threadIdx.x % COLS, part_id = (int)threadIdx.x / COLS;
    const int j = bd.row0 + jl;
    const gcu32_t part = (gcu32_t) sd.part21;
    const int n1 = sd.n1_dev ? min(sd.n1, *(const PLSLAM_GLOBAL int32_t*) sd.n1_dev) : sd.n1;   // (the scan wrote the partials of these rows only)
    const int nwb = (n1 + 63) >> 6;
    const int n2p = (sd.n2 + 255) & ~255;
    uint32_t b0 = KEY_NONE, b1 = KEY_NONE;
    const uint32_t tw = ((uint32_t)j >> 5) & 63u;
    if (j < sd.n2) {
        auto wide = [](uint32_t k16, uint32_t wb) -> uint32_t { return ((k16 << 16) & 0xFF800000u) | ((k16 & 63u) + 64u * wb); };
        const uint32_t tw2 = tw | (tw << 16);
        uint32_t s0 = 0xFFFFFFFFu;
#pragma unroll 8
        for (int wb = part_id; wb < nwb; wb += PARTS) {
            const uint32_t e = part[(size_t)wb * n2p + j] - tw2;
            const uint32_t k = wide(e & 0xFFFFu, (uint32_t)wb);
            s0 = k < b0 ? e : s0;
            b1 = umin_(b1, umax_(b0, k));
            b0 = umin_(b0, k);
        }
        if (b0 < (257u << KEY_IDX_BITS)) {
            b1 = umin_(b1, wide(s0 >> 16, (b0 & KEY_IDX_MASK) >> 6));
            if (b1 >= (257u << KEY_IDX_BITS)) b1 = KEY_NONE;
        } else {
            b0 = b1 = KEY_NONE;
        }
    }
    if (PARTS > 1) {
        red[2 * threadIdx.x] = b0;
        red[2 * threadIdx.x + 1] = b1;
        __syncthreads();
        if (part_id == 0) {
#pragma unroll
            for (int q = 1; q < PARTS; ++q) merge2(b0, b1, red[2 * (q * COLS + jl)], red[2 * (q * COLS + jl) + 1]);
        }
    }
    // stvo-pl matchNNR: accept iff (float)d0 < (float)d1 * nnr (one fp32 multiply); fewer than two neighbours: no match
    auto ratio_pick = [&](uint32_t q0, uint32_t q1) -> int {
        if (q1 == KEY_NONE) return -1;
        const float d0 = (float)(q0 >> KEY_IDX_BITS);
        const float d1n = __fmul_rn((float)(q1 >> KEY_IDX_BITS), p.nnr);
        return d0 < d1n ? (int)(q0 & KEY_IDX_MASK) : -1;
    };
    bool pair = false;
    if (part_id == 0 && j < sd.n2) {
        ((gu2_t) reinterpret_cast<u32x2_t*>(sd.keys21))[j] = u32x2_t{b0, b1};          // diagnostics (plslam_match_plan_dump)
        const int istar = ratio_pick(b0, b1);
        if (istar >= 0) {
            const int jg = j + (int)((sd.b - p.d2) >> 5);                               // the column within the problem
            const int ns = p.nsplit > 1 ? p.nsplit : 1;
            const auto tmp = (const PLSLAM_GLOBAL u32x2_t*) reinterpret_cast<const u32x2_t*>(p.nsplit > 1 ? p.split_tmp : p.keys12);
            uint32_t r0 = KEY_NONE, r1 = KEY_NONE;
#pragma unroll 4
            for (int s = 0; s < ns; ++s) {
                const u32x2_t q = tmp[(size_t)s * p.n1 + istar];
                const uint32_t off = (uint32_t)(s * p.cstep);
                merge2(r0, r1, q.x == KEY_NONE ? KEY_NONE : q.x + off, q.y == KEY_NONE ? KEY_NONE : q.y + off);
            }
            pair = ratio_pick(r0, r1) == jg;
            if (pair) ((PLSLAM_GLOBAL int32_t*) p.matches_12)[istar] = jg;
        }
    }
    if (p.n_matches) {
        const int found = (int)__popcll(__ballot(pair));
        if ((threadIdx.x & 63) == 0 && found) (void)atomic_add_global(p.n_matches, found);
    }
}

// plslam_match_plan_dump of a two-launch column-split plan: the rows' merged pairs (what the finalize kernel of the three-launch
// form stores to keys12_out on its way), a lane per row
__global__ void __launch_bounds__(256)
k_split_rows_dump(const ProblemDesc* __restrict__ probs, const BlockDesc* __restrict__ blocks)
{
    const BlockDesc bd = blocks[blockIdx.x];
    if (bd.item < 0) return;                               // (padding entry of a dealt table)
    const ProblemDesc p = probs[bd.item];
    const int i1 = bd.row0 + (int)threadIdx.x;
    if (i1 >= p.n1 || p.nsplit <= 1) return;
    const auto tmp = (const PLSLAM_GLOBAL u32x2_t*) reinterpret_cast<const u32x2_t*>(p.split_tmp);
    uint32_t r0 = KEY_NONE, r1 = KEY_NONE;
    for (int s = 0; s < p.nsplit; ++s) {
        const u32x2_t q = tmp[(size_t)s * p.n1 + i1];
        const uint32_t off = (uint32_t)(s * p.cstep);
        merge2(r0, r1, q.x == KEY_NONE ? KEY_NONE : q.x + off, q.y == KEY_NONE ? KEY_NONE : q.y + off);
    }
    ((gu2_t) reinterpret_cast<u32x2_t*>(p.keys12_out))[i1] = u32x2_t{r0, r1};
}

int merge_partials16_cols(int parts) { return 256 / (parts >= 16 ? 16 : parts >= 4 ? 4 : 1); }

// d_blocks: the merge kernel's table, one entry per (sub-problem, merge_partials16_cols(parts) columns)
int launch_split_post(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int parts, const ProblemDesc* d_probs, hipStream_t s)
{
    if (nblocks <= 0) return PLSLAM_OK;
    if (parts >= 16) hipLaunchKernelGGL((k_split_post<16>), dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks, d_probs);
    else if (parts >= 4) hipLaunchKernelGGL((k_split_post<4>), dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks, d_probs);
    else hipLaunchKernelGGL((k_split_post<1>), dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks, d_probs);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int launch_split_rows_dump(const ProblemDesc* d_probs, const BlockDesc* d_fin_blocks, int nblocks, hipStream_t s)
{
    if (nblocks <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_split_rows_dump, dim3(nblocks), dim3(256), 0, s, d_probs, d_fin_blocks);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

// d_blocks: one entry per (problem, merge_partials16_cols(parts) columns)
int launch_merge_partials16(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int parts, hipStream_t s)
{
    if (nblocks <= 0) return PLSLAM_OK;
    if (parts >= 16) hipLaunchKernelGGL((k_merge_partials16<16>), dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks);
    else if (parts >= 4) hipLaunchKernelGGL((k_merge_partials16<4>), dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks);
    else hipLaunchKernelGGL((k_merge_partials16<1>), dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int launch_scan_sym_mfma_g(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero,
                           int nzero, bool multi_window, bool directed, bool fused, hipStream_t s)
{
    if (nblocks <= 0) return PLSLAM_OK;
#define PLSLAM_MG_LAUNCH(M, D, F) \
    hipLaunchKernelGGL((k_scan_sym_mfma_g<M, D, F>), dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks, d_zero, nzero)
#define PLSLAM_MG_LAUNCH2(M, D) { if (fused) PLSLAM_MG_LAUNCH(M, D, true); else PLSLAM_MG_LAUNCH(M, D, false); }
    if (multi_window) { if (directed) PLSLAM_MG_LAUNCH2(true, true) else PLSLAM_MG_LAUNCH2(true, false) }
    else              { if (directed) PLSLAM_MG_LAUNCH2(false, true) else PLSLAM_MG_LAUNCH2(false, false) }
#undef PLSLAM_MG_LAUNCH2
#undef PLSLAM_MG_LAUNCH
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

}  // namespace plslam
