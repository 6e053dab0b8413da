// hamming_mfma.hip -- K1e: symmetric 256-bit Hamming kNN-2 scan on the matrix cores (gfx950).
//
// Same contract as K1b/K1b' (hamming.hip): for a mutual problem (a: n1 rows, b: n2 rows) produce
// keys12[i] = best-2 over j and the column partials part21[i-block][j] = best-2 over the block's i,
// keys = (distance << 23) | index, i.e. cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) order in both
// directions (reference call sites src/mapHandler.cpp:277,424,597,712,3223,3249).
//
// The Hamming distance of two bit rows IS a contraction: with s(x) = 1 - 2x in {+1,-1},
//     sum_k s(a_k) s(b_k) = 256 - 2 d(a,b).
// Both sides are expanded to fp4 (e2m1) codes of +-1 -- A rows to -s(a) with the block scale 2^6, B rows to
// s(b) -- the accumulator starts at 2^23 + 16384 + tag, and four v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64
// each) leave 2^23 + 128 d + tag: every partial sum is an integer below 2^24, so the fp32 accumulation is
// exact, the low 16 bits of the float ARE the 16-bit key (d << 7 | tag), and the match tables are
// bit-identical to the XOR+popcount kernels.  (The first version used v_mfma_i32_32x32x32_i8 over +-1 bytes:
// twice the MFMAs, twice the LDS operand reads -- which measured 19 % of the kernel -- and 32 more VGPRs.)
// Per 32x32 tile a wave issues 4 MFMAs (the matrix pipe) instead of 16 x 16 VALU ops per lane; what
// stays on the VALU is the best-2 bookkeeping: 8 packed 16-bit ops per 2 elements for both directions.
//
// Work decomposition = K1b': one workgroup per 256 rows of `a` (same block tables, same partial
// table, same merge + finalize kernels).  4 waves; wave w keeps its 64 rows (2 M-tiles) expanded in
// 32 VGPRs for the whole scan.  `b` is streamed in tiles of 32 rows: the 256 lanes expand one raw
// dword each (32 bits -> 32 fp4 codes, through a 256-entry byte -> 8-code table in LDS) into a
// double-buffered LDS tile whose 144-byte row stride makes the ds_read_b128 operand reads
// conflict-free.  Only the lane->k mapping shared by the A and the B operand matters for a
// contraction over all k, so no assumption about the instruction's internal k order is made.
// C/D layout (dtype-independent): col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
//   row direction : lane keeps best-2 per (M-tile, reg) over the columns it sees (j = lane & 31 mod 32) as
//                   packed 16-bit keys (d << 7 | tile + LOC); per window of 64 tiles the 32 column classes of
//                   a row are combined through an LDS transpose (one lane per row) and merged into keys12.
//   column direction: best-2 over the lane's 32 accumulators (the key comes out of the MFMA: tag = local
//                   row), halves and lane ^ 32 combined in the 16-bit domain, the four waves through LDS,
//                   one partial per 256 rows of `a`, as K1b'.  Not built in the DIRECTED instantiation.
#include "common.hpp"

#include <type_traits>

// build-time experiments for tools/scan_time.py (results are WRONG with any of them on):
//   1 = no workgroup barrier, 2 = no epilogue, 3 = no MFMA
#ifndef PLSLAM_MF_EXPERIMENT
#define PLSLAM_MF_EXPERIMENT 0
#endif

namespace plslam {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4), aligned(4)));   // descriptor rows are only 4-byte aligned
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Pointers read from the launch tables are GENERIC to the compiler, and a generic access is a FLAT instruction, which
// counts on lgkmcnt as well as vmcnt: the `s_waitcnt lgkmcnt(0)` in front of every workgroup barrier then waits for
// the raw-row PREFETCH of two tiles ahead (round 2 finding: every tile paid a memory latency).  With the address
// space spelled out the loads are global_load (vmcnt only) and stay in flight across the barrier.
#define PLSLAM_GLOBAL __attribute__((address_space(1)))
typedef const PLSLAM_GLOBAL uint32_t* gcu32_t;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef PLSLAM_GLOBAL u32x2_t* gu2_t;

namespace {

constexpr int MF_TILE_N = 32;                 // b rows per tile
constexpr int MF_KSTEPS = 4;                  // 256 bits = 4 x K 64
constexpr int MF_ROW_STRIDE = 144;            // bytes per expanded b row in LDS (128 + 16: 4-bank skew)
constexpr int MF_TILE_BYTES = MF_TILE_N * MF_ROW_STRIDE;
// fp4 (e2m1) codes: +1.0 = 0x2, -1.0 = 0xA.  b side: bit 0 -> +1, bit 1 -> -1 = s(b); the a side is the b code
// XOR 0x8 per nibble (= -s(a)) and carries the block scale 2^6 (E8M0 133), the b side 2^0 (E8M0 127).
constexpr uint32_t FP4_NEG = 0x88888888u;
constexpr int SCALE_A = 133, SCALE_B = 127;
constexpr float ACC_MAGIC = 8388608.0f;       // 2^23: float bits = 0x4B000000 + integer part

__device__ __forceinline__ uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax_(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ void merge2(uint32_t& a0, uint32_t& a1, uint32_t c0, uint32_t c1)
{
    const uint32_t lo = umin_(a0, c0);
    const uint32_t hi = umin_(umax_(a0, c0), umin_(a1, c1));
    a0 = lo;
    a1 = hi;
}
// packed 16-bit min / max.  Inline asm on purpose: written with the vector builtins the compiler sinks
// the row-direction pushes out of the MFMA block into a block of their own (96 VALU ops with no MFMA to
// hide under, 6 spilled VGPRs; measured 6.10 ms vs 5.85 ms).
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
// two sorted streams of 16-bit keys, one per half of the register
__device__ __forceinline__ void pk_push2(uint32_t& b0, uint32_t& b1, uint32_t key)
{
    b1 = pk_min16(b1, pk_max16(b0, key));
    b0 = pk_min16(b0, key);
}
// accumulators of the two M-tiles (2^23 + 128 d + tag, tag <= 127) side by side: hi.lo16 << 16 | lo.lo16
__device__ __forceinline__ uint32_t pack_acc(float lo, float hi, uint32_t sel_uniform /* 0x05040100 */)
{
    // The BUILTIN, not inline asm: this is the one instruction that reads MFMA results directly, and on gfx950 the
    // wait states between an MFMA and a VALU access to its destination registers are the COMPILER's job (s_nop); its
    // hazard recognizer does not look inside asm statements.  As `asm("v_perm_b32 ...")` the first two packs of the
    // unpipelined epilogue issued right behind the last MFMA of the set: accumulators 0 and 1 (tile rows 0, 1, 4, 5)
    // were read -- and register 0 overwritten -- while still in flight.  Right most of the time, wrong when waves of
    // co-resident workgroups delayed the matrix pipe: ~1 % of the intermediate keys of a loaded batch differed from
    // run to run, 1e-6 of the table entries at a 0.9 ratio ("K1e determinism";
    // tools/determinism_check.py is the instrument).  The v_pk_min/max asm below only ever sees pack_acc's result.
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), sel_uniform);
}
// 16-bit keys are (d << 7) | tag7: the A bytes are -64 s(a), so with C = 16384 + tag the accumulator
// itself is 128 d + tag (<= (256 << 7) + 127 = 0x807F); anything above is "none"
constexpr uint32_t KEY16_MAX = 0x807Fu;
__device__ __forceinline__ uint32_t key16_to_key32(uint32_t k16, uint32_t tag_bias, uint32_t idx_base,
                                                   uint32_t idx_scale)
{
    return k16 > KEY16_MAX ? KEY_NONE
                           : (((k16 >> 7) << KEY_IDX_BITS) | (idx_base + ((k16 & 127u) - tag_bias) * idx_scale));
}
// one byte of a descriptor -> 8 fp4 codes of s(bit): bit k -> nibble k = 0x2 | (bit << 3)
__device__ __forceinline__ uint32_t expand_byte_fp4(uint32_t byte)
{
    uint32_t x = (byte | (byte << 12)) & 0x000F000Fu;      // 4 + 4 bits
    x = (x | (x << 6)) & 0x03030303u;                      // 2 bits per byte
    x = (x | (x << 3)) & 0x11111111u;                      // 1 bit per nibble (at bit 0)
    return (x << 3) | 0x22222222u;
}
__device__ __forceinline__ int xcd_remap_(int orig, int nwg) { return (orig & 7) * (nwg >> 3) + (orig >> 3); }

}  // namespace

// MULTI = false: every problem of the launch has n2 <= 2048 (one window of 64 tiles; the window bounds are
// compile-time facts).  MULTI = true: any n2 (the extra live state costs ~2 % through register pressure,
// which is why the common case has its own instantiation; capi.hip picks per plan).
// DIRECTED = true: only keys12 (row direction) is produced -- non-mutual problems and plain knnMatch(k=2):
// no column keys, no column partials, 5 VALU ops per 2 distances.
// experiment: -DPLSLAM_MF_SINGLE_SET=1 -> one accumulator set, no M(t)/E(t-1) overlap inside a wave, 4 waves per SIMD
#ifndef PLSLAM_MF_SINGLE_SET
#define PLSLAM_MF_SINGLE_SET 0
#endif
template <bool MULTI, bool DIRECTED>
#ifndef PLSLAM_MF_WAVES
#define PLSLAM_MF_WAVES 3
#endif
__global__ void __launch_bounds__(256, PLSLAM_MF_SINGLE_SET ? 4 : PLSLAM_MF_WAVES)      // 3 waves per SIMD: <= 168 unified VGPRs
k_scan_sym_mfma(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks,
                int32_t* __restrict__ zero, int nzero)
{
    // one buffer, two lives: the double-buffered b tile during the scan (9 216 B), the row-result transpose
    // [wave][row 0..63][33] after it (33 792 B)
    constexpr int ROWX_STRIDE = 33;               // dwords per row: lane = row reads are conflict-free
    __shared__ __attribute__((aligned(16))) uint8_t smem[4 * 64 * ROWX_STRIDE * 4];
    uint8_t* const btile = smem;
    __shared__ uint32_t colbuf[2][4][MF_TILE_N];  // [tile & 1][wave][column] = best | second << 16    1 024 B
    __shared__ uint32_t blut[256];                // byte of a b row -> its 8 s(b) fp4 codes  1 024 B

    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nzero; i += 256) zero[i] = 0;

    const int wg = xcd_remap_(blockIdx.x, gridDim.x);
    const BlockDesc bd = blocks[wg];
    if (bd.item < 0) return;                       // padding entry of the XCD-striped table
    const SymDesc sd = syms[bd.item];
    const int n1 = sd.n1, n2 = sd.n2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, g = lane >> 5;
    const int i0 = bd.row0;                        // first of this workgroup's 256 a-rows
    const int iw = i0 + 64 * w;                    // first of this wave's 64
    const gcu32_t araw = (gcu32_t) reinterpret_cast<const uint32_t*>(sd.a);
    const gcu32_t braw = (gcu32_t) reinterpret_cast<const uint32_t*>(sd.b);

    // expansion table first (the A operands use it too)
    blut[tid] = expand_byte_fp4((uint32_t)tid);
    __syncthreads();
    // ---- A operands: rows iw + 32 mt + c, bits [64 ks + 32 g, +32) of each = raw dword 2 ks + g, as
    // fp4 codes of -s(a) (the b code with the sign nibble-bit flipped); the factor 64 is the block scale ----
    i32x4 afrag[2][MF_KSTEPS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = iw + 32 * mt + c;
        const int rrow = row < n1 ? row : n1 - 1;                 // clamped; masked in the epilogue
        const gcu32_t p = araw + (size_t)rrow * 8 + g;           // this lane's dword of each K-step: 2 ks + g
#pragma unroll
        for (int ks = 0; ks < MF_KSTEPS; ++ks) {
            const uint32_t raw = p[2 * ks];
            afrag[mt][ks].x = (int)(blut[raw & 0xFFu] ^ FP4_NEG);
            afrag[mt][ks].y = (int)(blut[(raw >> 8) & 0xFFu] ^ FP4_NEG);
            afrag[mt][ks].z = (int)(blut[(raw >> 16) & 0xFFu] ^ FP4_NEG);
            afrag[mt][ks].w = (int)(blut[raw >> 24] ^ FP4_NEG);
        }
    }
    // accumulator start: 2^23 + 16384 + LOC(reg): the sum is 2^23 + 128 d + LOC, every partial sum an
    // integer below 2^24, so fp32 accumulation is exact and the float's low 16 bits ARE the column key
    // (d << 7 | LOC) of the element (LOC = (reg & 3) + 8 (reg >> 2) = its local row within the half-tile)
    f32x16 cinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[r] = ACC_MAGIC + 16384.0f + (float)((r & 3) + 8 * (r >> 2));
    const int scale_a = SCALE_A, scale_b = SCALE_B;
    const uint32_t pack_sel = 0x05040100u;

    // row-direction state: per accumulator register r, the best two 16-bit keys (d << 6 | tile) of the
    // lane's column class, M-tile 0 in the low halves and M-tile 1 in the high halves
    uint32_t rb[16][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) rb[r][0] = rb[r][1] = 0xFFFFFFFFu;

    const bool rows_ragged = iw + 64 > n1;         // wave-uniform: some of this wave's rows do not exist
    const uint32_t ibase = (uint32_t)(iw + 4 * g); // + local index = a-row of an accumulator
    const uint32_t ghtag = (uint32_t)(4 * g) | ((uint32_t)(4 * g + 32) << 16);   // see finish_columns
    const gu2_t part = (gu2_t) reinterpret_cast<u32x2_t*>(sd.part21) + (size_t)(i0 >> 8) * n2;

    // expansion duty of this lane: b row (tid >> 3) of the tile, dword (tid & 7) of it
    const int ej = tid >> 3, ewd = tid & 7;
    const int ntiles = (n2 + MF_TILE_N - 1) / MF_TILE_N;
    auto load_raw = [&](int t) __attribute__((always_inline)) -> uint32_t {
        int j = t * MF_TILE_N + ej;
        j = j < n2 ? j : n2 - 1;
        return braw[(size_t)j * 8 + ewd];
    };
    // b-side expansion goes through the table: 4 ds_read_b32 + one ds_write_b128 per raw dword
    auto expand_store = [&](uint32_t raw, int buf) __attribute__((always_inline)) {
        uint8_t* dst = btile + buf * MF_TILE_BYTES + ej * MF_ROW_STRIDE + ewd * 16;
        i32x4 v;
        v.x = (int)blut[raw & 0xFFu];
        v.y = (int)blut[(raw >> 8) & 0xFFu];
        v.z = (int)blut[(raw >> 16) & 0xFFu];
        v.w = (int)blut[raw >> 24];
        *reinterpret_cast<i32x4*>(dst) = v;
    };
    // lanes 0..31 of ONE wave: widen the 4 waves' 16-bit column results of tile t (tag = row within the wave)
    // to (d << 23 | a-row), combine, and write the workgroup's partial
    auto flush_columns = [&](int t) __attribute__((always_inline)) {
        if (!DIRECTED && lane < MF_TILE_N) {
            uint32_t k0 = KEY_NONE, k1 = KEY_NONE;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) {
                const uint32_t e = colbuf[t & 1][ww][lane];
                merge2(k0, k1, key16_to_key32(e & 0xFFFFu, 0u, (uint32_t)(i0 + 64 * ww), 1u),
                       key16_to_key32(e >> 16, 0u, (uint32_t)(i0 + 64 * ww), 1u));
            }
            const int j = t * MF_TILE_N + lane;
            if (j < n2) part[j] = u32x2_t{k0, k1};
        }
    };

    // Epilogue of one accumulator register pair (rows LOC and LOC + 32 of the wave, column j0 + c).
    // MASKED = false is the steady state (every row of this wave and every column of the tile exists);
    // the ragged cases are separate instantiations OUTSIDE the steady-state loop.
    // 8 VALU ops for the two distances: pack (= the column keys), + tile (= the row keys; every candidate
    // of rb[R] carries the same LOC, so (d, tile + LOC) orders like (d, tile)), 3 + 3 to push them.
#define PLSLAM_MF_EPI_ROW(R)                                                                       \
    {                                                                                              \
        constexpr uint32_t LOC = ((R) & 3) + 8 * ((R) >> 2);                                       \
        uint32_t kc = pack_acc(acc0[R], acc1[R], pack_sel);                                        \
        uint32_t kr = kc + tpair;                                                                  \
        if (MASKED) {                                                                              \
            kr = col_ok ? kr : 0xFFFFFFFFu;                                                        \
            kc |= ((int)(ibase + LOC) < n1 ? 0u : 0x0000FFFFu) |                                   \
                  ((int)(ibase + LOC + 32u) < n1 ? 0u : 0xFFFF0000u);                              \
        }                                                                                          \
        pk_push2(rb[R][0], rb[R][1], kr);                                                          \
        if (!DIRECTED) pk_push2(cb0, cb1, kc);                                                     \
    }
    {
#define WT0 (MULTI ? wt0v : 0)
#define WT1 (MULTI ? wt1v : ntiles)
    // The 16-bit row keys hold 64 tile numbers, so the scan runs in WINDOWS of 64 tiles (2048 columns): after
    // each window the row state is reduced and merged into keys12, then restarted.  WT0 = first tile of the
    // current window, WT1 = one past its last.  In the MULTI = false instantiation both are
    // compile-time facts (0 and ntiles): no extra live registers.
    int wt0v = 0, wt1v = ntiles < 64 ? ntiles : 64;
    uint32_t raw1 = 0u;                                    // raw b dword of tile t+1 of the coming step
    // column best-2 of a finished tile: the two halves of (cb0, cb1) are sorted streams over disjoint rows
    // of the same column -> best 2 of the lane, then of the wave (lane ^ 32), as 16-bit keys
    auto finish_columns = [&](int t, uint32_t cb0, uint32_t cb1) __attribute__((always_inline)) {
        if (DIRECTED) return;
        // Stay in the 16-bit domain: the tag of a column key is LOC (bits 0,1,3,4 of the row within the wave);
        // OR-ing in bit 2 (= g) and bit 5 (= M-tile, the high halves) makes it the full row within the wave,
        // so keys of the two halves and of lane ^ 32 compare directly (0xFFFF stays 0xFFFF).
        cb0 |= ghtag;
        cb1 |= ghtag;
        const uint32_t e0 = cb0 & 0xFFFFu, o0 = cb0 >> 16, e1 = cb1 & 0xFFFFu, o1 = cb1 >> 16;
        uint32_t m0 = umin_(e0, o0), m1 = umin_(umax_(e0, o0), umin_(e1, o1));
        const uint32_t other = (uint32_t)__shfl_xor((int)(m0 | (m1 << 16)), 32);
        merge2(m0, m1, other & 0xFFFFu, other >> 16);
        if (lane < MF_TILE_N) colbuf[t & 1][w][lane] = m0 | (m1 << 16);
    };
    // E(t) on its own (the last tile has no following M step to hide under)
    auto epilogue = [&](int t, const f32x16& acc0, const f32x16& acc1, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const bool col_ok = t * MF_TILE_N + c < n2;
        const uint32_t tpair = (uint32_t)(t - WT0) * 0x00010001u;    // tile number within the window, both halves
        uint32_t cb0 = 0xFFFFFFFFu, cb1 = 0xFFFFFFFFu;
        PLSLAM_MF_EPI_ROW(0) PLSLAM_MF_EPI_ROW(1) PLSLAM_MF_EPI_ROW(2) PLSLAM_MF_EPI_ROW(3)
        PLSLAM_MF_EPI_ROW(4) PLSLAM_MF_EPI_ROW(5) PLSLAM_MF_EPI_ROW(6) PLSLAM_MF_EPI_ROW(7)
        PLSLAM_MF_EPI_ROW(8) PLSLAM_MF_EPI_ROW(9) PLSLAM_MF_EPI_ROW(10) PLSLAM_MF_EPI_ROW(11)
        PLSLAM_MF_EPI_ROW(12) PLSLAM_MF_EPI_ROW(13) PLSLAM_MF_EPI_ROW(14) PLSLAM_MF_EPI_ROW(15)
        finish_columns(t, cb0, cb1);
    };
    // One pipeline step = M(t) fused with E(t-1):
    //   M(t): barrier, then the 16 MFMAs of tile t into (m0, m1); the raw dwords of tile t+2 are requested
    //         and tile t+1 (requested one step earlier: its latency is off the critical path) is expanded;
    //   E(t-1): best-2 bookkeeping of tile t-1 from ITS accumulators (acc0, acc1) -- independent of M(t).
    // Measured alternatives to the order below, all 5.8-5.9 ms or worse (this one: 5.78 ms): MFMAs in pairs
    // with the scheduler placing the VALU work, with or without sched_group_barrier patterns (2 MFMA : 22 / 30
    // VALU, 1 : 11); all eight B operands prefetched before the first MFMA; co-resident workgroups started
    // half a step apart; pairs of MFMAs hand-fenced with two register pairs of E (+4 %); LDS round trips
    // (operand reads, lane exchange) issued a quarter-epilogue ahead of their use (+8 %).
    auto step = [&](int t, f32x16& m0, f32x16& m1, const f32x16& acc0, const f32x16& acc1, bool with_prev,
                    auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const uint32_t raw2 = t + 2 < WT1 ? load_raw(t + 2) : 0u;
        if (PLSLAM_MF_EXPERIMENT != 1) __syncthreads();   // tile t expanded; colbuf of tile t-2 complete
        if (t - WT0 > 1 && w == (t & 3)) flush_columns(t - 2);      // the waves take turns
        const uint8_t* bt = btile + (t & 1) * MF_TILE_BYTES + c * MF_ROW_STRIDE + 16 * g;
        const bool col_ok = (t - 1) * MF_TILE_N + c < n2;
        const uint32_t tpair = (uint32_t)(t - 1 - WT0) * 0x00010001u; // tile number within the window, both halves
        uint32_t cb0 = 0xFFFFFFFFu, cb1 = 0xFFFFFFFFu;
        i32x4 bf = *reinterpret_cast<const i32x4*>(bt);
        // program order, fenced: [MFMA] [two accumulator register pairs of E(t-1): 16 VALU] [MFMA] [16 VALU]
        // ... -- single MFMAs, evenly spaced, so that this wave is never stalled at the issue of the second
        // MFMA of a pair while the matrix pipe is busy with the first
#define PLSLAM_MF_MMA(ACC, MT, KS, CIN)                                                            \
        {                                                                                          \
            const i32x8 a8 = {afrag[MT][KS].x, afrag[MT][KS].y, afrag[MT][KS].z, afrag[MT][KS].w, 0, 0, 0, 0}; \
            const i32x8 b8 = {bcur.x, bcur.y, bcur.z, bcur.w, 0, 0, 0, 0};                         \
            if (PLSLAM_MF_EXPERIMENT != 3)                                                         \
                ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, CIN, 4, 4, 0, scale_a, 0, scale_b); \
            else ACC[KS] = __builtin_bit_cast(float, bcur.x);                                      \
        }
#define PLSLAM_MF_KSTEP(KS, CIN0, CIN1)                                                            \
        {                                                                                          \
            const i32x4 bcur = bf;                                                                 \
            if ((KS) < MF_KSTEPS - 1) bf = *reinterpret_cast<const i32x4*>(bt + 32 * ((KS) + 1));   \
            PLSLAM_MF_MMA(m0, 0, KS, CIN0)                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (with_prev && PLSLAM_MF_EXPERIMENT != 2) { PLSLAM_MF_EPI_ROW(4 * (KS)) PLSLAM_MF_EPI_ROW(4 * (KS) + 1) } \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            PLSLAM_MF_MMA(m1, 1, KS, CIN1)                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                     \
            if (with_prev && PLSLAM_MF_EXPERIMENT != 2) { PLSLAM_MF_EPI_ROW(4 * (KS) + 2) PLSLAM_MF_EPI_ROW(4 * (KS) + 3) } \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
        PLSLAM_MF_KSTEP(0, cinit, cinit)
        PLSLAM_MF_KSTEP(1, m0, m1) PLSLAM_MF_KSTEP(2, m0, m1) PLSLAM_MF_KSTEP(3, m0, m1)
#undef PLSLAM_MF_KSTEP
#undef PLSLAM_MF_MMA
        expand_store(raw1, (t + 1) & 1);           // past the last tile: a harmless rewrite of the idle buffer
        raw1 = raw2;
        if (PLSLAM_MF_EXPERIMENT == 2) asm volatile("" ::"v"(m0), "v"(m1));
        if (with_prev && PLSLAM_MF_EXPERIMENT != 2) finish_columns(t - 1, cb0, cb1);
    };
    // One window: S(WT0) | S(WT0+1)+E(WT0) | S(WT0+2)+E(WT0+1) | ... | E(WT1-1).  Two accumulator sets
    // alternate (unrolled by two: no accumulator is ever copied).  Only the last tile of the scan can lack
    // columns.
    auto pipeline = [&](auto steady_tag) __attribute__((always_inline)) {
        const bool last_partial = WT1 == ntiles && (n2 % MF_TILE_N) != 0;
#if PLSLAM_MF_SINGLE_SET
        f32x16 A0, A1;
        for (int t = WT0; t < WT1; ++t) {
            step(t, A0, A1, A0, A1, false, steady_tag);
            __syncthreads();                       // the flush of tile t-2 (same colbuf parity as tile t) is done
            if (t == WT1 - 1 && last_partial) epilogue(t, A0, A1, std::true_type{}); else epilogue(t, A0, A1, steady_tag);
        }
        return;
#else
        f32x16 A0, A1, B0, B1;
        step(WT0, A0, A1, A0, A1, false, steady_tag);
        int t = WT0 + 1;
        for (; t + 1 < WT1; t += 2) {
            step(t, B0, B1, A0, A1, true, steady_tag);
            step(t + 1, A0, A1, B0, B1, true, steady_tag);
        }
        // The trailing epilogue E(WT1-1) writes colbuf[(WT1-1) & 1] -- the buffer the flush of tile WT1-3 READS at the
        // start of the last step (one wave, right behind that step's barrier).  In the steady state a barrier lies between
        // a tile's flush and the next write of its buffer; here it must be added, or a wave that races through its last
        // step can overwrite its slot before a delayed flusher (e.g. an instruction-cache miss in the rarely executed
        // flush path) has read it: rare wrong column partials (seen as 1 in ~10^6 table entries under load).
        // Loop exit.  The step that just ended issued MFMAs whose destination registers are dead on some of the paths
        // below (a window with nothing left to consume them): the allocator reuses those registers at once, and on the
        // shortest such path tools/check_mfma_hazards.py counts only 8 wait states between the last v_mfma and a VALU
        // write to one of its destination registers -- fewer than the 11-12 the compiler puts in front of its own
        // accumulator reads.  Never seen to misbehave; 16 idle cycles once per window make it a non-question.
        asm volatile("s_nop 7\n\ts_nop 7");
        if (t < WT1) {                             // t == WT1 - 1: one more tile, into set B
            step(t, B0, B1, A0, A1, true, steady_tag);
            __syncthreads();
            if (last_partial) epilogue(t, B0, B1, std::true_type{}); else epilogue(t, B0, B1, steady_tag);
        } else {                                   // tile WT1 - 1 is in set A
            __syncthreads();
            if (last_partial) epilogue(t - 1, A0, A1, std::true_type{}); else epilogue(t - 1, A0, A1, steady_tag);
        }
#endif
    };
    // Row results of a window.  Every lane holds, per accumulator register, the best two 16-bit keys
    // (d, tile + LOC) of ITS column class for two rows.  Transpose through LDS so that one lane owns one row:
    // lane l reads the 32 class entries of row l in class order, widens them to (key16 << 16 | class) -- which
    // orders like (d, j = 32 tile + class) because every entry of a row carries the same LOC -- and keeps the
    // best two; only those two are converted to (d << 23 | j).  (A 5-step cross-lane butterfly per register
    // cost ~860 VALU ops + 320 ds_bpermute per wave; this is ~250 + 64.)  Callers guarantee that all waves
    // are past their last operand read of `smem`; the region used here is private to the wave.
    auto finish_rows = [&]() __attribute__((always_inline)) {
        uint32_t* rowx = reinterpret_cast<uint32_t*>(smem) + w * (64 * ROWX_STRIDE);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = (r & 3) + 8 * (r >> 2) + 4 * g;
            // (best | second << 16) of M-tile 0 (low halves) and of M-tile 1 (high halves)
            rowx[lrow * ROWX_STRIDE + c] = (rb[r][0] & 0xFFFFu) | (rb[r][1] << 16);
            rowx[(32 + lrow) * ROWX_STRIDE + c] = (rb[r][0] >> 16) | (rb[r][1] & 0xFFFF0000u);
            rb[r][0] = rb[r][1] = 0xFFFFFFFFu;                  // restart for the next window
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
            if (WT0 > 0) {                                  // later windows: merge with the windows before
                const u32x2_t prev = *out;
                merge2(r0, r1, prev.x, prev.y);
            }
            *out = u32x2_t{r0, r1};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    for (;;) {
        expand_store(load_raw(WT0), 0);            // WT0 is a multiple of 64: buffer parity restarts at 0
        raw1 = WT0 + 1 < WT1 ? load_raw(WT0 + 1) : 0u;
        if (!rows_ragged) pipeline(std::false_type{}); else pipeline(std::true_type{});
        // the last two tiles' column partials of the window are still in LDS
        __syncthreads();
        if (WT1 - WT0 > 1 && w == (WT1 & 3)) flush_columns(WT1 - 2);
        if (w == ((WT1 + 1) & 3)) flush_columns(WT1 - 1);
        finish_rows();
        if (!MULTI || wt1v == ntiles) break;
        __syncthreads();                           // smem becomes the b tile again; colbuf is free
        wt0v = wt1v;
        wt1v = ntiles < wt0v + 64 ? ntiles : wt0v + 64;
    }
#undef WT0
#undef WT1
    }
#undef PLSLAM_MF_EPI_ROW
}

int launch_scan_sym_mfma(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero,
                         int nzero, bool multi_window, bool directed, hipStream_t s)
{
    if (nblocks <= 0) return PLSLAM_OK;
#define PLSLAM_MF_LAUNCH(M, D) \
    hipLaunchKernelGGL((k_scan_sym_mfma<M, D>), dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks, d_zero, nzero)
    if (multi_window) { if (directed) PLSLAM_MF_LAUNCH(true, true); else PLSLAM_MF_LAUNCH(true, false); }
    else              { if (directed) PLSLAM_MF_LAUNCH(false, true); else PLSLAM_MF_LAUNCH(false, false); }
#undef PLSLAM_MF_LAUNCH
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

}  // namespace plslam
