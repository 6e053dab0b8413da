// hamming_mfma_d.hip -- K1g: the DIRECTED matrix-core Hamming kNN-2 scan (gfx950).
//
// One item = one directed scan of cv::BFMatcher(NORM_HAMMING).knnMatch(k = 2): keys[i] = the two smallest
// (distance << 23 | j) over the rows j of b, for every row i of a (reference call sites src/mapHandler.cpp:277,424,597,
// 712,3223,3249: StVO::match runs it once per direction).  A mutual problem is TWO items, (d1 -> d2) and (d2 -> d1) -- what
// the reference's two knnMatch calls evaluate -- instead of K1f's one symmetric pass that serves both directions from one
// distance.  Why that is faster although every distance is now executed twice: the symmetric pass pays for its column
// direction with 48 + ~20 packed VALU instructions per 64 x 32 wave tile (exact best-2 per column and tile), 16 packs,
// a 1.2 GB partial table and a merge kernel, and sits on the VALU issue limit with the matrix pipe 30 % busy (round 2:
// 153 VALU per 8 MFMA).  The row direction alone needs ONE min per distance, taken straight from the accumulator's low
// half (v_min_u16, the one comparison in the fast VALU class) -- no pack, no per-tile column work, no partials, no merge:
// ~55 VALU per 8 MFMA, so the scan is paced by the matrix pipe.
//
// Distances: as K1e/K1f -- four v_mfma_scale_f32_32x32x64_f8f6f4 per 32 x 32 tile over fp4 (e2m1) codes of +-1, accumulator
// start 2^23 + 16384 + tag, result 2^23 + 128 d + tag: every partial sum an integer below 2^24, so the fp32 accumulation
// is exact and the float's low 16 bits ARE the key (d << 7 | tag).
//
// Row direction = K1f's "minimum now, second best later": per (row, column class = lane) the running minimum over a GROUP
// of 16 tiles; group minima are pushed into a sorted pair (parked in LDS) once per group; after the scan one lane per row
// merges its 32 classes: B0 = best key, B1 = best key outside B0's (group, class); the true second best is min(B1, best
// of the other 15 members of B0's (group, class)), recomputed with XOR + popcount from the raw rows.
//
// What is new in the layout: WHICH b row sits in (tile, lane) is free (only widen() must know), so inside a full group of
// 512 b rows the mapping is CLASS-MAJOR: j = 512 G + 16 class + tile-in-group.  The 16 members of a (group, class) are then
// 16 CONSECUTIVE rows = 512 contiguous bytes = 4 cache lines for the second-best recomputation, where K1f's j0 +- 32 k
// touched 16 lines per row (2 KB of L2 -> L1 traffic per row of a: 10 x the b stream itself).  The tile load gathers 32
// bytes from each of 32 lines; the other three quarters of a line are the next three tiles' (L1 hits).  The ragged rest of
// b (n2 mod 512 rows) is one more group in tile-major order (j = base + 32 tile + class), so that only the LAST tile of a
// scan has lanes without a column (one masked instantiation outside the steady loop).
//
// The tag of a key is the tile number within the group (4 bits, from the accumulator seed: 16 v_add per tile in the
// shadow of the MFMAs); the group number within the window (3 bits) is added at push time: windows of 128 tiles = 4096
// columns (K1f: 64 tiles).
#include "common.hpp"

#include <type_traits>

// build-time experiments (tools/build_exp.py; results are WRONG with any of them on), a bit mask:
//   1 no workgroup barrier   2 no row minima   4 no MFMA   16 no second-best fix-up   64 no group push
//   128 no expansion of the b tile (no global load, no LDS write)   256 no operand reads from LDS   1024 no finish_rows
#ifndef PLSLAM_MD_EXPERIMENT
#define PLSLAM_MD_EXPERIMENT 0
#endif
#define PLSLAM_MD_X(bit) ((PLSLAM_MD_EXPERIMENT & (bit)) != 0)

namespace plslam {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4), aligned(4)));   // descriptor rows are only 4-byte aligned
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
#define PLSLAM_GLOBAL __attribute__((address_space(1)))
typedef const PLSLAM_GLOBAL uint32_t* gcu32_t;
typedef const PLSLAM_GLOBAL u32x4_t* gcu32x4_t;
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef PLSLAM_GLOBAL u32x2_t* gu2_t;

namespace {

constexpr int MD_TILE_N = 32;                 // b rows per tile
constexpr int MD_KSTEPS = 4;                  // 256 bits = 4 x K 64
constexpr int MD_ROW_STRIDE = 144;            // bytes per expanded b row in LDS (128 + 16: 4-bank skew)
constexpr int MD_TILE_BYTES = MD_TILE_N * MD_ROW_STRIDE;
constexpr int MD_GROUP = 16;                  // tiles per group (512 b rows)
constexpr int MD_GROUP_ROWS = MD_GROUP * MD_TILE_N;
constexpr int MD_WINDOW = 128;                // tiles per window: tag = group in window (3 bits) << 4 | tile in group (4 bits)
constexpr uint32_t FP4_NEG = 0x88888888u;
constexpr uint32_t FP4_ONE = 0x22222222u;
constexpr int SCALE_A = 133, SCALE_B = 127;   // E8M0: 2^6 on the a side, 2^0 on the b side
constexpr uint32_t ACC_BITS = 0x4B000000u + 16384u;   // float bits of 2^23 + 16384
constexpr uint32_t KEY16_MAX = 0x807Fu;       // 16-bit keys are (d << 7) | tag7; anything above is "none"

__device__ __forceinline__ uint32_t umin_(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax_(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ void merge2(uint32_t& a0, uint32_t& a1, uint32_t c0, uint32_t c1)
{
    const uint32_t lo = umin_(a0, c0);
    const uint32_t hi = umin_(umax_(a0, c0), umin_(a1, c1));
    a0 = lo;
    a1 = hi;
}
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
// the running minimum of a (row, class): low halves only (v_min_u16: the one comparison of the fast VALU class).  Inline
// asm, because every C++ spelling ends up in the SLP vectoriser (two scalar minima -> v_perm + v_pk_min_u16 + two unpacks).
// The accumulator operand is an MFMA destination, and the wait states between an MFMA and a VALU read of its result are
// software's job, which the compiler does not do for asm operands ("K1e determinism"): every use
// below sits at least two MFMAs and a dozen instructions behind the MFMA that produced its operand, and
// tools/check_mfma_hazards.py proves that on the final listing in the CPU test suite (tests/test_mfma_encoding_cpu.py).
__device__ __forceinline__ uint32_t min_lo16(uint32_t running, uint32_t acc_bits)
{
    uint32_t r;
    asm("v_min_u16 %0, %1, %2" : "=v"(r) : "v"(running), "v"(acc_bits));
    return r;
}
// 32 bits of a descriptor -> 32 fp4 codes of s(bit): dword s holds bits 4k + s, nibble k = 0x2 | bit << 3 (see K1f)
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

// the b row that sits in (tile t, class cls) of a scan over n2 rows: class-major inside full groups of 512 rows,
// tile-major in the ragged rest (see the file header); host-visible so that tests can state the layout
__host__ __device__ inline int md_row_of(int t, int cls, int n2)
{
    const int gbase = (t >> 4) * MD_GROUP_ROWS;
    return gbase + MD_GROUP_ROWS <= n2 ? gbase + MD_GROUP * cls + (t & 15) : gbase + MD_TILE_N * (t & 15) + cls;
}
__host__ __device__ inline int md_ntiles(int n2)
{
    const int full = n2 / MD_GROUP_ROWS, rem = n2 - full * MD_GROUP_ROWS;
    return full * MD_GROUP + (rem + MD_TILE_N - 1) / MD_TILE_N;
}

__global__ void __launch_bounds__(256, 3)      // 3 waves per SIMD: <= 168 unified VGPRs
k_scan_dir_mfma(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks, int32_t* __restrict__ zero, int nzero)
{
    // one buffer, two lives: during the scan the double-buffered b tile (9 216 B) followed by the PARKED sorted pairs
    // ([wave][reg][lane] x 8 B = 32 768 B); after the scan the row-result transpose [wave][row 0..63][33] (33 792 B)
    constexpr int ROWX_STRIDE = 33;               // dwords per row: lane = row reads are conflict-free
    constexpr int PARK_OFF = 2 * MD_TILE_BYTES;
    constexpr int SMEM_BYTES = PARK_OFF + 4 * 16 * 64 * 8;
    static_assert(SMEM_BYTES >= 4 * 64 * ROWX_STRIDE * 4, "the transpose must fit");
    __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM_BYTES];
    uint8_t* const btile = smem;
    u32x2_t* const park = reinterpret_cast<u32x2_t*>(smem + PARK_OFF) + (threadIdx.x >> 6) * (16 * 64) + (threadIdx.x & 63);

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
    const gcu32_t araw = (gcu32_t) reinterpret_cast<const uint32_t*>(sd.a);
    const gcu32_t braw = (gcu32_t) reinterpret_cast<const uint32_t*>(sd.b);
    const int iw = bd.row0 + 64 * w;               // first of this wave's 64 rows of a

    // ---- A operands: rows iw + 32 mt + c, raw dword 2 ks + g of each, as fp4 codes of -s(a); the factor 64 is the block
    // scale.  Rows past n1 are clamped duplicates (their results are dropped in finish_rows) ----
    i32x4 afrag[2][MD_KSTEPS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = iw + 32 * mt + c;
        const int rrow = row < n1 ? row : n1 - 1;
        const gcu32_t p = araw + (size_t)rrow * 8 + g;
#pragma unroll
        for (int ks = 0; ks < MD_KSTEPS; ++ks) afrag[mt][ks] = expand_dword_fp4<true>(p[2 * ks]);
    }
    const int scale_a = SCALE_A, scale_b = SCALE_B;

    // row-direction state: gm[mt * 16 + r] (low half) = running minimum of the 16-bit keys (d << 7 | tile in group) of the
    // current group for row 32 mt + LOC(r) + 4 g of the wave, column class c
    uint32_t gm[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) gm[r] = 0xFFFFu;

    const int ntiles = md_ntiles(n2);
    const int nfull = (n2 / MD_GROUP_ROWS) * MD_GROUP;             // tiles of full (class-major) groups
    // expansion duty of this lane: b row of class (tid >> 3) of the tile, dword (tid & 7) of it
    const int ej = tid >> 3, ewd = tid & 7, ej16 = MD_GROUP * ej;
    auto load_raw = [&](int t) __attribute__((always_inline)) -> uint32_t {
        // md_row_of(t, ej, n2) with the two forms spelled out (t is wave-uniform)
        const int gbase = (t >> 4) * MD_GROUP_ROWS, tt = t & 15;
        const bool full = t < nfull;                                // wave-uniform: scalar selects, no branch
        int j = gbase + (full ? tt : MD_TILE_N * tt) + (full ? ej16 : ej);
        j = j < n2 ? j : n2 - 1;
        return braw[(uint32_t)(j * 8 + ewd)];                       // 32-bit offset from a scalar base (descriptor sets < 4 GB)
    };
    auto expand_store = [&](uint32_t raw, int buf) __attribute__((always_inline)) {
        uint8_t* dst = btile + buf * MD_TILE_BYTES + ej * MD_ROW_STRIDE + ewd * 16;
        *reinterpret_cast<i32x4*>(dst) = expand_dword_fp4<false>(raw);
    };

    int wt0 = 0, wt1 = ntiles < MD_WINDOW ? ntiles : MD_WINDOW;      // the current window of tiles
    uint32_t raw1 = 0u, raw2 = 0u, raw3 = 0u;                        // raw b dwords of tiles t+1, t+2, t+3 of the coming step

    // a group is over: its minima (two rows per packed key: M-tile 0 low, M-tile 1 high) get the group number and go into the
    // parked sorted pairs; the minima restart
    auto push_groups = [&](int t) __attribute__((always_inline)) {
        if (PLSLAM_MD_X(64)) return;
        const uint32_t gtag = (uint32_t)(((t - wt0) >> 4) << 4) * 0x00010001u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const u32x2_t v = park[r * 64];
            uint32_t b0 = v.x, b1 = v.y;
            const uint32_t key = pk_add16_sat(__builtin_amdgcn_perm(gm[16 + r], gm[r], 0x05040100u), gtag);
            pk_push2(b0, b1, key);
            park[r * 64] = u32x2_t{b0, b1};
            gm[r] = 0xFFFFu;
            gm[16 + r] = 0xFFFFu;
        }
    };

    // One tile: barrier | operand reads | expansion of tile t+1 | prefetch of tile t+4 | M-tile 0: 4 MFMAs, the minima of
    // M-tile 1 of tile t-1 between them | M-tile 1: 4 MFMAs, the minima of M-tile 0 of THIS tile between the later ones
    // (its chain is complete two MFMAs earlier).  One accumulator set, matrix and VALU work in flight together.
    f32x16 m0, m1;
    // the accumulator start value, in VECTOR registers on purpose (asm volatile: opaque to the compiler, which otherwise keeps
    // the wave-uniform value in 16 SGPRs and copies it into both accumulators every tile)
    u32x16 seed;
#pragma unroll
    for (int r = 0; r < 16; ++r) seed[r] = ACC_BITS;
    asm volatile("" : "+v"(seed));
    auto tile_step = [&](int t, bool with_prev, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;      // the last tile of a scan: lanes without a column
        if (!PLSLAM_MD_X(1)) __syncthreads();  // tile t expanded; every wave is past its reads of the other buffer
        const uint8_t* bt = btile + (t & 1) * MD_TILE_BYTES + c * MD_ROW_STRIDE + 16 * g;
        i32x4 bfr[MD_KSTEPS];
#pragma unroll
        for (int ks = 0; ks < MD_KSTEPS; ++ks)
            bfr[ks] = PLSLAM_MD_X(256) ? i32x4{(int)FP4_ONE + t, (int)FP4_ONE, (int)FP4_ONE + ks, (int)FP4_ONE}
                                       : *reinterpret_cast<const i32x4*>(bt + 32 * ks);
        if (!PLSLAM_MD_X(128)) expand_store(raw1, (t + 1) & 1);   // past the last tile: a harmless rewrite of the idle buffer
        raw1 = raw2;
        raw2 = raw3;
        if (!PLSLAM_MD_X(128)) raw3 = load_raw(t + 4);
        const f32x16 cseed = __builtin_bit_cast(f32x16, seed);
        // lanes whose column does not exist (last tile only): their keys become "none"
        const uint32_t colmask = MASKED ? (md_row_of(t, c, n2) < n2 ? 0u : 0xFFFFu) : 0u;
#define PLSLAM_MD_MMA(ACC, MT, KS, CIN)                                                            \
        {                                                                                          \
            const i32x8 a8 = {afrag[MT][KS].x, afrag[MT][KS].y, afrag[MT][KS].z, afrag[MT][KS].w, 0, 0, 0, 0}; \
            const i32x8 b8 = {bfr[KS].x, bfr[KS].y, bfr[KS].z, bfr[KS].w, 0, 0, 0, 0};             \
            if (!PLSLAM_MD_X(4))                                                                   \
                ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, CIN, 4, 4, 0, scale_a, 0, scale_b); \
            else { const f32x16 cin_ = CIN; ACC = cin_; ACC[KS] = __builtin_bit_cast(float, bfr[KS].x ^ a8[0]); } \
            asm volatile("" : "+v"(ACC));    /* pins the MFMA here (no instruction) */              \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
#define PLSLAM_MD_MIN4(ACC, BASE, R0, ON)                                                          \
        {                                                                                          \
            if ((ON) && !PLSLAM_MD_X(2)) {                                                         \
                const u32x16 ab_ = __builtin_bit_cast(u32x16, ACC);                                \
                gm[(BASE) + (R0)]     = min_lo16(gm[(BASE) + (R0)],     ab_[(R0)]     | colmask_prev_or_cur); \
                gm[(BASE) + (R0) + 1] = min_lo16(gm[(BASE) + (R0) + 1], ab_[(R0) + 1] | colmask_prev_or_cur); \
                gm[(BASE) + (R0) + 2] = min_lo16(gm[(BASE) + (R0) + 2], ab_[(R0) + 2] | colmask_prev_or_cur); \
                gm[(BASE) + (R0) + 3] = min_lo16(gm[(BASE) + (R0) + 3], ab_[(R0) + 3] | colmask_prev_or_cur); \
            }                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            const uint32_t colmask_prev_or_cur = 0u;               // tile t-1 is never the masked one
            PLSLAM_MD_MMA(m0, 0, 0, cseed)  PLSLAM_MD_MIN4(m1, 16, 0, with_prev)
            PLSLAM_MD_MMA(m0, 0, 1, m0)     PLSLAM_MD_MIN4(m1, 16, 4, with_prev)
            PLSLAM_MD_MMA(m0, 0, 2, m0)     PLSLAM_MD_MIN4(m1, 16, 8, with_prev)
            PLSLAM_MD_MMA(m0, 0, 3, m0)     PLSLAM_MD_MIN4(m1, 16, 12, with_prev)
        }
        // tile t-1 closed a group: push before this tile's minima start on the restarted registers (wave-uniform branch)
        if (with_prev && ((t - 1 - wt0) & (MD_GROUP - 1)) == MD_GROUP - 1) push_groups(t - 1);
        {
            const uint32_t colmask_prev_or_cur = colmask;
            PLSLAM_MD_MMA(m1, 1, 0, cseed)
            PLSLAM_MD_MMA(m1, 1, 1, m1)
            // the tile number within the group rides in the accumulator seed: + 1 per tile, back to 0 at a group's start
            // (cseed was read by the two chain heads above; the compiler keeps the wait states of that hazard)
            {
#pragma unroll
                for (int r = 0; r < 16; ++r) seed[r] += 1u;            // inline constant: the fast VALU class
                if (((t + 1 - wt0) & (MD_GROUP - 1)) == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) seed[r] -= (uint32_t)MD_GROUP;
                }
                asm volatile("" : "+v"(seed));
            }
            __builtin_amdgcn_sched_barrier(0);
            PLSLAM_MD_MMA(m1, 1, 2, m1)     PLSLAM_MD_MIN4(m0, 0, 0, true)  PLSLAM_MD_MIN4(m0, 0, 4, true)
            PLSLAM_MD_MMA(m1, 1, 3, m1)     PLSLAM_MD_MIN4(m0, 0, 8, true)  PLSLAM_MD_MIN4(m0, 0, 12, true)
        }
    };
    // the minima of M-tile 1 of the window's last tile (no following step to hide under), and the last group
    auto drain = [&](int t, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const uint32_t colmask_prev_or_cur = MASKED ? (md_row_of(t, c, n2) < n2 ? 0u : 0xFFFFu) : 0u;
        PLSLAM_MD_MIN4(m1, 16, 0, true) PLSLAM_MD_MIN4(m1, 16, 4, true) PLSLAM_MD_MIN4(m1, 16, 8, true) PLSLAM_MD_MIN4(m1, 16, 12, true)
        push_groups(t);
    };
#undef PLSLAM_MD_MIN4
#undef PLSLAM_MD_MMA

    // Row results of a window.  Every lane holds, per packed register, the best two GROUP minima (16-bit keys (d, tile in
    // window)) of ITS column class for two rows.  Transpose through LDS so that one lane owns one row: lane l reads the 32
    // class entries of row l in class order, widens them to (key16 << 16 | class) -- which orders like (d, j): inside a
    // full group j = 512 G + 16 class + tile, in the ragged group ties between classes of DIFFERENT tiles would order by
    // class first, so there the composite is (d, tile, class) = key16 << 16 | class as well (j = base + 32 tile + class)
    // but across a full and the ragged group... both composites compare (d, group) first, and within one group either
    // (class, tile) [full] or (tile, class) [ragged] IS the j order, so the low bits are arranged per group kind below.
    auto finish_rows = [&]() __attribute__((always_inline)) {
        uint32_t* rowx = reinterpret_cast<uint32_t*>(smem) + w * (64 * ROWX_STRIDE);
        u32x2_t rb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rb[r] = park[r * 64];
        __syncthreads();                           // every wave holds its pairs: the transpose may overwrite the parking area
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = (r & 3) + 8 * (r >> 2) + 4 * g;
            rowx[lrow * ROWX_STRIDE + c] = (rb[r].x & 0xFFFFu) | (rb[r].y << 16);
            rowx[(32 + lrow) * ROWX_STRIDE + c] = (rb[r].x >> 16) | (rb[r].y & 0xFFFF0000u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // order key of an entry = (d : 9, group : 3, x : 9) with x = class << 4 | tile in a full group and tile << 5 | class
        // in the ragged one: within a group that is the j order, across groups (d, group) decides.  21 bits + "none" on top
        const int rag_g = (nfull - wt0) >> 4;      // the ragged group's number within this window (>= 8: not in this window)
        auto okey = [&](uint32_t k16, uint32_t cls) -> uint32_t {
            const uint32_t d = k16 >> 7, gi = (k16 >> 4) & 7u, tt = k16 & 15u;
            const uint32_t x = (int)gi == rag_g ? (tt << 5 | cls) : (cls << 4 | tt);
            return k16 > KEY16_MAX ? 0xFFFFFFFFu : (d << 12 | gi << 9 | x);
        };
        uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
        const uint32_t* mine = rowx + lane * ROWX_STRIDE;
#pragma unroll 8
        for (int cls = 0; cls < 32; ++cls) {
            const uint32_t e = mine[cls];
            merge2(k0, k1, okey(e & 0xFFFFu, (uint32_t)cls), okey(e >> 16, (uint32_t)cls));
        }
        // order key -> (d << 23 | j)
        auto tile_cls = [&](uint32_t k, uint32_t& t, uint32_t& cls) {
            const uint32_t gi = (k >> 9) & 7u, x = k & 511u;
            const bool rag = (int)gi == rag_g;
            const uint32_t tt = rag ? x >> 5 : x & 15u;
            cls = rag ? x & 31u : x >> 4;
            t = (uint32_t)wt0 + (gi << 4) + tt;
        };
        auto widen = [&](uint32_t k) -> uint32_t {
            if (k == 0xFFFFFFFFu) return KEY_NONE;
            uint32_t t, cls;
            tile_cls(k, t, cls);
            return ((k >> 12) << KEY_IDX_BITS) | (uint32_t)md_row_of((int)t, (int)cls, n2);
        };
        const int row = iw + lane;
        if (row < n1) {
            const gu2_t out = (gu2_t) reinterpret_cast<u32x2_t*>(sd.keys12) + row;
            uint32_t r0 = widen(k0), r1 = widen(k1);
            if (k0 != 0xFFFFFFFFu && !PLSLAM_MD_X(16)) {
                // the other members of the winner's (group, class): consecutive rows in a full group, 32 apart in the ragged one
                uint32_t t0, cls0;
                tile_cls(k0, t0, cls0);
                const uint32_t tg = t0 & ~(uint32_t)(MD_GROUP - 1);
                const bool rag = tg >= (uint32_t)nfull;
                const uint32_t jbase = (tg >> 4) * MD_GROUP_ROWS + (rag ? cls0 : MD_GROUP * cls0);
                const uint32_t jstep = rag ? MD_TILE_N : 1u;
                const gcu32x4_t ap = (gcu32x4_t)(araw + (size_t)row * 8);
                const u32x4_t a_lo = ap[0], a_hi = ap[1];
#pragma unroll
                for (int k = 0; k < MD_GROUP; ++k) {
                    const uint32_t j = jbase + (uint32_t)k * jstep;
                    const bool ok = (uint32_t)k != (t0 & (uint32_t)(MD_GROUP - 1)) && j < (uint32_t)n2;
                    const gcu32x4_t bp = (gcu32x4_t)(braw + (size_t)(ok ? j : jbase) * 8);
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
            if (wt0 > 0) {                                  // later windows: merge with the windows before
                const u32x2_t prev = *out;
                merge2(r0, r1, prev.x, prev.y);
            }
            *out = u32x2_t{r0, r1};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    for (;;) {
#pragma unroll
        for (int r = 0; r < 16; ++r) park[r * 64] = u32x2_t{0xFFFFFFFFu, 0xFFFFFFFFu};   // wave-private: no barrier needed
        expand_store(load_raw(wt0), 0);            // wt0 is a multiple of 128: buffer parity restarts at 0
        raw1 = load_raw(wt0 + 1);
        raw2 = load_raw(wt0 + 2);
        raw3 = load_raw(wt0 + 3);
        // only the last tile of the scan can lack columns
        const bool last_partial = wt1 == ntiles && (n2 % MD_TILE_N) != 0;
        const int tsteady = last_partial ? wt1 - 1 : wt1;
        if (wt0 < tsteady) {
            tile_step(wt0, false, std::false_type{});
            for (int t = wt0 + 1; t < tsteady; ++t) tile_step(t, true, std::false_type{});
        }
        if (last_partial) {
            tile_step(wt1 - 1, wt0 < tsteady, std::true_type{});
            drain(wt1 - 1, std::true_type{});
        } else {
            drain(wt1 - 1, std::false_type{});
        }
        __syncthreads();                           // every wave is past its last operand read of the b tile
        if (!PLSLAM_MD_X(1024)) finish_rows();
        if (wt1 == ntiles) break;
        __syncthreads();                           // smem becomes the b tile (+ parking area) again
        wt0 = wt1;
        wt1 = ntiles < wt0 + MD_WINDOW ? ntiles : wt0 + MD_WINDOW;
#pragma unroll
        for (int r = 0; r < 16; ++r) seed[r] = ACC_BITS;
        asm volatile("" : "+v"(seed));
    }
}

int launch_scan_dir_mfma(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero, int nzero, hipStream_t s)
{
    if (nblocks <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_scan_dir_mfma, dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks, d_zero, nzero);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

}  // namespace plslam
