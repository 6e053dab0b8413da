// hamming_mfma_h.hip -- K1h: the matrix-core symmetric Hamming kNN-2 scan with MINIMUM-ONLY bookkeeping in BOTH
// directions (gfx950).  Round 3; the default.
//
// Contract, work decomposition, block tables: those of K1f (hamming_mfma_g.hip) -- keys12[i] = best-2 over j; the column
// partials are per WORKGROUP (256 rows of a; K1f: per wave): part21[256-row block of a][column slot] = one word
// (d0 << 17 | row0 << 9 | d1): the block's best row and the distance of its best row outside that row's group of 16 (two words
// (d << 8 | row) when exact key tables are asked for); keys = (distance << 23) | index =
// cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) order in both directions (reference call sites src/mapHandler.cpp:277,424,597,
// 712,3223,3249).  The distances come out of the same four v_mfma_scale_f32_32x32x64_f8f6f4 per 32 x 32 tile (fp4 codes of
// +-1, accumulator = 2^23 + 128 d + tag, exact).
//
// K1f was bound by VALU issue: 153 VALU instructions per 8 MFMAs (round 2 PMC), of which 48 were the exact best-2 push of
// the column direction, 16 the per-tile tile-number tags in the accumulator seeds, ~20 the per-tile column finish.  Here:
//
//  1. Column direction = "minimum now, second best later" too.  A lane's 16 accumulator registers of an M-tile are a GROUP of
//     16 rows of a; per tile one packed min chain over the registers gives the group minima (16 v_pk_min_u16 instead of 48),
//     the workgroup's 16 groups per column (4 waves x 2 M-tiles x 2 lane halves, combined through LDS once per 4 tiles) give
//     B0 = the best and B1 = the best OUTSIDE B0's group, and that pair is the column partial (a quarter of K1f's partial
//     table: the table is written once and read once, 1.24 GB each way at C2 / 4096 pairs before).  The merge kernel
//     (k_merge_fix16) combines the 256-row blocks the same way -- K0 = best
//     key, K1 = best key outside K0's 16-row group -- and then recomputes the 15 other members of K0's group with XOR +
//     popcount from the raw rows: second best = min(K1, best of those).  Exact, tie order included.
//  2. WHICH row of a sits in (M-tile, MFMA row) is free, so a lane's 16 registers hold 16 CONSECUTIVE rows of a
//     (row within the workgroup's 256 = 128 mt + 32 wave + 16 g + r): the recomputation reads 512 contiguous bytes, and the
//     register number IS the row within the group -- the tag (32 wave + 16 g + r: the row within the M-tile's half of the
//     block) rides in the accumulator seed as a per-register CONSTANT (no per-tile seed updates).  M-tile 1 = the block's
//     upper 128 rows: the 16-bit keys of ONE half of a packed register order like (d, row) across all four waves, which is
//     what lets the workgroup's column minima be combined with packed instructions (combine_columns).
//  3. WHICH row of b sits in (tile, lane) is free too: inside a group of 16 tiles the mapping is class-major,
//     j = 512 G + S class + tile-in-group (S = 16; in the ragged last group S = ceil(rest / 32) and the group has S tiles), so
//     the row direction's second-best recomputation -- all members of the winner's (group, class) -- reads S consecutive
//     rows of b (K1f: 16 cache lines per row of a, 10 x the b stream itself).  No tile number in the row keys: the
//     recomputation finds the exact column.  Columns that do not exist (ragged group only) get zero codes (distance 128) and
//     a per-lane penalty in the seed: no masking instructions in any tile.
//
// The index of the SECOND best row key is exact only on request (SymDesc::flags bit 0: one more recomputation, used by
// plslam_knn2_hamming256 and the key-dump tests): StVO::match needs the second best's DISTANCE only (ratio test).
#include "mfma_h_common.hpp"

#include <type_traits>

// build-time experiments (tools/build_exp.py; results are WRONG with any of them on), a bit mask:
//   1 no workgroup barrier   2 no bookkeeping rows   4 no MFMA   8 no column store   16 no second-best fix-up
//   32 no finish_columns     64 no group push        128 no expansion of the b tile   256 no operand reads from LDS
//   512 no pack              1024 no finish_rows
#ifndef PLSLAM_MH_EXPERIMENT
#define PLSLAM_MH_EXPERIMENT 0
#endif
#define PLSLAM_MH_X(bit) ((PLSLAM_MH_EXPERIMENT & (bit)) != 0)

namespace plslam {

// (typedefs, constants and device helpers: mfma_h_common.hpp)

int mh_slot_of_column(int n2, int j) { return MhLayout(n2).slot_of(j); }     // for tests / tools (plslam_match_plan_dump readers)
// row within the workgroup's 256 rows of a held by wave w, M-tile mt, MFMA row m: 128 mt + 32 w + 16 g + r with
// m = (r & 3) + 8 (r >> 2) + 4 g
__host__ __device__ inline int mh_block_row(int w, int mt, int m) { return 128 * mt + 32 * w + 16 * ((m >> 2) & 1) + (m & 3) + 4 * (m >> 3); }

#if PLSLAM_BUILD_LEGACY_SCANS      // K1h's scan kernel (mfma_form 4): round 3's default, kept as a cross-check; its merge kernel below serves K1i
// DIRECTED = true: only keys12 (row direction) is produced.
template <bool DIRECTED>
__global__ void __launch_bounds__(256, 3)      // 3 waves per SIMD: <= 168 unified VGPRs
k_scan_sym_mfma_h(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks, int32_t* __restrict__ zero, int nzero)
{
    // one buffer, two lives: during the scan the double-buffered b tile (9 216 B) followed by the PARKED sorted pairs of the
    // row direction ([wave][reg][lane] x 8 B = 32 768 B); after the scan the row-result transpose [wave][row 0..63][33]
    constexpr int ROWX_STRIDE = 33;               // dwords per row: lane = row reads are conflict-free
    constexpr int PARK_OFF = 2 * MH_TILE_BYTES;
    constexpr int SMEM_BYTES = PARK_OFF + 4 * 16 * 64 * 8;
    static_assert(SMEM_BYTES >= 4 * 64 * ROWX_STRIDE * 4, "the transpose must fit");
    __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM_BYTES];
    // column minima of the last 8 tiles, [tile & 7][wave][lane] (1 KB per tile): every lane parks its two group minima per
    // tile (one LDS store, no cross-lane step on the tile's path).  Once per 8 tiles the workgroup combines them, a lane
    // per column: the 4 waves' 2 x 2 group minima of it become ONE word for the workgroup's 256 rows (combine_columns)
    __shared__ __attribute__((aligned(16))) uint32_t colstage[MH_CGROUP * 256];
    // raw b dwords in flight: tile t's 256 dwords (one per lane: its expansion duty) in rawring[t % 3], written by LDS-DMA
    // (global_load_lds_dword: no VGPR destination) and read back by the lane that asked for them
    __shared__ __attribute__((aligned(16))) uint32_t rawring[3][256];      // (3 slots: with 4 the workgroup's LDS passes 160 KB / 3)
    uint8_t* const btile = smem;
    u32x2_t* const park = reinterpret_cast<u32x2_t*>(smem + PARK_OFF) + (threadIdx.x >> 6) * (16 * 64) + (threadIdx.x & 63);

    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nzero; i += 256) zero[i] = 0;

    const int wg = xcd_remap_(blockIdx.x, gridDim.x);
    const BlockDesc bd = blocks[wg];
    if (bd.item < 0) return;                       // padding entry of the XCD-striped table
    const SymDesc sd = syms[bd.item];
    const int n1 = sd.n1, n2 = sd.n2;
    const MhLayout L(n2);
    const int ntiles = L.ntiles, nfull = L.nfull, rag_s = L.rag_s;
    const int n2p = (MH_TILE_N * ntiles + 255) & ~255;            // slots per row of the partial table (= n2 rounded up to 256)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, g = lane >> 5;
    const gcu32_t araw = (gcu32_t) reinterpret_cast<const uint32_t*>(sd.a);
    const int iw = bd.row0 + 32 * w;               // first of this wave's 32 rows of M-tile 0; M-tile 1: + 128

    // ---- A operands: MFMA row c of M-tile mt = block row mh_block_row(w, mt, c); raw dword 2 ks + g of each, as fp4 codes of
    // -s(a); the factor 64 is the block scale.  Rows past n1 are clamped duplicates (masked / dropped below) ----
    i32x4 afrag[2][MH_KSTEPS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = bd.row0 + mh_block_row(w, mt, c);
        const int rrow = row < n1 ? row : n1 - 1;
        const gcu32_t p = araw + (size_t)rrow * 8 + g;
#pragma unroll
        for (int ks = 0; ks < MH_KSTEPS; ++ks) afrag[mt][ks] = expand_dword_fp4<true>(p[2 * ks]);
    }
    const int scale_a = SCALE_A, scale_b = SCALE_B;

    // row-direction state per accumulator register r (M-tile 0 in the low halves, M-tile 1 in the high halves):
    //   gm[r]    running minimum of the 16-bit keys (d << 7 | 32 w + 16 g + r) of the current group of 16 tiles, column class c
    //   park[r]  (LDS) the best two GROUP minima (d << 7 | group in window << 5 | 16 g + r) of the lane's column class
    uint32_t gm[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) gm[r] = 0xFFFFFFFFu;

    // accumulator start: 2^23 + 16384 + 32 w + 16 g + r (the row within its M-tile's half of the block: constant per register and lane) [+ the
    // penalty of a lane whose column does not exist].  In VECTOR registers on purpose (asm volatile: opaque to the compiler, which would keep
    // wave-uniform values in SGPRs and copy them into both accumulators every tile)
    u32x16 seed;
#pragma unroll
    for (int r = 0; r < 16; ++r) seed[r] = ACC_BITS + (uint32_t)(32 * w + 16 * g + r);   // tag = the row within its M-tile's 128
    asm volatile("" : "+v"(seed));
    // ragged group: this lane's class has `lim` tiles with a column (0 .. rag_s); from tile-in-group == lim on it has none
    const int rest = n2 - (nfull >> 4) * MH_GROUP_ROWS;
    // (recomputed where it is needed -- twice per scan -- instead of living in a register through the tile loop)
    auto lane_lim = [&]() __attribute__((always_inline)) -> int {
        const int cc = (int)(threadIdx.x & 31u), v = rest - rag_s * cc;
        return rag_s == 0 ? 0 : (v < 0 ? 0 : (v > rag_s ? rag_s : v));
    };
    const int lim_part = rag_s ? rest % rag_s : 0;                 // the one class that is cut (wave-uniform): its lim, 0 = none is

    const bool rows_ragged = iw + 128 + 32 > n1;   // wave-uniform: some of this wave's rows do not exist
    // rows of this lane's two groups that exist: register r of M-tile mt holds row iw + 128 mt + 16 g + r
#define PLSLAM_MH_NV_LO (n1 - iw - 16 * (int)((threadIdx.x >> 5) & 1u))
    // the workgroup's row of the partial table: n2p words (exact key tables: pairs of words)
    const bool wide_part = !DIRECTED && (sd.flags & 1);
    const gu32_t part = DIRECTED ? (gu32_t) nullptr : (gu32_t) sd.part21 + (size_t)(bd.row0 >> 8) * n2p * (wide_part ? 2 : 1);
    uint32_t* const cstage = colstage + 64 * w;        // + 256 (tile & 7) + lane
#pragma unroll
    for (int k = 0; k < 2; ++k) *reinterpret_cast<i32x4*>(colstage + 8 * tid + 4 * k) = i32x4{-1, -1, -1, -1};   // (two barriers before the first use)

    // expansion duty of this lane: the b row of class (tid >> 3) of the tile, dword (tid & 7) of it.  Byte offset of that
    // dword = [group, tile in group: scalar] + [class x stride: per lane, one value for full groups, one for the ragged one]
    const int ej = tid >> 3, ewd4 = (tid & 7) * 4;
    const PLSLAM_GLOBAL char* const bbytes = (const PLSLAM_GLOBAL char*) sd.b;
    auto load_raw = [&](int t) __attribute__((always_inline)) -> uint32_t {
        // everything wave-uniform is scalar: the group's base goes into the scalar address, the lane adds (class x stride +
        // tile in group) rows, clamped to the last row of b (rows past the end are duplicates that get zero codes)
        const int tc = t < ntiles ? t : ntiles - 1;
        const int gbase = (tc >> 4) * MH_GROUP_ROWS;
        const uint32_t s = tc < nfull ? (uint32_t)MH_GROUP : (uint32_t)rag_s;
        uint32_t row = __umul24((uint32_t)ej, s) + (uint32_t)(tc & 15);      // v_mad_u32_u24: full rate (v_mul_lo_u32 is not)
        const uint32_t last = (uint32_t)(n2 - 1 - gbase);
        row = row < last ? row : last;
        return *reinterpret_cast<gcu32_t>(bbytes + (size_t)gbase * 32 + (row * 32u + (uint32_t)ewd4));
    };
    // The same load for the steady loop as LDS-DMA, issued through inline asm, with the matching wait issued by hand.
    // Why by hand: the compiler's wait-count pass gives up at the loop's back edge and behind the loop's rare paths (stores
    // of column results, the ragged group's branches) and inserts s_waitcnt vmcnt(0) once per unrolled iteration -- a wait
    // for the load it has JUST issued, i.e. a memory latency every four tiles.  Why LDS-DMA: an asm load into a VGPR counts
    // as written when the statement ends, and the compiler is free to copy that register before the data lands (it did:
    // a first version with VGPR destinations produced different tables from run to run).  Loads return in order, a tile's
    // dword is used three steps after its request and two younger requests are then in flight: vmcnt(2) is exact, and
    // anything else the loop sends to memory in between (a store of column results at most) only makes it stricter.
    // Every lane reads back the dword its own request wrote: no barrier between the wait and the ds_read.
    auto load_raw_async = [&](int t, int slot) __attribute__((always_inline)) {
        const int tc = t < ntiles ? t : ntiles - 1;
        const int gbase = (tc >> 4) * MH_GROUP_ROWS;
        const uint32_t s = tc < nfull ? (uint32_t)MH_GROUP : (uint32_t)rag_s;
        uint32_t row = __umul24((uint32_t)ej, s) + (uint32_t)(tc & 15);
        const uint32_t last = (uint32_t)(n2 - 1 - gbase);
        row = row < last ? row : last;
        const uint32_t voff = row * 32u + (uint32_t)ewd4;
        const PLSLAM_GLOBAL char* sbase = bbytes + (size_t)gbase * 32;
        // M0 = the wave's 256 bytes of the slot (the hardware adds lane x 4); M0 is compiler-reserved: saved and restored here
        const uint32_t lds_dst = (uint32_t)(uintptr_t)(&rawring[slot][64 * w]);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
    };
    auto take_raw = [&](int slot) __attribute__((always_inline)) -> uint32_t {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        return rawring[slot][tid];
    };
    // tile `tn` (the next one) into buffer `buf`; in the ragged group the rows that do not exist get zero codes
    auto expand_store = [&](uint32_t raw, int buf, int tn) __attribute__((always_inline)) {
        uint8_t* dst = btile + buf * MH_TILE_BYTES + ej * MH_ROW_STRIDE + ewd4 * 4;
        i32x4 v = expand_dword_fp4<false>(raw);
        if (tn >= nfull) {                                          // wave-uniform
            const int vm = (int)(__umul24((uint32_t)rag_s, (uint32_t)ej) + (uint32_t)(tn & 15)) < rest ? -1 : 0;
            v &= i32x4{vm, vm, vm, vm};
        }
        *reinterpret_cast<i32x4*>(dst) = v;
    };

    int wt0 = 0, wt1 = ntiles < MH_WINDOW ? ntiles : MH_WINDOW;      // the current window of tiles
    int ring_slot = 1;                                               // rawring slot of the NEXT tile
    // (The tile loop is unrolled by four so that the ring slot and the b-tile buffer of a step are compile-time facts.  K1f
    // rotated a three-deep register ring with moves: the move reads the newest register, so every tile waited for the load it
    // had just issued -- s_waitcnt vmcnt(0), a full memory latency per tile.)

    // Block kb of 8 tiles (column slots 256 kb .. 256 kb + 255) is complete in colstage: every wave's stores of it are behind a
    // workgroup barrier.  A lane per column: the 16 group minima of the column -- 4 waves x 2 lane halves, M-tile 0 in the low
    // and M-tile 1 in the high halves of the words -- are 8 words whose halves order like (d, row) among themselves (the tag
    // is the row within the M-tile's 128), so the best two per half come out of a PACKED network (20 instructions for
    // both halves); widened to (d << 8 | row within the workgroup's 256), the best two of those four: K0 = the best row, K1
    // = the best row outside K0's group of 16 (every key IS a group minimum).  One word leaves per column: K0 << 9 | K1's
    // distance; with exact key tables asked for (SymDesc::flags) the pair (K0, K1).  The caller's barrier behind it lets the
    // slots be rewritten.
    // (the workgroup's row of the partial table as a SCALAR base: a pointer computed from readfirstlane values is a vector
    // value to the compiler -- and then a spilled one)
    const uint64_t part_u = (uint64_t)(uintptr_t)part;
    const uint64_t part_s = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(part_u >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)part_u);
    auto combine_columns = [&](int kb) __attribute__((always_inline)) {
        if (DIRECTED || PLSLAM_MH_X(8)) return;
        // (the lane number from mbcnt, not from threadIdx: two instructions here instead of a value kept -- and spilled --
        // across the tile loop, whose reload would wait on vmcnt(0), i.e. on the raw-row prefetch; asm volatile: the
        // builtin form is loop-invariant, gets hoisted -- and spilled all the same)
        uint32_t l_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l_));
        // column slot 64 w + l_ of the block: tile 2 w + (l_ >> 5), class l_ & 31
        const uint32_t* const src = colstage + 512 * w + (((l_ & 32u) << 3) | (l_ & 31u));
        uint32_t p[8];
#pragma unroll
        for (int v = 0; v < 4; ++v) { p[2 * v] = src[64 * v]; p[2 * v + 1] = src[64 * v + 32]; }
        // best two of 8, both halves at once: 4 sorted pairs, then 3 merges of sorted pairs
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { lo[q] = pk_min16(p[2 * q], p[2 * q + 1]); hi[q] = pk_max16(p[2 * q], p[2 * q + 1]); }
        auto pk_merge = [](uint32_t& a0, uint32_t& a1, uint32_t c0, uint32_t c1) {
            const uint32_t m = pk_max16(a0, c0);
            a0 = pk_min16(a0, c0);
            a1 = pk_min16(m, pk_min16(a1, c1));
        };
        pk_merge(lo[0], hi[0], lo[1], hi[1]);
        pk_merge(lo[2], hi[2], lo[3], hi[3]);
        pk_merge(lo[0], hi[0], lo[2], hi[2]);
        // (d << 7 | t) + (d << 7) [+ 128: M-tile 1] = (d << 8 | row in the block); "none" 0xFFFF -> 0x1FF7F / 0x1FFFF: d = 511
        const uint32_t e0 = lo[0] & 0xFFFFu, e1 = hi[0] & 0xFFFFu, u0 = lo[0] >> 16, u1 = hi[0] >> 16;
        uint32_t k0 = e0 + (e0 & 0xFF80u), k1 = e1 + (e1 & 0xFF80u);
        merge2(k0, k1, u0 + (u0 & 0xFF80u) + 128u, u1 + (u1 & 0xFF80u) + 128u);
        if (PLSLAM_MH_X(32)) k0 = p[0];
        const uint32_t slot4 = (uint32_t)(256 * kb + 64 * w) * 4u;                             // (scalar) n2p is a multiple of 256
        if (!wide_part) {
            const uint32_t e = (k0 << 9) | (k1 >> 8);
            PLSLAM_GLOBAL uint32_t* dst = (PLSLAM_GLOBAL uint32_t*)((PLSLAM_GLOBAL char*)(uintptr_t)(part_s + slot4) + 4u * l_);
            if (PLSLAM_NT_STREAMS) __builtin_nontemporal_store(e, dst);
            else *dst = e;
        } else {
            const u32x2_t e = {k0, k1};
            PLSLAM_GLOBAL u32x2_t* dst = (PLSLAM_GLOBAL u32x2_t*)((PLSLAM_GLOBAL char*)(uintptr_t)(part_s + 2u * slot4) + 8u * l_);
            if (PLSLAM_NT_STREAMS) __builtin_nontemporal_store(e, dst);
            else *dst = e;
        }
        // (slots of tiles of a last, partial block that never ran hold older values: never read, and a function of the
        // inputs like everything else in the table)
    };
    // a row group is over: its minima get the group number and go into the parked sorted pairs; the minima restart
    auto push_groups = [&](int t) __attribute__((always_inline)) {
        if (PLSLAM_MH_X(64)) return;
        // the tag's wave bits become the group number (an XOR: "none" stays above every key, whatever its low bits)
        const uint32_t gtag = (uint32_t)(((((t - wt0) >> 4) ^ w) & 3) << 5) * 0x00010001u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const u32x2_t v = park[r * 64];
            uint32_t b0 = v.x, b1 = v.y;
            pk_push2(b0, b1, gm[r] ^ gtag);
            park[r * 64] = u32x2_t{b0, b1};
            gm[r] = 0xFFFFFFFFu;
        }
    };
    // column minima of a finished tile: parked as they are (the seeds made the tags), combined in combine_columns
    auto finish_columns = [&](int t, uint32_t cm) __attribute__((always_inline)) {
        if (DIRECTED || PLSLAM_MH_X(32)) { asm volatile("" ::"v"(cm)); return; }
        cstage[(t & (MH_CGROUP - 1)) * 256 + lane] = cm;
    };

    // Bookkeeping of one packed key pair kc[R] (rows r of the lane's two groups, column class c): ONE packed min into the
    // row direction's group minimum, ONE into the column direction's group minimum.  MASKED: rows of a that do not exist
    // must not win a column.
    // (two column chains, even / odd registers: a packed op that reads the result of the packed op two slots earlier costs
    // an s_nop -- round 2 counted 19 of them per tile)
#define PLSLAM_MH_EPI_ROW(R)                                                                       \
    {                                                                                              \
        uint32_t kcv = kc[R];                                                                      \
        gm[R] = pk_min16(gm[R], kcv);                                                              \
        if (!DIRECTED) {                                                                           \
            if (MASKED) kcv |= ((R) < PLSLAM_MH_NV_LO ? 0u : 0x0000FFFFu) | ((R) < PLSLAM_MH_NV_LO - 128 ? 0u : 0xFFFF0000u); \
            if ((R) & 1) cm1 = pk_min16(cm1, kcv); else cm = pk_min16(cm, kcv);                    \
        }                                                                                          \
    }
    // Software pipeline, ONE accumulator set (as K1f):  M(t): 8 MFMAs -> P(t): 16 v_perm pack the 32 accumulators into 16
    // key pairs kc[] -> E(t): bookkeeping from kc[], issued BETWEEN the MFMAs of M(t+1).
    //   step(t) = barrier | operand reads | M(t) x E(t-1) | expand(t+1) | finish_columns(t-1) | P(t)
    uint32_t kc[16];
    auto tile_step = [&](int t, auto u_tag, bool with_prev, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr int U = decltype(u_tag)::value;                      // t & 3
        if (!PLSLAM_MH_X(1)) __syncthreads();  // tile t expanded; every wave is past its reads of the other buffer
        const uint8_t* bt = btile + (U & 1) * MH_TILE_BYTES + c * MH_ROW_STRIDE + 16 * g;
        uint32_t cm = 0xFFFFFFFFu, cm1 = 0xFFFFFFFFu;
        // operand reads: K-steps 0 and 1 now, 2 and 3 behind the MFMAs of steps 0 and 1 (their registers): 8 live registers, not 16
        auto read_b = [&](int ks) __attribute__((always_inline)) -> i32x4 {
            return PLSLAM_MH_X(256) ? i32x4{(int)FP4_ONE + t, (int)FP4_ONE, (int)FP4_ONE + ks, (int)FP4_ONE}
                                    : *reinterpret_cast<const i32x4*>(bt + 32 * ks);
        };
        i32x4 bfr[MH_KSTEPS];
        bfr[0] = read_b(0);
        bfr[1] = read_b(1);
        // the next tile's raw dword (requested three steps ago) leaves the ring now, with the operand reads: its LDS latency is
        // long over when the expansion behind the MFMAs needs it
        uint32_t raw_next = 0u;
        if (!PLSLAM_MH_X(128)) raw_next = take_raw(ring_slot);
        // ragged group: lanes whose class has run out of columns take the penalty from this tile on (at most two tiles of a
        // scan change anything: the group's first -- classes without any column -- and the one where the cut class ends)
        if (t >= nfull && ((t & 15) == 0 || (t & 15) == lim_part)) {
            const uint32_t pen = lane_lim() == (t & 15) ? COL_PENALTY : 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) seed[r] += pen;
            asm volatile("" : "+v"(seed));
        }
        const f32x16 cseed = __builtin_bit_cast(f32x16, seed);
        f32x16 m0, m1;
#define PLSLAM_MH_MMA(ACC, MT, KS, CIN)                                                            \
        {                                                                                          \
            const i32x8 a8 = {afrag[MT][KS].x, afrag[MT][KS].y, afrag[MT][KS].z, afrag[MT][KS].w, 0, 0, 0, 0}; \
            const i32x8 b8 = {bfr[KS].x, bfr[KS].y, bfr[KS].z, bfr[KS].w, 0, 0, 0, 0};             \
            if (!PLSLAM_MH_X(4))                                                                   \
                ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, CIN, 4, 4, 0, scale_a, 0, scale_b); \
            else { const f32x16 cin_ = CIN; ACC = cin_; ACC[KS] = __builtin_bit_cast(float, bfr[KS].x ^ a8[0]); } \
            asm volatile("" : "+v"(ACC));    /* pins the MFMA here (no instruction) */              \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
#define PLSLAM_MH_EPI2(R)                                                                          \
        {                                                                                          \
            if (!PLSLAM_MH_X(2)) { PLSLAM_MH_EPI_ROW(R) PLSLAM_MH_EPI_ROW((R) + 1) }               \
            __builtin_amdgcn_sched_barrier(0);                                                     \
        }
        __builtin_amdgcn_sched_barrier(0);
        PLSLAM_MH_EPI2(0)  PLSLAM_MH_MMA(m0, 0, 0, cseed)
        PLSLAM_MH_EPI2(2)  PLSLAM_MH_MMA(m1, 1, 0, cseed)
        bfr[2] = read_b(2);
        PLSLAM_MH_EPI2(4)  PLSLAM_MH_MMA(m0, 0, 1, m0)
        PLSLAM_MH_EPI2(6)  PLSLAM_MH_MMA(m1, 1, 1, m1)
        bfr[3] = read_b(3);
        PLSLAM_MH_EPI2(8)  PLSLAM_MH_MMA(m0, 0, 2, m0)
        PLSLAM_MH_EPI2(10) PLSLAM_MH_MMA(m1, 1, 2, m1)
        PLSLAM_MH_EPI2(12) PLSLAM_MH_MMA(m0, 0, 3, m0)
        PLSLAM_MH_EPI2(14) PLSLAM_MH_MMA(m1, 1, 3, m1)
#undef PLSLAM_MH_EPI2
#undef PLSLAM_MH_MMA
        // behind the last MFMA, in front of the pack that needs its result: the expansion of the next tile (its buffer was
        // read for the last time before this step's barrier) and the prefetch -- independent work that covers the matrix
        // pipe's latency (measured: the 16 packs cost as much time as the 33 bookkeeping ops while they sat right behind it)
        if (!PLSLAM_MH_X(128)) {
            expand_store(raw_next, (U + 1) & 1, t + 1);               // past the last tile: a harmless rewrite of the idle buffer
            load_raw_async(t + 4, ring_slot);                         // three tiles ahead of its use, into the slot just read
            ring_slot = ring_slot == 2 ? 0 : ring_slot + 1;           // (scalar)
        }
        if (with_prev) {
            // block (t - 9) / 8 of column results: its last tile was parked in the step before this one, by every wave before
            // this step's barrier; tile t - 1 is about to take the block's first slot: a second barrier (workgroup-uniform
            // branch).  (The blocks of the window before were finished behind its loop.)
            if (!DIRECTED && U == 1 && ((t - 1) & (MH_CGROUP - 1)) == 0 && t - 9 >= wt0) {
                combine_columns((t - 9) >> 3);
                if (!PLSLAM_MH_X(1)) __syncthreads();
            }
            finish_columns(t - 1, pk_min16(cm, cm1));
            // wave-uniform: tile t-1 closed a row group
            if (U == 0 && ((t - 1) & (MH_GROUP - 1)) == MH_GROUP - 1) push_groups(t - 1);
        }
        // P(t): the key pairs of tile t; the accumulators are dead from here on
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float f0 = m0[r], f1 = m1[r];
            if (!PLSLAM_MH_X(512)) kc[r] = pack_acc(f0, f1);
        }
        if (PLSLAM_MH_X(512)) { asm volatile("" ::"v"(m0), "v"(m1)); kc[0] = __builtin_bit_cast(uint32_t, (float)m0[0]); }
    };
    // E(t) on its own (the last tile of a window has no following M step to hide under)
    auto epilogue = [&](int t, auto masked_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        uint32_t cm = 0xFFFFFFFFu, cm1 = 0xFFFFFFFFu;
        PLSLAM_MH_EPI_ROW(0) PLSLAM_MH_EPI_ROW(1) PLSLAM_MH_EPI_ROW(2) PLSLAM_MH_EPI_ROW(3)
        PLSLAM_MH_EPI_ROW(4) PLSLAM_MH_EPI_ROW(5) PLSLAM_MH_EPI_ROW(6) PLSLAM_MH_EPI_ROW(7)
        PLSLAM_MH_EPI_ROW(8) PLSLAM_MH_EPI_ROW(9) PLSLAM_MH_EPI_ROW(10) PLSLAM_MH_EPI_ROW(11)
        PLSLAM_MH_EPI_ROW(12) PLSLAM_MH_EPI_ROW(13) PLSLAM_MH_EPI_ROW(14) PLSLAM_MH_EPI_ROW(15)
        finish_columns(t, pk_min16(cm, cm1));
    };
    auto pipeline = [&](auto masked_tag) __attribute__((always_inline)) {
        // (wt0 is a multiple of 64: t & 3 of the unrolled steps is static.  The first step has no previous tile: its
        // bookkeeping runs on "none" keys, its column / group actions are skipped)
#pragma unroll
        for (int r = 0; r < 16; ++r) kc[r] = 0xFFFFFFFFu;
        for (int tb = wt0; tb < wt1; tb += 4) {
            tile_step(tb, std::integral_constant<int, 0>{}, tb != wt0, masked_tag);
            if (tb + 1 < wt1) tile_step(tb + 1, std::integral_constant<int, 1>{}, true, masked_tag);
            if (tb + 2 < wt1) tile_step(tb + 2, std::integral_constant<int, 2>{}, true, masked_tag);
            if (tb + 3 < wt1) tile_step(tb + 3, std::integral_constant<int, 3>{}, true, masked_tag);
        }
        // the window's last tile opens a block of columns while the block before it still waits in the slots (the step that
        // would have combined it does not exist): combine it now
        if (!DIRECTED && ((wt1 - 1) & (MH_CGROUP - 1)) == 0 && wt1 - 9 >= wt0) {
            __syncthreads();
            combine_columns((wt1 - 9) >> 3);
            __syncthreads();
        }
        epilogue(wt1 - 1, masked_tag);
        push_groups(wt1 - 1);                      // the (possibly partial) last row group
    };
#undef PLSLAM_MH_EPI_ROW

    // Row results of a window.  Every lane holds, per packed register, the best two GROUP minima (16-bit keys
    // (d, group in window, r)) of ITS column class for two rows.  Transpose through LDS so that one lane owns one row:
    // lane l reads the 32 class entries of local row l in class order and widens them to (key16 << 16 | class), which
    // orders like (d, j) between DIFFERENT (group, class) pairs: j = 512 G + S class + tile, and every entry of a row
    // carries the same r.  The best entry names the (group, class) that holds the best column; the lane recomputes ALL
    // members of it from the raw rows (S consecutive rows of b): the first minimum is the best key, the others compete with
    // the second entry (the best key outside that (group, class)) for second best.
    auto finish_rows = [&]() __attribute__((always_inline)) {
        uint32_t* rowx = reinterpret_cast<uint32_t*>(smem) + w * (64 * ROWX_STRIDE);
        u32x2_t rb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) rb[r] = park[r * 64];
        __syncthreads();                           // every wave holds its pairs: the transpose may overwrite the parking area
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lrow = 16 * g + r;           // M-tile 0; M-tile 1: + 32
            rowx[lrow * ROWX_STRIDE + c] = __builtin_amdgcn_perm(rb[r].y, rb[r].x, 0x05040100u);          // (x.lo | y.lo << 16)
            rowx[(32 + lrow) * ROWX_STRIDE + c] = __builtin_amdgcn_perm(rb[r].y, rb[r].x, 0x07060302u);   // (x.hi | y.hi << 16)
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
        const int row = iw + lane + (lane & 32) * 3;          // lanes 32..63: M-tile 1's rows, 128 further on
        if (row < n1) {
            const gu2_t out = (gu2_t) reinterpret_cast<u32x2_t*>(sd.keys12) + row;
            const gcu32x4_t ap = (gcu32x4_t)(araw + (size_t)row * 8);
            const u32x4_t a_lo = ap[0], a_hi = ap[1];
            // (key16 << 16 | class) -> first row of the (group, class), its stride count, the distance
            auto group_of = [&](uint32_t k, uint32_t& jbase, uint32_t& cnt) {
                const uint32_t t0 = (uint32_t)wt0 + (((k >> 21) & 3u) << 4);              // first tile of the group
                const uint32_t s = t0 < (uint32_t)nfull ? (uint32_t)MH_GROUP : (uint32_t)rag_s;
                jbase = (t0 >> 4) * MH_GROUP_ROWS + s * (k & 0xFFFFu);
                cnt = s;
            };
            // all members of a (group, class): the smallest (d << 23 | j) and the second smallest
            // (four candidates' rows are requested together: 16 dependent round trips to L2 otherwise -- the loop measured 9 % of
            // the scan, almost all of it latency)
            auto rescan = [&](uint32_t jbase, uint32_t cnt, uint32_t& best, uint32_t& second) {
                best = second = KEY_NONE;
                const uint32_t left = (uint32_t)n2 - jbase;                       // >= 1: the class's first column exists
                const uint32_t nvalid = cnt < left ? cnt : left;
                const PLSLAM_GLOBAL char* const rb = bbytes + (size_t)jbase * 32;
#pragma unroll
                for (int k0_ = 0; k0_ < MH_GROUP; k0_ += 4) {
                    u32x4_t bl[4], bh[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t kk = (uint32_t)(k0_ + q) < nvalid ? (uint32_t)(k0_ + q) : nvalid - 1u;   // past the end: a duplicate, masked below
                        const gcu32x4_t bp = (gcu32x4_t)(rb + kk * 32u);
                        bl[q] = bp[0];
                        bh[q] = bp[1];
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t d = hamming256(a_lo, a_hi, bl[q], bh[q]);
                        const uint32_t cand = (uint32_t)(k0_ + q) < nvalid ? ((d << KEY_IDX_BITS) | (jbase + (uint32_t)(k0_ + q))) : KEY_NONE;
                        second = umin_(second, umax_(best, cand));
                        best = umin_(best, cand);
                    }
                }
            };
            uint32_t r0 = KEY_NONE, r1 = KEY_NONE;
            if ((k0 >> 16) <= KEY16_MAX && !PLSLAM_MH_X(16)) {
                uint32_t jb, cnt, in2;
                group_of(k0, jb, cnt);
                rescan(jb, cnt, r0, in2);
                if ((k1 >> 16) <= KEY16_MAX) {
                    // the best key outside the winner's (group, class): its distance is exact, its column is the first of
                    // its (group, class) unless the exact index was asked for and it IS the second best
                    uint32_t jb1, cnt1;
                    group_of(k1, jb1, cnt1);
                    uint32_t o1 = ((k1 >> 23) << KEY_IDX_BITS) | jb1;
                    if ((sd.flags & 1) && (o1 >> KEY_IDX_BITS) <= (in2 >> KEY_IDX_BITS)) {
                        uint32_t b1, s1;
                        rescan(jb1, cnt1, b1, s1);
                        o1 = b1;
                    }
                    r1 = umin_(in2, o1);
                } else {
                    r1 = in2;
                }
            } else if (PLSLAM_MH_X(16)) {
                r0 = k0; r1 = k1;
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
        expand_store(load_raw(wt0), 0, wt0);       // wt0 is a multiple of 128: buffer parity restarts at 0
        load_raw_async(wt0 + 1, 1);
        load_raw_async(wt0 + 2, 2);
        load_raw_async(wt0 + 3, 0);
        ring_slot = 1;                             // the slot of tile wt0 + 1
        if (!rows_ragged) pipeline(std::false_type{}); else pipeline(std::true_type{});
        __syncthreads();                           // every wave is past its last operand read of the b tile, every column minimum is parked
        // the window's last block of column results (full or partial)
        combine_columns((wt1 - 1) >> 3);
        if (!PLSLAM_MH_X(1024)) finish_rows();
        if (wt1 == ntiles) break;
        __syncthreads();                           // smem becomes the b tile (+ parking area) again
        wt0 = wt1;
        wt1 = ntiles < wt0 + MH_WINDOW ? ntiles : wt0 + MH_WINDOW;
    }
}
#endif  // PLSLAM_BUILD_LEGACY_SCANS

// K1c''  merge of K1h's column partials + the second-best recomputation: keys21[j] = best-2 over all rows of a.
// part[256-row block][slot] = (d0 << 17 | row0 << 9 | d1): the block's best row and the distance of its best row OUTSIDE
// row0's aligned group of 16 rows (FIX: the pair of words (d0 << 8 | row0, d1 << 8 | row1)).  Over the blocks: K0 = the best
// row, K1 = the best of (every other block's best row, the winner block's second entry) = the best row outside K0's group of 16.
// FIX = false: keys21[j] = (K0, K1) as they are -- K1's index is not a row then (all ones) -- and the finalize kernel completes
// the second best for the columns it needs (ProblemDesc::lazy21); FIX = true (exact key tables asked for): the 15 other
// rows of K0's group are recomputed here for every column (512 contiguous bytes of a).  PARTS lanes share a column (tall
// problems: see K1f).
template <int PARTS, bool FIX>
__global__ void __launch_bounds__(256)
k_merge_fix16(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks, int nblocks)
{
  // (a capped grid walks the block table: plslam_ctx option "post_workgroups")
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    constexpr int COLS = 256 / PARTS;
    // PARTS == 1: a lane takes SPT slots, 256 apart (one block-table entry per 1024 slots): the kernel is a chain of dependent
    // round trips (block entry -> descriptor -> partials -> store) with a few loads at its end, and four slots' loads
    // share the chain
    constexpr int SPT = PARTS == 1 ? MH_MERGE_SPT : 1;
    __shared__ uint32_t red[PARTS > 1 ? 512 : 2];
    const BlockDesc bd = blocks[blk];
    const SymDesc sd = syms[bd.item];
    // lanes walk the partial table in SLOT order (coalesced reads of every block's row); the slot's column is where the result goes
    const int jl = (int)threadIdx.x % COLS, part_id = (int)threadIdx.x / COLS;
    const MhLayout L(sd.n2);
    const gcu32_t part = (gcu32_t) sd.part21;
    const int nwb = (sd.n1 + 255) >> 8;
    const int n2p = (MH_TILE_N * L.ntiles + 255) & ~255;
    int slot[SPT], j[SPT];
    uint32_t b0[SPT], b1[SPT], s0[SPT];     // s0: the winner block's second entry, kept raw: it joins at the end
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        slot[q] = bd.row0 + jl + 256 * q;
        j[q] = slot[q] < MH_TILE_N * L.ntiles ? L.row_of(slot[q] >> 5, slot[q] & 31) : sd.n2;     // >= n2: the slot holds no column
        if (j[q] >= sd.n2) slot[q] = 0;       // (reads a valid word, drops the result)
        b0[q] = b1[q] = KEY_NONE;
        s0[q] = 0xFFFFFFFFu;
    }
    auto wide = [](uint32_t k17, uint32_t wb) -> uint32_t {         // (d << 8 | row in block) -> (d << 23 | row)
        return ((k17 >> 8) << KEY_IDX_BITS) | ((k17 & 255u) + 256u * wb);
    };
    // only a block's BEST key is widened and merged per entry
    // PLSLAM_MERGE_WB (round 5): the default form (one word per entry, a lane per slot) requests the words of up to this many
    // row blocks together -- a 1500-row problem's six in ONE round trip instead of three
#ifndef PLSLAM_MERGE_WB
#define PLSLAM_MERGE_WB 8
#endif
    if (PARTS == 1 && !FIX && PLSLAM_MERGE_WB > 2) {
        constexpr int WB = PLSLAM_MERGE_WB > 2 ? PLSLAM_MERGE_WB : 2;
        for (int wb0 = 0; wb0 < nwb; wb0 += WB) {
            uint32_t e[WB][SPT];
#pragma unroll
            for (int u = 0; u < WB; ++u) {
                if (wb0 + u < nwb) {                  // (uniform)
#pragma unroll
                    for (int q = 0; q < SPT; ++q)
                        e[u][q] = PLSLAM_NT_STREAMS ? __builtin_nontemporal_load(&part[(size_t)(wb0 + u) * n2p + slot[q]])
                                                    : part[(size_t)(wb0 + u) * n2p + slot[q]];
                }
            }
#pragma unroll
            for (int u = 0; u < WB; ++u) {
                if (wb0 + u < nwb) {
#pragma unroll
                    for (int q = 0; q < SPT; ++q) {
                        const uint32_t k = wide(e[u][q] >> 9, (uint32_t)(wb0 + u)), e1 = ((e[u][q] & 511u) << 8) | 255u;
                        s0[q] = k < b0[q] ? e1 : s0[q];
                        b1[q] = umin_(b1[q], umax_(b0[q], k));
                        b0[q] = umin_(b0[q], k);
                    }
                }
            }
        }
    } else
#pragma unroll 2
    for (int wb = part_id; wb < nwb; wb += PARTS) {
        uint32_t e0[SPT], e1[SPT];
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            if (FIX) {
                const gu2c_t pp = (gu2c_t) part + ((size_t)wb * n2p + slot[q]);
                const u32x2_t e = PLSLAM_NT_STREAMS ? __builtin_nontemporal_load(pp) : *pp;
                e0[q] = e.x; e1[q] = e.y;
            } else {
                const uint32_t e = PLSLAM_NT_STREAMS ? __builtin_nontemporal_load(&part[(size_t)wb * n2p + slot[q]])
                                                     : part[(size_t)wb * n2p + slot[q]];
                e0[q] = e >> 9; e1[q] = ((e & 511u) << 8) | 255u;     // the second entry's row is not in the word
            }
        }
#pragma unroll
        for (int q = 0; q < SPT; ++q) {
            const uint32_t k = wide(e0[q], (uint32_t)wb);
            s0[q] = k < b0[q] ? e1[q] : s0[q];
            b1[q] = umin_(b1[q], umax_(b0[q], k));
            b0[q] = umin_(b0[q], k);
        }
    }
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        if (j[q] < sd.n2 && b0[q] < (257u << KEY_IDX_BITS)) {
            const uint32_t k1 = FIX ? wide(s0[q], (b0[q] & KEY_IDX_MASK) >> 8) : (((s0[q] >> 8) << KEY_IDX_BITS) | KEY_IDX_MASK);
            b1[q] = umin_(b1[q], k1);
            if (b1[q] >= (257u << KEY_IDX_BITS)) b1[q] = KEY_NONE;
        } else {
            b0[q] = b1[q] = KEY_NONE;
        }
    }
    if (PARTS > 1) {
        // the parts' pairs: each is (best of its blocks, best outside THAT best's group): the overall best's pair partner is
        // still "outside its group", and any other part's best is outside it too (another block, or the same block's
        // other group only if it came as a second entry -- which is also outside) -- merge2 keeps exactly that
        if (blk != (int)blockIdx.x) __syncthreads();          // the entry before: every lane is past its reads of red[]
        red[2 * threadIdx.x] = b0[0];
        red[2 * threadIdx.x + 1] = b1[0];
        __syncthreads();
        if (part_id == 0) {
#pragma unroll
            for (int q = 1; q < PARTS; ++q) merge2(b0[0], b1[0], red[2 * (q * COLS + jl)], red[2 * (q * COLS + jl) + 1]);
        }
    }
#pragma unroll
    for (int q = 0; q < SPT; ++q) {
        if (part_id == 0 && j[q] < sd.n2) {
            if (FIX && b0[q] != KEY_NONE) {
                const gcu32_t araw = (gcu32_t) reinterpret_cast<const uint32_t*>(sd.a);
                const gcu32x4_t bp = (gcu32x4_t)((gcu32_t) reinterpret_cast<const uint32_t*>(sd.b) + (size_t)j[q] * 8);
                const u32x4_t b_lo = bp[0], b_hi = bp[1];
                const uint32_t i0 = b0[q] & KEY_IDX_MASK, ibase = i0 & ~15u;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const uint32_t i = ibase + (uint32_t)k;
                    const bool ok = i != i0 && i < (uint32_t)sd.n1;
                    const gcu32x4_t ap = (gcu32x4_t)(araw + (size_t)(ok ? i : i0) * 8);
                    const u32x4_t a_lo = ap[0], a_hi = ap[1];
                    const uint32_t d = hamming256(a_lo, a_hi, b_lo, b_hi);
                    b1[q] = umin_(b1[q], ok ? ((d << KEY_IDX_BITS) | i) : KEY_NONE);
                }
            }
            ((gu2_t) reinterpret_cast<u32x2_t*>(sd.keys21))[j[q]] = u32x2_t{b0[q], b1[q]};
        }
    }
  }
}

int merge_fix16_cols(int parts) { return parts >= 16 ? 16 : parts >= 4 ? 64 : 256 * MH_MERGE_SPT; }

// d_blocks: one entry per (problem, merge_fix16_cols(parts) column slots)
int launch_merge_fix16(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int parts, bool fix, hipStream_t s, int grid_cap)
{
    if (nblocks <= 0) return PLSLAM_OK;
    const int grid = grid_cap > 0 && grid_cap < nblocks ? grid_cap : nblocks;
#define PLSLAM_MH_MERGE(P) { if (fix) hipLaunchKernelGGL((k_merge_fix16<P, true>), dim3(grid), dim3(256), 0, s, d_sym, d_blocks, nblocks); \
                             else hipLaunchKernelGGL((k_merge_fix16<P, false>), dim3(grid), dim3(256), 0, s, d_sym, d_blocks, nblocks); }
    if (parts >= 16) PLSLAM_MH_MERGE(16)
    else if (parts >= 4) PLSLAM_MH_MERGE(4)
    else PLSLAM_MH_MERGE(1)
#undef PLSLAM_MH_MERGE
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

#if PLSLAM_BUILD_LEGACY_SCANS
int launch_scan_sym_mfma_h(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero, int nzero,
                           bool directed, hipStream_t s)
{
    if (nblocks <= 0) return PLSLAM_OK;
    if (directed) hipLaunchKernelGGL((k_scan_sym_mfma_h<true>), dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks, d_zero, nzero);
    else hipLaunchKernelGGL((k_scan_sym_mfma_h<false>), dim3(nblocks), dim3(256), 0, s, d_sym, d_blocks, d_zero, nzero);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}
#endif  // PLSLAM_BUILD_LEGACY_SCANS

}  // namespace plslam
