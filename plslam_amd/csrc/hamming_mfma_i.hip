// hamming_mfma_i.hip -- K1i: K1h's matrix-core symmetric Hamming kNN-2 scan (minimum-only bookkeeping in both directions,
// class-major layouts, workgroup-level column partials) with the two M-tiles of a wave software-pipelined AGAINST EACH OTHER
// (gfx950).  Round 4; the default.
//
// Contract, tables, layouts, arithmetic, partial table, merge kernel: those of K1h (hamming_mfma_h.hip): keys12[i] = best-2
// over j, part21[256-row block of a][column slot] = (d0 << 17 | row0 << 9 | d1), keys = (distance << 23) | index =
// cv::BFMatcher(NORM_HAMMING).knnMatch(k=2) order (reference call sites src/mapHandler.cpp:277,424,597,712,3223,3249).
//
// What changes is WHEN a wave's VALU work can issue.  K1h packs accumulator r of M-tile 0 with accumulator r of M-tile 1 into
// one register of two 16-bit keys: the pack needs BOTH accumulator sets complete, so a wave's tile is eight MFMAs with four
// VALU instructions between two of them (17 of 32 matrix-pipe cycles) followed by ~40 VALU instructions with no MFMA in
// flight (pack, expansion of the next b tile, address work).  Three such waves share a SIMD; whenever two of them are in
// their MFMA phase they take turns on the matrix pipe and their (in-order) VALU work trickles behind the blocked MFMAs.  PMC
// (profiles/r3_w_pmc_a.txt): VALU 70 % busy, matrix pipe 37 % -- neither unit is saturated.
//
// Here a packed register holds rows q and q + 8 of ONE M-tile's lane group (both from the same accumulator set), so the
// bookkeeping of a set depends on that set only, and the tile is two half-phases:
//     phase 1   the 4 MFMAs of M-tile 0 (tile t)   x   pack + bookkeeping of M-tile 1 (tile t-1)
//     (behind)  expansion of tile t+1, prefetch of tile t+4, column results of tile t-1 parked
//     phase 2   the 4 MFMAs of M-tile 1 (tile t)   x   pack + bookkeeping of M-tile 0 (tile t)
// Every MFMA has six independent VALU instructions behind it, no accumulator is read before the four MFMAs after it have been
// issued, and the packed key registers of K1h (16 VGPRs) disappear: a key pair is consumed the moment it is packed.  The
// column minima of a 16-row group are now the fold of the two halves of a register (rows 0-7 | rows 8-15: one more packed
// min with op_sel), parked as the same word K1h parks: everything behind the tile loop is K1h's.
#include "mfma_h_common.hpp"

#include <type_traits>

// PLSLAM_MI_F16 = 1 (default): the UNSCALED matrix instruction and three-input packed minima.
//  * fp4 codes of +-4 on both sides and v_mfma_f32_32x32x64_f8f6f4 without block scales: a product is +-16, the accumulator
//    2^23 + 32 d + tag, the 16-bit key d << 5 | tag with a FIVE-bit tag 16 g + r (the row within the lane pair's 32 rows of a
//    wave).  The wave's number, which K1h carries in the key, comes back where the workgroup's column minima are combined
//    (once per 8 tiles); the row direction never needed (g, r) and overwrites the five bits with the group number.  The
//    scaled instruction costs the VALU port ~11 cycles beside a packed-VALU stream, this one ~7, and the matrix pipe takes
//    one per 26 instead of 32.6 cycles (tools/issue_mix_microbench.hip, profiles/r4_issue_mix_microbench.txt).
//  * Every key is below 0x7C00 ("none" = 0x7BFF), i.e. a positive finite half float, and half floats of one sign order like
//    their bit patterns: gfx950's v_pk_minimum3_f16 is an exact THREE-input packed integer minimum on them, denormals
//    included (tools/pk_min3_f16_check.hip: 0 of 202 k packed triples wrong; same issue rate as v_pk_min_u16).  Column
//    direction: the 8 key pairs of an accumulator set in 4 instructions instead of 8.  Row direction: a tile's key pairs are
//    KEPT on even tiles and folded together with the odd tile's (minimum, kept, new): 8 instructions per two tiles instead of
//    16.  Per tile and M-tile pair 16 v_perm + 16 minima instead of 16 + 34.
// 0: K1h's arithmetic (scaled instruction, 7-bit tags, two-input 16-bit minima).
#ifndef PLSLAM_MI_F16
#define PLSLAM_MI_F16 1
#endif
#define PLSLAM_MI_UNSCALED PLSLAM_MI_F16
// PLSLAM_MI_ROWLOOK (with PLSLAM_MI_F16): M-tiles whose row direction keeps an even tile's pairs for the odd tile's fold --
// 0 none, 1 M-tile 1 only (8 more registers), 2 both (16 more)
#ifndef PLSLAM_MI_ROWLOOK
#define PLSLAM_MI_ROWLOOK 0
#endif
// build-time experiments (tools/build_exp.py; results are WRONG with any of them on): 1 no tile barrier, 2 no operand reads
// from LDS, 4 no MFMA, 8 no bookkeeping (pack + minima), 16 no expansion / prefetch, 32 no gathers in the row finish's
// re-evaluation of the winner cell (its arithmetic stays), 64 no re-evaluation at all, 128 (round 6, with the DIRECTED
// instantiation: `mfma_form 6`'s UPPER BOUND) the whole bookkeeping of an accumulator set = eight v_min3_f32 on the unpacked
// accumulators into two lane-local running minima per M-tile -- no pack, no row cells, no group pushes, no row finish: what a
// two-directed-pass scan with column-direction minima only could cost at best (tools/form6_bound.sh)
#ifndef PLSLAM_MI_X
#define PLSLAM_MI_X 0
#endif
#define PLSLAM_MI_LOOK(MT) (PLSLAM_MI_F16 && ((MT) == 1 ? PLSLAM_MI_ROWLOOK >= 1 : PLSLAM_MI_ROWLOOK >= 2))
// PLSLAM_MI_R5 (round 5; bit set, default all): the per-ITEM work of a wave -- by the counters 39 % of its VALU instructions
// and by the knock-out builds half of the launch -- made leaner.  Same keys, same tables.
//   1  row finish: the 32 classes of a row are merged as "best two FIRST entries + the best class's second entry" (3 VALU per
//      class instead of 6)
//   2  row finish: the winner cell's 16 columns are re-evaluated from ONE base address with immediate offsets and no per-
//      candidate validity: a cell that the end of b cuts is read from 16 columns further down instead -- the extra columns are
//      real columns of earlier cells, which can neither beat nor tie the winner (see finish_rows)
//   4  group pushes: the first push of a window only writes (nothing is parked yet); the tag replacement is one v_and_or; the
//      window's last push stays in registers and is consumed by the row finish directly
//   8  expansion: the validity mask of a ragged group's tile only from the first tile some class has no column for
//  16  the scalar bookkeeping of the prefetch is carried from step to step instead of being rebuilt: the ring slot as a byte
//      offset (one scalar add builds M0, one vector add the lane's read address), the full-group row offset (pf32)
// 0 = round 4's code (A/B builds: tools/build_exp.py hamming_mfma_i.hip r4:-DPLSLAM_MI_R5=0)
#ifndef PLSLAM_MI_R5
#define PLSLAM_MI_R5 31
#endif
// PLSLAM_MI_R6 (round 6; bit set, default all): the tile body's bookkeeping WITHOUT THE PACK.
//   1  An accumulator is the float 2^23 + key: its low half IS the 16-bit key, its high half the constant 0x4B00 -- as a half
//      float 14.0, above every real key (<= 0x3FFF) and below "none".  v_pk_minimum3_f16 takes op_sel per source, so
//          gm = pk_min3(gm, acc[q], acc[q + 8])  op_sel:[0,0,1] op_sel_hi:[1,1,0]
//      is  gm.lo = min(gm.lo, key(acc[q]), 0x4B00), gm.hi = min(gm.hi, 0x4B00, key(acc[q + 8])): the row direction's update of
//      a row PAIR straight from the two unpacked accumulators -- the v_perm that packed them (16 of the 52 VALU instructions
//      of a wave-tile) is gone; "none" and the penalty keys of columns that do not exist are capped at 0x4B00, still above
//      MI_KEY16_MAX wherever they are tested.  The column direction folds the same two accumulators into a float minimum with
//      v_min3_f32 (2^23 + key orders like the key; one chain per accumulator set, started by a two-input v_min_f32 of the set's
//      first pair; the parked word is one v_perm of the two M-tiles' minima -- the values K1h parks).  Per accumulator set
//      8 + 8 instead of 8 + 8 + 4 and one instruction instead of three per parked word: 33 VALU per wave-tile instead of 43
//      (directed: 16 instead of 32).  The accumulators are inline-asm operands now: the
//      compiler does not count MFMA -> VALU wait states for them, tools/check_mfma_hazards.py does (tests/test_abi.py).
// 0 = round 5's code
//   2  (experiment) the b tile in LDS without the 16 padding bytes per row: rows of 128 bytes whose 16-byte chunks are XOR-
//      swizzled with the row number (chunk ^ (row & 7) ^ (row >> 3 & 1): conflict-free b128 operand reads and expansion stores on
//      32 or 64 banks) -- the kilobyte this frees is the FOURTH slot of the prefetch ring (the slot becomes a compile-time fact
//      of every unrolled step, a tile's raw words are requested one step earlier) at the same three workgroups per CU
// 0 = round 5's code
#ifndef PLSLAM_MI_R6
#define PLSLAM_MI_R6 1
#endif
#if PLSLAM_MI_R6 & 2
#define PLSLAM_MI_SWZ 1
#ifndef PLSLAM_MI_RING4
#define PLSLAM_MI_RING4 1
#endif
#else
#define PLSLAM_MI_SWZ 0
#endif
#if PLSLAM_MI_F16 && (PLSLAM_MI_R6 & 1) && !PLSLAM_MI_ROWLOOK
#define PLSLAM_MI_NOPACK 1
#else
#define PLSLAM_MI_NOPACK 0
#endif
// PLSLAM_MI_PERSIST = N > 0 (experiment): at most N persistent workgroups, each walking its XCD's row of the block table
#ifndef PLSLAM_MI_PERSIST
#define PLSLAM_MI_PERSIST 0
#endif
// PLSLAM_MI_PRIO = 1 (default): s_setprio 1 around the tile loops (0 = none)
#ifndef PLSLAM_MI_PRIO
#define PLSLAM_MI_PRIO 1
#endif

namespace plslam {

namespace {
#if PLSLAM_MI_UNSCALED
constexpr uint32_t MI_MAG = FP4_FOUR;
constexpr int MI_DSHIFT = 5;                                  // key16 = d << 5 | tag5
constexpr uint32_t MI_ACC_BITS = 0x4B000000u + 4096u;         // float bits of 2^23 + 4096: the contraction is 32 d - 4096
constexpr uint32_t MI_KEY16_MAX = 0x3FFFu;                    // real keys end at 256 << 5 | 31
constexpr uint32_t MI_COL_PENALTY = 0x4000u;                  // zero codes ("distance 128": 0x1000 + tag) + this: above every real key, below the wrap
constexpr int MI_SCALE_A = 0, MI_SCALE_B = 0;                 // both zero: the compiler selects the unscaled instruction
constexpr uint32_t MI_NONE16 = 0x7BFFu;                       // the largest finite half float: above every key and every penalty key
#else
constexpr uint32_t MI_MAG = FP4_ONE;
constexpr int MI_DSHIFT = 7;
constexpr uint32_t MI_ACC_BITS = ACC_BITS;
constexpr uint32_t MI_KEY16_MAX = KEY16_MAX;
constexpr uint32_t MI_COL_PENALTY = COL_PENALTY;
constexpr int MI_SCALE_A = SCALE_A, MI_SCALE_B = SCALE_B;
constexpr uint32_t MI_NONE16 = 0xFFFFu;
#endif
constexpr uint32_t MI_NONE32 = MI_NONE16 * 0x00010001u;
#ifndef PLSLAM_MI_RESCAN_BATCH
#define PLSLAM_MI_RESCAN_BATCH 2
#endif
constexpr int MI_RESCAN_BATCH = PLSLAM_MI_RESCAN_BATCH;
#ifndef PLSLAM_MI_RESCAN_BATCH16
#define PLSLAM_MI_RESCAN_BATCH16 2
#endif
constexpr int MI_RESCAN_BATCH16 = PLSLAM_MI_RESCAN_BATCH16;      // the unguarded rescan's rows in flight (8 VGPRs each)
__device__ __forceinline__ uint32_t pk_min3_f16(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
}  // namespace

// PLSLAM_MI_PROF (experiment builds only, tools/k1i_profile.py): every wave adds the shader cycles (s_memtime) it spends in
// its prologue, in the tile loops and behind them (column combine, row finish) to g_mi_prof, and its tile count
#ifdef PLSLAM_MI_PROF
constexpr int MI_PROF_WGS = 65536;
__device__ unsigned long long g_mi_prof[MI_PROF_WGS * 4];      // per workgroup (wave 0): prologue, loops, finish, tiles
#define PLSLAM_MI_TICK() __builtin_readcyclecounter()
#endif

// DIRECTED = true: only keys12 (row direction) is produced.
// PLSLAM_MI_VGPRS (round 6): the kernel's register budget.  Three waves per SIMD allow 168 -- and leave 8 of a SIMD's 512
// registers per lane free, so a wave of the stages behind the scan (k_merge_fix16, k_finalize: the NEXT step's scan runs beside
// them) starts only where a scan wave has left.  Without the pack the scan fits 144 with no spill (80 registers per lane free on
// every SIMD: one stage wave per SIMD BESIDE three scan waves) -- built and measured with the stage stream above, below and level
// with the scan stream (profiles/r6_e_*, r6_f_*, r6_g_*): the stage kernels then do run beside the scan from its first
// workgroup on, and the step does not move (2.45 ms in every configuration, the scan 2.40-2.44 ms inside the loop against 2.2 alone).
// What the stages cost the step is their INSTRUCTIONS (96 M wave-level VALU instructions per step against the scan's 0.81 G:
// 11 %, and the in-loop scan is 9-10 % longer than alone), wherever their waves sit: the scan is bound by instruction issue.
// Default 168 (the compiler's own choice under three waves per SIMD).
// (amdgpu_num_vgpr counts the architected half of gfx950's unified file: the attribute takes budget / 2.)
#ifndef PLSLAM_MI_VGPRS
#define PLSLAM_MI_VGPRS 168
#endif
// PLSLAM_MI_TAIL = 1 (experiment build; NOT in the product library: it needs agent-scope fences, which
// tests/test_abi.py keeps out of every kernel): the LAST workgroup of a problem to finish -- one atomic on a per-problem counter,
// agent-scope release before it, acquire behind it on that workgroup only -- merges the problem's row-block partials into keys21
// itself (what k_merge_fix16<1, false> does), and the plan run launches no merge kernel (context option "scan_tail").
#ifndef PLSLAM_MI_TAIL
#define PLSLAM_MI_TAIL 0
#endif
#if PLSLAM_MI_TAIL
#define PLSLAM_MI_TAIL_PARAM , int32_t* __restrict__ tail_counts
#else
#define PLSLAM_MI_TAIL_PARAM
#endif
template <bool DIRECTED>
__global__ void __launch_bounds__(256, 3) __attribute__((amdgpu_num_vgpr(PLSLAM_MI_VGPRS / 2)))
k_scan_sym_mfma_i(const SymDesc* __restrict__ syms, const BlockDesc* __restrict__ blocks, int32_t* __restrict__ zero, int nzero, int nblocks
                  PLSLAM_MI_TAIL_PARAM)
{
    // one buffer, two lives: during the scan the double-buffered b tile (9 216 B) followed by the PARKED sorted pairs of the
    // row direction ([wave][slot][lane] x 8 B = 32 768 B); after the scan the row-result transpose [wave][row 0..63][33]
    constexpr int ROWX_STRIDE = 33;               // dwords per row: lane = row reads are conflict-free
    constexpr int MI_ROW_STRIDE = PLSLAM_MI_SWZ ? 128 : MH_ROW_STRIDE, MI_TILE_BYTES = MH_TILE_N * MI_ROW_STRIDE;
    constexpr int PARK_OFF = 2 * MI_TILE_BYTES;
    constexpr int SMEM_BYTES = PARK_OFF + 4 * 16 * 64 * 8;
    static_assert(SMEM_BYTES >= 4 * 64 * ROWX_STRIDE * 4, "the transpose must fit");
    __shared__ __attribute__((aligned(16))) uint8_t smem[SMEM_BYTES];
    // column minima of the last 8 tiles, [tile & 7][wave][lane] (1 KB per tile): see K1h
    __shared__ __attribute__((aligned(16))) uint32_t colstage[MH_CGROUP * 256];
    // raw b dwords in flight (LDS-DMA ring, 3 slots): see K1h.  (A fourth slot would make the slot a compile-time fact of every
    // unrolled step -- 10 scalar instructions less per tile -- but the kilobyte takes the kernel from 26 to 27 LDS granules
    // of 2 KB: TWO workgroups per CU instead of three, measured 2.84 against 2.52 ms; PLSLAM_MI_RING4 builds it.)
#ifndef PLSLAM_MI_RING4
#define PLSLAM_MI_RING4 0
#endif
    constexpr int RING = PLSLAM_MI_RING4 ? 4 : 3;
    __shared__ __attribute__((aligned(16))) uint32_t rawring[RING][256];
    uint8_t* const btile = smem;
    u32x2_t* const park = reinterpret_cast<u32x2_t*>(smem + PARK_OFF) + (threadIdx.x >> 6) * (16 * 64) + (threadIdx.x & 63);

#ifdef PLSLAM_MI_PROF
    const unsigned long long prof_t0 = PLSLAM_MI_TICK();
    unsigned long long prof_loop = 0, prof_fin = 0, prof_tl = 0;
#endif
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nzero; i += 256) zero[i] = 0;
#if PLSLAM_MI_F16
    // v_pk_minimum3_f16 is used as an exact integer minimum on 16-bit keys; keys of distances below 32 are half-precision
    // DENORMALS, which a wave in flush mode would read as zero.  HIP's default mode preserves them; the kernel does not
    // depend on that: MODE.FP_DENORM[3:2] (f16 / f64, bits 7:6 of the MODE register) = 3, in and out, for this wave.
    // (Nothing else here is touched by it: the matrix instruction works on integers far above the denormal range, the row
    // kernels' f64 arithmetic lives in other kernels.)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 6, 2), 3");
#endif

#if PLSLAM_MI_PERSIST
    // PERSISTENT workgroups (experiment): workgroup p stays on its XCD's row of the table (p & 7) and walks it in steps of
    // gridDim.x / 8 entries
    for (int vb = blockIdx.x; vb < nblocks; vb += gridDim.x) {
    if (vb != (int)blockIdx.x) __syncthreads();    // the item before: every wave is past its reads of the LDS buffers
    const int wg = (vb & 7) * (nblocks >> 3) + (vb >> 3);
    const BlockDesc bd = blocks[wg];
    if (bd.item < 0) continue;                     // padding entry of the XCD-striped table
#else
    (void)nblocks;
    const int wg = xcd_remap_(blockIdx.x, gridDim.x);
    const BlockDesc bd = blocks[wg];
    if (bd.item < 0) return;                       // padding entry of the XCD-striped table
#endif
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

    i32x4 afrag[2][MH_KSTEPS];            // the A operands (filled behind the first requests for b: see below)

    // row-direction state per SLOT s = 8 mt + q (low half: row 16 g + q of the lane's group of M-tile mt, high half: row
    // 16 g + q + 8 of the same group):
    //   gm[s]    running minimum of the 16-bit keys (d << 5 | 16 g + r; scaled form: d << 7 | 32 w + 16 g + r) of the current
    //            group of 16 tiles, column class c
    //   park[s]  (LDS) the best two GROUP minima (d << 5 | group in window; scaled form: d << 7 | group << 5 | 16 g + r) of the
    //            lane's column class
    //   kp[s]    (PLSLAM_MI_F16) the key pair of the even tile before, waiting for the odd tile's to be folded in with it
    uint32_t gm[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) gm[s] = MI_NONE32;
#if PLSLAM_MI_F16
    uint32_t kp[16];       // (only the slots of the M-tiles that look back are ever touched)
#endif

    // accumulator start: 2^23 + 4096 + 16 g + r (scaled form: 2^23 + 16384 + 32 w + 16 g + r), constant per register and lane
    // (in VECTOR registers: see K1h)
    u32x16 seed;
#pragma unroll
    for (int r = 0; r < 16; ++r) seed[r] = MI_ACC_BITS + (uint32_t)((PLSLAM_MI_UNSCALED ? 0 : 32 * w) + 16 * g + r);
    asm volatile("" : "+v"(seed));
    const int rest = n2 - (nfull >> 4) * MH_GROUP_ROWS;
    auto lane_lim = [&]() __attribute__((always_inline)) -> int {
        const int cc = (int)(threadIdx.x & 31u), v = rest - rag_s * cc;
        return rag_s == 0 ? 0 : (v < 0 ? 0 : (v > rag_s ? rag_s : v));
    };
    const int lim_part = rag_s ? rest % rag_s : 0;                 // the one class that is cut (wave-uniform): its lim, 0 = none is
    // the first tile of the ragged group in which some class has no column (the classes' column counts fall with the class
    // number: the last class's count); rag_s when every class is full
    const int mask_from = (PLSLAM_MI_R5 & 8) ? (rest - 31 * rag_s < 0 ? 0 : rest - 31 * rag_s) : 0;

    const bool block_ragged = bd.row0 + 256 > n1;  // workgroup-uniform: some groups of 16 rows may hold no row of a at all
    const bool wide_part = !DIRECTED && (sd.flags & 1);
    const gu32_t part = DIRECTED ? (gu32_t) nullptr : (gu32_t) sd.part21 + (size_t)(bd.row0 >> 8) * n2p * (wide_part ? 2 : 1);
    uint32_t* const cstage = colstage + 64 * w;        // + 256 (tile & 7) + lane
#pragma unroll
    for (int k = 0; k < 2; ++k) *reinterpret_cast<i32x4*>(colstage + 8 * tid + 4 * k) = i32x4{-1, -1, -1, -1};   // (two barriers before the first use)

    // expansion duty of this lane: the b row of class (tid >> 3) of the tile, dword (tid & 7) of it (K1h)
    const int ej = tid >> 3, ewd4 = (tid & 7) * 4;
    const PLSLAM_GLOBAL char* const bbytes = (const PLSLAM_GLOBAL char*) sd.b;
    auto load_raw = [&](int t) __attribute__((always_inline)) -> uint32_t {
        const int tc = t < ntiles ? t : ntiles - 1;
        const int gbase = (tc >> 4) * MH_GROUP_ROWS;
        const uint32_t s = tc < nfull ? (uint32_t)MH_GROUP : (uint32_t)rag_s;
        uint32_t row = __umul24((uint32_t)ej, s) + (uint32_t)(tc & 15);
        const uint32_t last = (uint32_t)(n2 - 1 - gbase);
        row = row < last ? row : last;
        return *reinterpret_cast<gcu32_t>(bbytes + (size_t)gbase * 32 + (row * 32u + (uint32_t)ewd4));
    };
    // the steady loop's loads: LDS-DMA through inline asm with the matching wait issued by hand (vmcnt(2): K1h explains why).
    // The address: ONE scalar base (b) + a 32-bit byte offset per lane (rows of b are below 2^23); M0 -- the LDS destination --
    // is the compiler's reserved register, which it does not use in this kernel (gfx9 LDS instructions do not need it): it
    // is declared clobbered instead of being saved and restored (tests/test_abi.py: no other m0 in the kernel's ISA).
    // (slot_b: the ring slot as a BYTE offset, 1024 x slot)
    auto load_raw_async = [&](int t, uint32_t slot_b) __attribute__((always_inline)) {
        const int tc = t < ntiles ? t : ntiles - 1;
        const uint32_t s = tc < nfull ? (uint32_t)MH_GROUP : (uint32_t)rag_s;
        const uint32_t first = ((uint32_t)(tc & ~15) << 5) + (uint32_t)(tc & 15);      // the group's first row + the tile within the group
        uint32_t row = __umul24((uint32_t)ej, s) + first;
        const uint32_t last = (uint32_t)(n2 - 1);
        row = row < last ? row : last;
        const uint32_t voff = row * 32u + (uint32_t)ewd4;
        const uint32_t lds_dst = (uint32_t)(uintptr_t)(&rawring[0][64 * w]) + slot_b;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(bbytes), "s"(lds_dst) : "memory", "m0");
    };
    // (the lane's own dword: its index 8 ej + ewd4 / 4 = tid from the two values the expansion keeps anyway -- a register
    // holding tid through the tile loop is spilled, and the reload waits for vmcnt(0): the whole prefetch)
    // the same inside FULL groups (tile t and the group it lies in: 16 tiles of 32 columns that all exist): no clamps, the lane's
    // part of the offset is a constant -- one vector instruction
    // (PLSLAM_MI_R5 & 16: the scalar part of the offset is carried from step to step -- pf32 -- instead of being rebuilt)
    uint32_t pf32 = 0;
    auto load_raw_async_full = [&](int t, uint32_t slot_b) __attribute__((always_inline)) {
        const uint32_t first32 = (PLSLAM_MI_R5 & 16) ? pf32 : (((uint32_t)(t & ~15) << 5) + (uint32_t)(t & 15)) * 32u;     // (scalar)
        const uint32_t voff = (uint32_t)(ej * (MH_GROUP * 32) + ewd4) + first32;
        const uint32_t lds_dst = (uint32_t)(uintptr_t)(&rawring[0][64 * w]) + slot_b;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" : : "v"(voff), "s"(bbytes), "s"(lds_dst) : "memory", "m0");
    };
    auto take_raw = [&](uint32_t slot_b) __attribute__((always_inline)) -> uint32_t {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(&rawring[0][0]) + slot_b + (ej * 32 + ewd4));
    };
    auto expand_store = [&](uint32_t raw, int buf, int tn, bool full = false) __attribute__((always_inline)) {
        uint8_t* dst = PLSLAM_MI_SWZ ? btile + buf * MI_TILE_BYTES + ej * MI_ROW_STRIDE + 16 * ((tid & 7) ^ (ej & 7) ^ ((ej >> 3) & 1))
                                     : btile + buf * MI_TILE_BYTES + ej * MI_ROW_STRIDE + ewd4 * 4;
        i32x4 v = expand_dword_fp4<false, MI_MAG>(raw);
        if (!full && tn >= nfull && (tn & 15) >= mask_from) {       // wave-uniform
            const int vm = (int)(__umul24((uint32_t)rag_s, (uint32_t)ej) + (uint32_t)(tn & 15)) < rest ? -1 : 0;
            v &= i32x4{vm, vm, vm, vm};
        }
        *reinterpret_cast<i32x4*>(dst) = v;
    };

#if PLSLAM_MI_SWZ
    uint32_t bswz[MH_KSTEPS];
    {
        const uint32_t key = (uint32_t)((c & 7) ^ ((c >> 3) & 1));
#pragma unroll
        for (int ks = 0; ks < MH_KSTEPS; ++ks) bswz[ks] = (uint32_t)(c * MI_ROW_STRIDE) + 16u * ((uint32_t)(2 * ks + g) ^ key);
    }
#endif
    int wt0 = 0, wt1 = ntiles < MH_WINDOW ? ntiles : MH_WINDOW;      // the current window of tiles
    uint32_t ring_slot = 1024;                                       // rawring slot of the NEXT tile (as a byte offset)

    // block kb of 8 tiles of column results -> one word per column for the workgroup's 256 rows (K1h's combine_columns)
    const uint64_t part_u = (uint64_t)(uintptr_t)part;
    const uint64_t part_s = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(part_u >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)part_u);
    auto combine_columns = [&](int kb) __attribute__((always_inline)) {
        if (DIRECTED) return;
        uint32_t l_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=&v"(l_));
        // column slot 64 w + l_ of the block: tile 2 w + (l_ >> 5), class l_ & 31
        const uint32_t* const src = colstage + 512 * w + (((l_ & 32u) << 3) | (l_ & 31u));
        uint32_t p[8];
#pragma unroll
        for (int v = 0; v < 4; ++v) { p[2 * v] = src[64 * v]; p[2 * v + 1] = src[64 * v + 32]; }
        if (block_ragged) {
            // p[2 v + h] = the minima of the groups of rows row0 + 32 v + 16 h .. + 15 (low half) and 128 further on (high half):
            // a group past the end of a holds duplicates of its last row only
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int base = bd.row0 + 32 * (q >> 1) + 16 * (q & 1);                   // (scalar)
                const uint32_t strike = (base >= n1 ? MI_NONE16 : 0u) | (base + 128 >= n1 ? MI_NONE16 << 16 : 0u);
                p[q] = pk_max16(p[q], strike);
            }
        }
        auto pk_merge = [](uint32_t& a0, uint32_t& a1, uint32_t c0, uint32_t c1) {
            const uint32_t m = pk_max16(a0, c0);
            a0 = pk_min16(a0, c0);
            a1 = pk_min16(m, pk_min16(a1, c1));
        };
        uint32_t k0, k1;
#if PLSLAM_MI_UNSCALED
        // p[2 v + h]: wave v, lane half h; the keys (d << 5 | 16 h + r) of one half-word order like (d, row) WITHIN a wave only.
        if (!wide_part) {
            // the waves' sorted pairs (within a wave the keys order like (d, row)) ...
            uint32_t lo[4], hi[4], enc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { lo[q] = pk_min16(p[2 * q], p[2 * q + 1]); hi[q] = pk_max16(p[2 * q], p[2 * q + 1]); }
            // ... the best ROW over the waves: a wave's best key with the wave's number between distance and tag --
            // (d << 7 | 32 v + 16 h + r), K1h's key: x + 3 (x & ~31) + 32 v, SATURATING: "none" (0x7BFF) and the keys of penalised
            // columns (0x5000 ...) become 0xFFFF, above every key ...
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t t;
                asm("v_pk_add_u16 %0, %1, %2" : "=v"(t) : "v"(lo[q]), "v"((uint32_t)(32 * q) * 0x00010001u));
                asm("v_pk_mad_u16 %0, %1, %2, %3 clamp" : "=v"(enc[q]) : "v"(lo[q] & 0xFFE0FFE0u), "v"(0x00030003u), "v"(t));
            }
            const uint32_t best = pk_min16(pk_min16(enc[0], enc[1]), pk_min16(enc[2], enc[3]));
            // ... and the second smallest of the 8 group minima as a VALUE: whichever wave it comes from, its distance is that
            // of the best row outside the best row's group (all that is kept of it)
            pk_merge(lo[0], hi[0], lo[1], hi[1]);
            pk_merge(lo[2], hi[2], lo[3], hi[3]);
            pk_merge(lo[0], hi[0], lo[2], hi[2]);
            // best: (d << 7 | t) -> (d << 8 | row in the block) as in K1h; second: (d << 5 | t) -> d << 8
            const uint32_t e0 = best & 0xFFFFu, u0 = best >> 16, e1 = hi[0] & 0xFFE0u, u1 = (hi[0] >> 16) & 0xFFE0u;
            k0 = e0 + (e0 & 0xFF80u);
            k1 = e1 << 3;
            merge2(k0, k1, u0 + (u0 & 0xFF80u) + 128u, u1 << 3);
        } else {
            // exact key tables (diagnostics): every word gets its wave first -- (d << 7 | 32 v + 16 h + r), K1h's key -- and K1h's
            // network orders them
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t t = p[q] & 0x001F001Fu;
                uint32_t sh;
                asm("v_pk_lshlrev_b16 %0, %1, %2" : "=v"(sh) : "v"(0x00020002u), "v"(p[q] & 0xFFE0FFE0u));
                p[q] = sh | t | ((uint32_t)(q >> 1) << 5) * 0x00010001u;
            }
            uint32_t lo[4], hi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { lo[q] = pk_min16(p[2 * q], p[2 * q + 1]); hi[q] = pk_max16(p[2 * q], p[2 * q + 1]); }
            pk_merge(lo[0], hi[0], lo[1], hi[1]);
            pk_merge(lo[2], hi[2], lo[3], hi[3]);
            pk_merge(lo[0], hi[0], lo[2], hi[2]);
            const uint32_t e0 = lo[0] & 0xFFFFu, e1 = hi[0] & 0xFFFFu, u0 = lo[0] >> 16, u1 = hi[0] >> 16;
            k0 = e0 + (e0 & 0xFF80u); k1 = e1 + (e1 & 0xFF80u);
            merge2(k0, k1, u0 + (u0 & 0xFF80u) + 128u, u1 + (u1 & 0xFF80u) + 128u);
        }
#else
        uint32_t lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { lo[q] = pk_min16(p[2 * q], p[2 * q + 1]); hi[q] = pk_max16(p[2 * q], p[2 * q + 1]); }
        pk_merge(lo[0], hi[0], lo[1], hi[1]);
        pk_merge(lo[2], hi[2], lo[3], hi[3]);
        pk_merge(lo[0], hi[0], lo[2], hi[2]);
        const uint32_t e0 = lo[0] & 0xFFFFu, e1 = hi[0] & 0xFFFFu, u0 = lo[0] >> 16, u1 = hi[0] >> 16;
        k0 = e0 + (e0 & 0xFF80u); k1 = e1 + (e1 & 0xFF80u);
        merge2(k0, k1, u0 + (u0 & 0xFF80u) + 128u, u1 + (u1 & 0xFF80u) + 128u);
#endif
        const uint32_t slot4 = (uint32_t)(256 * kb + 64 * w) * 4u;                             // (scalar) n2p is a multiple of 256
        if (!wide_part) {
            // ("none" and penalty keys widen to distances above 511: the word's fields are 17 + 6 and 9 bits)
            const uint32_t e = (umin_(k0, 0x1FFFFu) << 9) | umin_(k1 >> 8, 511u);
            PLSLAM_GLOBAL uint32_t* dst = (PLSLAM_GLOBAL uint32_t*)((PLSLAM_GLOBAL char*)(uintptr_t)(part_s + slot4) + 4u * l_);
#if PLSLAM_MI_TAIL == 2      // (the partial written THROUGH to memory, agent scope: no L2 write-back fence before the counter)
            __hip_atomic_store((uint32_t*)dst, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#else
            if (PLSLAM_NT_STREAMS) __builtin_nontemporal_store(e, dst);
            else *dst = e;
#endif
        } else {
            const u32x2_t e = {k0, k1};
            PLSLAM_GLOBAL u32x2_t* dst = (PLSLAM_GLOBAL u32x2_t*)((PLSLAM_GLOBAL char*)(uintptr_t)(part_s + 2u * slot4) + 8u * l_);
            if (PLSLAM_NT_STREAMS) __builtin_nontemporal_store(e, dst);
            else *dst = e;
        }
    };
    // a row group is over: its minima get the group number and go into the parked sorted pairs; the minima restart.
    // FINAL (the window's last, possibly partial, group): the merged pairs stay in registers (rb) for the row finish.
    auto push_groups = [&](int t, auto final_tag, u32x2_t* rb) __attribute__((always_inline)) {
        constexpr bool FINAL = decltype(final_tag)::value;
        if (PLSLAM_MI_X & 128) { if (FINAL) { for (int s = 0; s < 16; ++s) rb[s] = u32x2_t{gm[s], 0u}; } return; }
        const uint32_t grp = (uint32_t)(((t - wt0) >> 4) & 3);
#if PLSLAM_MI_UNSCALED
        const uint32_t gtag = grp * 0x00010001u;        // the group number takes the tag's five bits
#else
        const uint32_t gtag = (uint32_t)(((grp ^ (uint32_t)w) & 3) << 5) * 0x00010001u;   // ... the tag's wave bits (an XOR)
#endif
#if PLSLAM_MI_UNSCALED && (PLSLAM_MI_R5 & 4)
        uint32_t gtv = gtag;
        asm volatile("" : "+v"(gtv));                   // in a vector register: the mask is the instruction's one scalar operand
        auto retag = [&](uint32_t x) -> uint32_t {
            uint32_t r;
            asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(0xFFE0FFE0u), "v"(gtv));
            return r;
        };
        if (grp == 0) {                                 // (wave-uniform) the window's first push: nothing is parked yet
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const u32x2_t v = {retag(gm[s]), 0xFFFFFFFFu};
                if (FINAL) rb[s] = v;
                else park[s * 64] = v;
                gm[s] = MI_NONE32;
            }
            return;
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const u32x2_t v = park[s * 64];
            uint32_t b0 = v.x, b1 = v.y;
            pk_push2(b0, b1, retag(gm[s]));
            if (FINAL) rb[s] = u32x2_t{b0, b1};
            else park[s * 64] = u32x2_t{b0, b1};
            gm[s] = MI_NONE32;
        }
#else
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const u32x2_t v = park[s * 64];
            uint32_t b0 = v.x, b1 = v.y;
            pk_push2(b0, b1, PLSLAM_MI_UNSCALED ? ((gm[s] & 0xFFE0FFE0u) | gtag) : (gm[s] ^ gtag));
            park[s * 64] = u32x2_t{b0, b1};
            gm[s] = MI_NONE32;
        }
        if (FINAL) {
#pragma unroll
            for (int s = 0; s < 16; ++s) rb[s] = park[s * 64];
        }
#endif
    };
    // column minima of a finished tile: c0 / c1 = the packed minima (rows 0-7 | rows 8-15 of the lane's group) of M-tile 0 / 1;
    // parked word = (group minimum of M-tile 0 | group minimum of M-tile 1 << 16), K1h's
    auto finish_columns = [&](int t, uint32_t c0, uint32_t c1) __attribute__((always_inline)) {
        if (DIRECTED) { asm volatile("" ::"v"(c0), "v"(c1)); return; }
        if (PLSLAM_MI_NOPACK) {      // c0 / c1: the float minima (2^23 + key) of the lane's 16 rows of M-tile 0 / 1
            cstage[(t & (MH_CGROUP - 1)) * 256 + lane] = __builtin_amdgcn_perm(c1, c0, 0x05040100u);
            return;
        }
        const uint32_t lo = __builtin_amdgcn_perm(c1, c0, 0x05040100u);      // (c0.lo | c1.lo << 16)
        const uint32_t hi = __builtin_amdgcn_perm(c1, c0, 0x07060302u);      // (c0.hi | c1.hi << 16)
        cstage[(t & (MH_CGROUP - 1)) * 256 + lane] = pk_min16(lo, hi);
    };

    // Bookkeeping of slots (MT, Q) and (MT, Q + 1): the accumulators Q and Q + 8 of set ACC become one packed key pair, consumed
    // at once.  PAR: the tile's parity.  (Rows of a that do not exist are clamped duplicates of the last row: inside that
    // row's own group of 16 they lose every tie to it, and the groups that hold nothing else are struck out where the column
    // minima are combined -- no masking per tile.)
#if PLSLAM_MI_F16
    // row direction: even tile -- the pair is kept; odd tile -- minimum, kept pair and new pair in ONE instruction.  Column
    // direction: both pairs and the running minimum in ONE instruction.
#define PLSLAM_MI_EPI2(ACC, MT, Q, PAR)                                                            \
    if (PLSLAM_MI_X & 8) { if ((Q) == 0) { asm volatile("" :: "v"(ACC)); cma = __builtin_bit_cast(uint32_t, (float)ACC[0]); } } else \
    if (PLSLAM_MI_X & 128) {                                                                       \
        asm("v_min3_f32 %0, %0, %1, %2" : "+v"(f6[2 * (MT)]) : "v"(ACC[Q]), "v"(ACC[(Q) + 8]));    \
        asm("v_min3_f32 %0, %0, %1, %2" : "+v"(f6[2 * (MT) + 1]) : "v"(ACC[(Q) + 1]), "v"(ACC[(Q) + 9])); \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } else if (PLSLAM_MI_NOPACK) {                                                                 \
        asm("v_pk_minimum3_f16 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "+v"(gm[8 * (MT) + (Q)]) : "v"(ACC[Q]), "v"(ACC[(Q) + 8])); \
        asm("v_pk_minimum3_f16 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "+v"(gm[8 * (MT) + (Q) + 1]) : "v"(ACC[(Q) + 1]), "v"(ACC[(Q) + 9])); \
        if (!DIRECTED) {   /* ONE chain, started by the set's first pair: no initial value, no join */ \
            if ((Q) == 0) asm("v_min_f32 %0, %1, %2" : "=v"(cfa) : "v"(ACC[Q]), "v"(ACC[(Q) + 8])); \
            else asm("v_min3_f32 %0, %0, %1, %2" : "+v"(cfa) : "v"(ACC[Q]), "v"(ACC[(Q) + 8]));    \
            asm("v_min3_f32 %0, %0, %1, %2" : "+v"(cfa) : "v"(ACC[(Q) + 1]), "v"(ACC[(Q) + 9]));   \
        }                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    } else                                                                                         \
    {                                                                                              \
        uint32_t kc0 = pack_acc(ACC[Q], ACC[(Q) + 8]), kc1 = pack_acc(ACC[(Q) + 1], ACC[(Q) + 9]); \
        if (!PLSLAM_MI_LOOK(MT)) {                                                                 \
            gm[8 * (MT) + (Q)] = pk_min16(gm[8 * (MT) + (Q)], kc0);                                \
            gm[8 * (MT) + (Q) + 1] = pk_min16(gm[8 * (MT) + (Q) + 1], kc1);                        \
        } else if ((PAR) == 0) { kp[8 * (MT) + (Q)] = kc0; kp[8 * (MT) + (Q) + 1] = kc1; }         \
        else {                                                                                     \
            gm[8 * (MT) + (Q)] = pk_min3_f16(gm[8 * (MT) + (Q)], kp[8 * (MT) + (Q)], kc0);         \
            gm[8 * (MT) + (Q) + 1] = pk_min3_f16(gm[8 * (MT) + (Q) + 1], kp[8 * (MT) + (Q) + 1], kc1); \
        }                                                                                          \
        if (!DIRECTED) cma = pk_min3_f16(cma, kc0, kc1);                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
#else
    // ONE packed min into the row direction's group minimum, ONE into the column direction's (two chains: a packed op that
    // reads the result of the packed op two slots earlier costs an s_nop).
#define PLSLAM_MI_EPI(ACC, MT, Q, CM)                                                              \
    {                                                                                              \
        uint32_t kcv = pack_acc(ACC[Q], ACC[(Q) + 8]);                                             \
        gm[8 * (MT) + (Q)] = pk_min16(gm[8 * (MT) + (Q)], kcv);                                    \
        if (!DIRECTED) CM = pk_min16(CM, kcv);                                                     \
    }
#define PLSLAM_MI_EPI2(ACC, MT, Q, PAR)                                                            \
    {                                                                                              \
        PLSLAM_MI_EPI(ACC, MT, Q, cma) PLSLAM_MI_EPI(ACC, MT, (Q) + 1, cmb)                        \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
#endif
#define PLSLAM_MI_MMA(ACC, MT, KS, CIN)                                                            \
    {                                                                                              \
        const i32x8 a8 = {afrag[MT][KS].x, afrag[MT][KS].y, afrag[MT][KS].z, afrag[MT][KS].w, 0, 0, 0, 0}; \
        const i32x8 b8 = {bfr[KS].x, bfr[KS].y, bfr[KS].z, bfr[KS].w, 0, 0, 0, 0};                 \
        if (!(PLSLAM_MI_X & 4)) ACC = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, CIN, 4, 4, 0, MI_SCALE_A, 0, MI_SCALE_B); \
        else { const f32x16 cin_ = CIN; ACC = cin_; ACC[KS] = __builtin_bit_cast(float, b8[0] ^ a8[0]); } \
        asm volatile("" : "+v"(ACC));    /* pins the MFMA here (no instruction) */                  \
        __builtin_amdgcn_sched_barrier(0);                                                         \
    }
    // loop-carried: the accumulators of M-tile 1 of the tile before (all "none" at a window's start) and the packed column
    // minima of its M-tile 0
    f32x16 m1;
    uint32_t cm0_prev = MI_NONE32;
    float f6[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};      // (PLSLAM_MI_X & 128 only)
    //   step(t) = barrier | operand reads | M0(t) x E1(t-1) | expand(t+1), prefetch(t+4) | columns(t-1) [| combine | push] | M1(t) x E0(t)
    // FULL: tiles t .. t + 4 lie in full groups (no ragged-group tests, the short prefetch address)
    auto tile_step = [&](int t, auto u_tag, bool with_prev, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr int U = decltype(u_tag)::value;                      // t & 3
        if (!(PLSLAM_MI_X & 1)) __syncthreads();   // tile t expanded; every wave is past its reads of the other buffer
        const uint8_t* bt = btile + (U & 1) * MI_TILE_BYTES + c * MI_ROW_STRIDE + 16 * g;
        i32x4 bfr[MH_KSTEPS];
#if PLSLAM_MI_SWZ
        // (the lane's four chunk addresses of buffer 0 are loop-invariant registers; the buffer is an immediate offset)
#define PLSLAM_MI_READ_B(KS) (*reinterpret_cast<const i32x4*>(btile + bswz[KS] + (U & 1) * MI_TILE_BYTES))
        (void)bt;
#else
#define PLSLAM_MI_READ_B(KS) ((PLSLAM_MI_X & 2) ? i32x4{(int)MI_MAG + t, (int)MI_MAG, (int)MI_MAG + (KS), (int)MI_MAG} : *reinterpret_cast<const i32x4*>(bt + 32 * (KS)))
#endif
        bfr[0] = PLSLAM_MI_READ_B(0);
        bfr[1] = PLSLAM_MI_READ_B(1);
        // ragged group: lanes whose class has run out of columns take the penalty from this tile on (K1h)
        if (!FULL && t >= nfull && ((t & 15) == 0 || (t & 15) == lim_part)) {
            const uint32_t pen = lane_lim() == (t & 15) ? MI_COL_PENALTY : 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) seed[r] += pen;
            asm volatile("" : "+v"(seed));
        }
        const f32x16 cseed = __builtin_bit_cast(f32x16, seed);
        f32x16 m0;
        uint32_t cma = MI_NONE32, cmb = MI_NONE32;
        float cfa = 0.0f;              // (PLSLAM_MI_NOPACK) the column direction's float minimum of the accumulator set in hand
        auto cf_join = [&]() __attribute__((always_inline)) -> uint32_t { return __builtin_bit_cast(uint32_t, cfa); };
        constexpr int PAR1 = (U + 1) & 1, PAR0 = U & 1;     // the parities of tile t-1 (phase 1) and of tile t (phase 2)
        __builtin_amdgcn_sched_barrier(0);
        // phase 1: M-tile 0 of tile t under the bookkeeping of M-tile 1 of tile t-1
        // (PLSLAM_MI_NOPACK: the accumulators of M-tile 1 are inline-asm operands -- the wait states behind the chain's last
        // MFMA, issued one bookkeeping block before the end of the step before, are counted by hand: 14 as the compiler counts
        // them for its own instructions (round 5's listing: `s_nop 5` here, in front of the v_perm).  They pass under the latency
        // of the operand reads just requested, which the first MFMA waits for anyway.)
        if (PLSLAM_MI_NOPACK && !(PLSLAM_MI_X & (8 | 128))) {
            if (DIRECTED) asm volatile("s_nop 7"); else asm volatile("s_nop 4");
            __builtin_amdgcn_sched_barrier(0);
        }
        PLSLAM_MI_EPI2(m1, 1, 0, PAR1) PLSLAM_MI_MMA(m0, 0, 0, cseed)
        bfr[2] = PLSLAM_MI_READ_B(2);
        PLSLAM_MI_EPI2(m1, 1, 2, PAR1) PLSLAM_MI_MMA(m0, 0, 1, m0)
        bfr[3] = PLSLAM_MI_READ_B(3);
        PLSLAM_MI_EPI2(m1, 1, 4, PAR1) PLSLAM_MI_MMA(m0, 0, 2, m0)
        // the next tile's raw dword (requested three steps ago) leaves the ring: an LDS latency ahead of its expansion
        const uint32_t raw_next = (PLSLAM_MI_X & 16) ? 0u : take_raw(RING == 4 ? 1024u * ((U + 1) & 3) : ring_slot);
        PLSLAM_MI_EPI2(m1, 1, 6, PAR1) PLSLAM_MI_MMA(m0, 0, 3, m0)
        const uint32_t cm1 = PLSLAM_MI_NOPACK ? cf_join() : PLSLAM_MI_F16 ? cma : pk_min16(cma, cmb);
        // behind the chain of M-tile 0: the expansion of the next tile (its buffer was read for the last time before this
        // step's barrier) and the prefetch -- independent work while the last MFMA of the chain completes
        if (!(PLSLAM_MI_X & 16)) {
        expand_store(raw_next, (U + 1) & 1, t + 1, FULL);         // past the last tile: a harmless rewrite of the idle buffer
        // three tiles ahead of its use -- three slots: into the slot just read; four: into this tile's own (read one step ago)
        if (FULL) {
            load_raw_async_full(t + 4, RING == 4 ? 1024u * U : ring_slot);
            // the next step's tile: the next row of the group's classes, or -- only behind a step with U = 0, the lean chunks
            // being four tiles long -- the first tile of the next group (16 x 32 rows further on)
            if (PLSLAM_MI_R5 & 16) pf32 += (U == 3 && ((t + 5) & 15) == 0) ? (MH_GROUP_ROWS - 15) * 32u : 32u;
        } else {
            load_raw_async(t + 4, RING == 4 ? 1024u * U : ring_slot);
        }
        } else asm volatile("" :: "v"(raw_next));
        ring_slot = ring_slot == 2048u ? 0u : ring_slot + 1024u;  // (scalar)
        if (with_prev) {
            // block (t - 9) / 8 of column results: its last tile was parked in the step before this one, by every wave before
            // this step's barrier; tile t - 1 is about to take the block's first slot: a second barrier (workgroup-uniform
            // branch).  (The blocks of the window before were finished behind its loop.)
            if (!DIRECTED && U == 1 && ((t - 1) & (MH_CGROUP - 1)) == 0 && t - 9 >= wt0) {
                combine_columns((t - 9) >> 3);
                __syncthreads();
            }
            finish_columns(t - 1, cm0_prev, cm1);
            // wave-uniform: tile t-1 closed a row group (both M-tiles of it are in the minima now)
            if (U == 0 && ((t - 1) & (MH_GROUP - 1)) == MH_GROUP - 1) push_groups(t - 1, std::false_type{}, nullptr);
        }
        __builtin_amdgcn_sched_barrier(0);
        // phase 2: M-tile 1 of tile t under the bookkeeping of M-tile 0 of tile t
        cma = MI_NONE32; cmb = MI_NONE32;
#if PLSLAM_MI_UNSCALED
        // The chain's first MFMA through inline asm with an early-clobber destination: for this loop-carried accumulator set
        // the compiler picks the tied form (destination = C operand) and copies the 16 seed registers into it every tile (8
        // v_mov_b64).  What the compiler cannot see is harmless by construction: the result is read next by the chain's second
        // MFMA (a builtin, six VALU instructions later: any MFMA -> MFMA wait state is long over) and by VALU instructions only
        // behind the chain's last MFMA, a builtin whose hazards the compiler tracks; the sources are VGPRs the compiler waits
        // for as for any asm operand.
        if (!(PLSLAM_MI_X & 4)) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %3 cbsz:4 blgp:4" : "=&v"(m1) : "v"(afrag[1][0]), "v"(bfr[0]), "v"(cseed));
        else { m1 = cseed; m1[0] = __builtin_bit_cast(float, bfr[0].x ^ afrag[1][0].x); asm volatile("" : "+v"(m1)); }
        __builtin_amdgcn_sched_barrier(0);
#else
        PLSLAM_MI_MMA(m1, 1, 0, cseed)
#endif
        PLSLAM_MI_EPI2(m0, 0, 0, PAR0)
        PLSLAM_MI_MMA(m1, 1, 1, m1)    PLSLAM_MI_EPI2(m0, 0, 2, PAR0)
        PLSLAM_MI_MMA(m1, 1, 2, m1)    PLSLAM_MI_EPI2(m0, 0, 4, PAR0)
        PLSLAM_MI_MMA(m1, 1, 3, m1)    PLSLAM_MI_EPI2(m0, 0, 6, PAR0)
        cm0_prev = PLSLAM_MI_NOPACK ? cf_join() : PLSLAM_MI_F16 ? cma : pk_min16(cma, cmb);
    };
    // the bookkeeping of M-tile 1 of a window's last tile on its own (no following step to hide under)
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        uint32_t cma = MI_NONE32, cmb = MI_NONE32;
#if PLSLAM_MI_F16
        float cfa = 0.0f;
        if (PLSLAM_MI_NOPACK) {
            // (the accumulators of M-tile 1 are inline-asm operands: the wait states behind the chain's last MFMA -- four VALU
            // instructions back in the last step -- are counted by hand here: 12 for the 8-pass form)
            asm volatile("s_nop 7\n\ts_nop 7");
            __builtin_amdgcn_sched_barrier(0);
            PLSLAM_MI_EPI2(m1, 1, 0, 0) PLSLAM_MI_EPI2(m1, 1, 2, 0) PLSLAM_MI_EPI2(m1, 1, 4, 0) PLSLAM_MI_EPI2(m1, 1, 6, 0)
            finish_columns(t, cm0_prev, __builtin_bit_cast(uint32_t, cfa));
            return;
        }
        if (t & 1) {                               // (wave-uniform) an odd last tile: the even tile's pairs of M-tile 1 are waiting
            PLSLAM_MI_EPI2(m1, 1, 0, 1) PLSLAM_MI_EPI2(m1, 1, 2, 1) PLSLAM_MI_EPI2(m1, 1, 4, 1) PLSLAM_MI_EPI2(m1, 1, 6, 1)
        } else {
            // an even last tile: its pairs go into the minima directly -- M-tile 1's now, M-tile 0's were kept in the last step
            PLSLAM_MI_EPI2(m1, 1, 0, 0) PLSLAM_MI_EPI2(m1, 1, 2, 0) PLSLAM_MI_EPI2(m1, 1, 4, 0) PLSLAM_MI_EPI2(m1, 1, 6, 0)
#pragma unroll
            for (int s = 8; s < 16; ++s) if (PLSLAM_MI_LOOK(1)) gm[s] = pk_min16(gm[s], kp[s]);      // (kept a moment ago)
            // (M-tile 1's were folded just now -- or never kept)
#pragma unroll
            for (int s = 0; s < 8; ++s) if (PLSLAM_MI_LOOK(0)) gm[s] = pk_min16(gm[s], kp[s]);
        }
        (void)cmb;
        finish_columns(t, cm0_prev, cma);
#else
        PLSLAM_MI_EPI(m1, 1, 0, cma) PLSLAM_MI_EPI(m1, 1, 1, cmb) PLSLAM_MI_EPI(m1, 1, 2, cma) PLSLAM_MI_EPI(m1, 1, 3, cmb)
        PLSLAM_MI_EPI(m1, 1, 4, cma) PLSLAM_MI_EPI(m1, 1, 5, cmb) PLSLAM_MI_EPI(m1, 1, 6, cma) PLSLAM_MI_EPI(m1, 1, 7, cmb)
        finish_columns(t, cm0_prev, pk_min16(cma, cmb));
#endif
    };
    auto pipeline = [&](u32x2_t* rb) __attribute__((always_inline)) {
        // (wt0 is a multiple of 64: t & 3 of the unrolled steps is static.  The first step has no previous tile: its
        // phase 1 runs on "none" accumulators, its column / group actions are skipped)
        {
            // (written HERE by sixteen moves the compiler can neither hoist out of the window loop nor fold: as a loop-invariant
            // vector it was kept alive across the tile loops -- at a 152-register budget spilled in the prologue and reloaded at
            // every window start behind an s_waitcnt vmcnt(0), i.e. behind the whole prefetch)
            u32x16 none;
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(none[r]) : "s"(MI_NONE32));     // (the low halves are what the bookkeeping takes)
            m1 = __builtin_bit_cast(f32x16, none);
#if PLSLAM_MI_F16
            // the first step folds "the tile before the window" (odd) with whatever is kept: nothing
#pragma unroll
            for (int s = 8; s < 16; ++s) if (PLSLAM_MI_LOOK(1)) kp[s] = MI_NONE32;
#endif
        }
        int tb = wt0;
        {
            // whole chunks of four tiles whose prefetches (four tiles ahead) stay inside full groups: the lean instantiation
            const int lim = (wt1 < nfull ? wt1 : nfull) - 7;          // tb + 3 + 4 < nfull and tb + 3 < wt1
            pf32 = (((uint32_t)((tb + 4) & ~15) << 5) + (uint32_t)((tb + 4) & 15)) * 32u;      // (the first lean step's prefetch: tile tb + 4)
            for (; tb < lim; tb += 4) {
                tile_step(tb, std::integral_constant<int, 0>{}, tb != wt0, std::true_type{});
                tile_step(tb + 1, std::integral_constant<int, 1>{}, true, std::true_type{});
                tile_step(tb + 2, std::integral_constant<int, 2>{}, true, std::true_type{});
                tile_step(tb + 3, std::integral_constant<int, 3>{}, true, std::true_type{});
            }
        }
        for (; tb < wt1; tb += 4) {
            tile_step(tb, std::integral_constant<int, 0>{}, tb != wt0, std::false_type{});
            if (tb + 1 < wt1) tile_step(tb + 1, std::integral_constant<int, 1>{}, true, std::false_type{});
            if (tb + 2 < wt1) tile_step(tb + 2, std::integral_constant<int, 2>{}, true, std::false_type{});
            if (tb + 3 < wt1) tile_step(tb + 3, std::integral_constant<int, 3>{}, true, std::false_type{});
        }
        // the window's last tile opens a block of columns while the block before it still waits in the slots (the step that
        // would have combined it does not exist): combine it now
        if (!DIRECTED && ((wt1 - 1) & (MH_CGROUP - 1)) == 0 && wt1 - 9 >= wt0) {
            __syncthreads();
            combine_columns((wt1 - 9) >> 3);
            __syncthreads();
        }
        epilogue(wt1 - 1);
        push_groups(wt1 - 1, std::true_type{}, rb);   // the (possibly partial) last row group: merged into registers
    };
#undef PLSLAM_MI_EPI2
#undef PLSLAM_MI_EPI
#undef PLSLAM_MI_MMA
#undef PLSLAM_MI_READ_B

    // Row results of a window: K1h's finish_rows with this kernel's slot -> row map (slot 8 mt + q: low half = local row
    // 32 mt + 16 g + q, high half = local row 32 mt + 16 g + q + 8 of the wave's 64).  One lane per row after the transpose;
    // the best entry names the CELL (group, class) that holds the best column, whose S members are recomputed from the raw rows.
    // rb: the parked sorted pairs with the window's last group merged in (push_groups, FINAL) -- in registers.
    auto finish_rows = [&](const u32x2_t* rb) __attribute__((always_inline)) {
        if (PLSLAM_MI_X & 128) {       // the lane-local minima leave as they are: one store per lane
            const int row_ = iw + lane + (lane & 32) * 3;
            if (row_ < n1) ((gu2_t) reinterpret_cast<u32x2_t*>(sd.keys12))[row_] =
                u32x2_t{__builtin_bit_cast(uint32_t, fminf(f6[0], f6[1])), __builtin_bit_cast(uint32_t, fminf(f6[2], f6[3])) + rb[0].x};
            return;
        }
        uint32_t* rowx = reinterpret_cast<uint32_t*>(smem) + w * (64 * ROWX_STRIDE);
        // (the barrier behind the tile loop stands between every wave's last read of the b tile / of its parked pairs and
        // these writes: the transpose may overwrite both)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int lrow = 32 * (s >> 3) + 16 * g + (s & 7);
            rowx[lrow * ROWX_STRIDE + c] = __builtin_amdgcn_perm(rb[s].y, rb[s].x, 0x05040100u);          // (x.lo | y.lo << 16)
            rowx[(lrow + 8) * ROWX_STRIDE + c] = __builtin_amdgcn_perm(rb[s].y, rb[s].x, 0x07060302u);    // (x.hi | y.hi << 16)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
        const uint32_t* mine = rowx + lane * ROWX_STRIDE;
#if PLSLAM_MI_R5 & 1
        // A class's word is its sorted pair (first | second << 16).  The smallest of the 64 entries is the smallest FIRST entry
        // (class c*); the second smallest is the second smallest first entry or c*'s second entry -- every other second entry
        // is no smaller than its own class's first.  (entry << 16 | class: distinct words, so the order is strict)
#pragma unroll
        for (int cls = 0; cls < 32; ++cls) {
            uint32_t x;
            asm("v_lshl_or_b32 %0, %1, 16, %2" : "=v"(x) : "v"(mine[cls]), "n"(cls));
            asm("v_med3_u32 %0, %1, %2, %0" : "+v"(k1) : "v"(k0), "v"(x));         // k0 <= k1: the middle one is the new second
            k0 = umin_(k0, x);
        }
        {
            const uint32_t cs = k0 & 31u;
            k1 = umin_(k1, (mine[cs] & 0xFFFF0000u) | cs);
        }
#else
#pragma unroll 8
        for (int cls = 0; cls < 32; ++cls) {
            const uint32_t e = mine[cls];
            merge2(k0, k1, (e << 16) | (uint32_t)cls, (e & 0xFFFF0000u) | (uint32_t)cls);
        }
#endif
        int row = iw + lane + (lane & 32) * 3;                // lanes 32..63: M-tile 1's rows, 128 further on
        // (opaque: the row's addresses -- its keys, its descriptor -- are loop-invariant over the windows, and hoisted in front
        // of the tile loops they live in registers the loops need: the compiler spilled them, 1 KB of scratch per wave)
        asm volatile("" : "+v"(row));
        if (row < n1) {
            const gu2_t out = (gu2_t) reinterpret_cast<u32x2_t*>(sd.keys12) + row;
            const gcu32x4_t ap = (gcu32x4_t)(araw + (size_t)row * 8);
            const u32x4_t a_lo = ap[0], a_hi = ap[1];
            // (key16 << 16 | class) -> first row of the cell (group, class), its stride count, the distance
            auto group_of = [&](uint32_t k, uint32_t& jbase, uint32_t& cnt) {
                const uint32_t t0 = (uint32_t)wt0 + (((k >> (PLSLAM_MI_UNSCALED ? 16 : 21)) & 3u) << 4);   // first tile of the group
                const uint32_t s = t0 < (uint32_t)nfull ? (uint32_t)MH_GROUP : (uint32_t)rag_s;
                jbase = (t0 >> 4) * MH_GROUP_ROWS + s * (k & 0xFFFFu);
                cnt = s;
            };
            // all members of a cell: the smallest (d << 23 | j) and the second smallest (MI_RESCAN_BATCH candidates' rows
            // are requested together: the accumulators are dead here, and the loop is a chain of L2 round trips)
            auto rescan = [&](uint32_t jbase, uint32_t cnt, uint32_t& best, uint32_t& second) {
                best = second = KEY_NONE;
                const uint32_t left = (uint32_t)n2 - jbase;                       // >= 1: the class's first column exists
                const uint32_t nvalid = cnt < left ? cnt : left;
                const PLSLAM_GLOBAL char* const rbp = bbytes + (size_t)jbase * 32;
#pragma unroll
                for (int k0_ = 0; k0_ < MH_GROUP; k0_ += MI_RESCAN_BATCH) {
                    u32x4_t bl[MI_RESCAN_BATCH], bh[MI_RESCAN_BATCH];
#pragma unroll
                    for (int q = 0; q < MI_RESCAN_BATCH; ++q) {
                        const uint32_t kk = (uint32_t)(k0_ + q) < nvalid ? (uint32_t)(k0_ + q) : nvalid - 1u;   // past the end: a duplicate, masked below
                        const gcu32x4_t bp = (gcu32x4_t)(rbp + kk * 32u);
                        bl[q] = bp[0];
                        bh[q] = bp[1];
                    }
#pragma unroll
                    for (int q = 0; q < MI_RESCAN_BATCH; ++q) {
                        const uint32_t d = hamming256(a_lo, a_hi, bl[q], bh[q]);
                        const uint32_t cand = (uint32_t)(k0_ + q) < nvalid ? ((d << KEY_IDX_BITS) | (jbase + (uint32_t)(k0_ + q))) : KEY_NONE;
                        second = umin_(second, umax_(best, cand));
                        best = umin_(best, cand);
                    }
                }
            };
#if PLSLAM_MI_R5 & 2
            // The winner cell on the fast road: SIXTEEN consecutive columns from one base with immediate offsets and no validity
            // tests -- the cell's own (16, or rag_s in the ragged group) plus, behind a cell of the ragged group, the first
            // columns of the next class, or, when the end of b cuts the window, the columns just in front of the cell
            // (base = n2 - 16).  The extra columns are REAL columns of other cells, and harmless: (i) the winner cell holds the
            // row's smallest (d, j) -- it is the smallest (d, group, class), and cells are consecutive ranges of j --, so no
            // extra column beats or ties-and-precedes its best member; (ii) an extra column that becomes the "second" of these
            // sixteen has the true distance of a column outside the cell, which is no smaller than o1's, the exact minimum over
            // all other cells: min(second, o1) is unchanged as a distance (and as an index wherever the index is exact).  Not
            // when the base would leave the window (its columns were already counted by the window before) or b has fewer
            // than 16 rows: such lanes -- none at the shipped sizes -- send their wave down the guarded road.
            auto rescan16 = [&](uint32_t jb, uint32_t& best, uint32_t& second) {
                uint32_t bk = 0xFFFFFFFFu, sk = 0xFFFFFFFFu;                     // (d << 4 | k)
                const PLSLAM_GLOBAL char* const rbp = bbytes + (size_t)jb * 32;
                constexpr int RB = MI_RESCAN_BATCH16;
#pragma unroll
                for (int k0_ = 0; k0_ < 16; k0_ += RB) {
                    u32x4_t bl[RB], bh[RB];
#pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        const gcu32x4_t bp = (gcu32x4_t)(rbp + (k0_ + q) * 32);
                        if (PLSLAM_MI_X & 32) {              // (experiment: no gathers -- the rows of a stand in)
                            bl[q] = a_hi + (uint32_t)(k0_ + q);
                            bh[q] = a_lo;
                            asm volatile("" :: "v"(bp));
                        } else {
                            bl[q] = bp[0];
                            bh[q] = bp[1];
                        }
                    }
#pragma unroll
                    for (int q = 0; q < RB; ++q) {
                        const uint32_t d = hamming256(a_lo, a_hi, bl[q], bh[q]);
                        uint32_t x;
                        asm("v_lshl_or_b32 %0, %1, 4, %2" : "=v"(x) : "v"(d), "n"(k0_ + q));
                        asm("v_med3_u32 %0, %1, %2, %0" : "+v"(sk) : "v"(bk), "v"(x));
                        bk = umin_(bk, x);
                    }
                }
                best = ((bk >> 4) << KEY_IDX_BITS) | (jb + (bk & 15u));
                second = ((sk >> 4) << KEY_IDX_BITS) | (jb + (sk & 15u));
            };
#endif
            uint32_t r0 = KEY_NONE, r1 = KEY_NONE;
            if ((k0 >> 16) <= MI_KEY16_MAX) {
                uint32_t jb, cnt, in2;
                group_of(k0, jb, cnt);
#if PLSLAM_MI_R5 & 2
                const uint32_t jsh = jb + 16u <= (uint32_t)n2 ? jb : (uint32_t)n2 - 16u;          // (wraps when n2 < 16: caught below)
                const bool guarded = n2 < 16 || jsh < (uint32_t)wt0 * MH_TILE_N;
                if (PLSLAM_MI_X & 64) { r0 = jsh; in2 = jb; }
                else if (__builtin_amdgcn_ballot_w64(guarded) == 0) rescan16(jsh, r0, in2);
                else rescan(jb, cnt, r0, in2);
#else
                rescan(jb, cnt, r0, in2);
#endif
                if ((k1 >> 16) <= MI_KEY16_MAX) {
                    // the best key outside the winner's cell: its distance is exact, its column is the first of
                    // its cell unless the exact index was asked for and it IS the second best
                    uint32_t jb1, cnt1;
                    group_of(k1, jb1, cnt1);
                    uint32_t o1 = ((k1 >> (16 + MI_DSHIFT)) << KEY_IDX_BITS) | jb1;
                    if ((sd.flags & 1) && (o1 >> KEY_IDX_BITS) <= (in2 >> KEY_IDX_BITS)) {
                        uint32_t b1, s1;
                        rescan(jb1, cnt1, b1, s1);
                        o1 = b1;
                    }
                    r1 = umin_(in2, o1);
                } else {
                    r1 = in2;
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

    // the first window's first four tiles of b are requested BEFORE the rows of a: one memory latency for both (a 200 x 200
    // problem is seven tiles long: its workgroup's time is mostly such latencies)
    uint32_t raw_first = load_raw(0);
    load_raw_async(1, 1024);
    load_raw_async(2, 2048);
    load_raw_async(3, RING == 4 ? 3072 : 0);
    // ---- A operands: MFMA row c of M-tile mt = block row 128 mt + 32 w + 16 g' + r' (K1h's mh_block_row): a lane's 16
    // accumulator registers of an M-tile are 16 CONSECUTIVE rows of a ----
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int row = bd.row0 + 128 * mt + 32 * w + 16 * ((c >> 2) & 1) + (c & 3) + 4 * (c >> 3);
        const int rrow = row < n1 ? row : n1 - 1;
        const gcu32_t p = araw + (size_t)rrow * 8 + g;
#pragma unroll
        for (int ks = 0; ks < MH_KSTEPS; ++ks) afrag[mt][ks] = expand_dword_fp4<true, MI_MAG>(p[2 * ks]);
    }

#ifdef PLSLAM_MI_PROF
    const unsigned long long prof_t1 = PLSLAM_MI_TICK();
#endif
    for (;;) {
#ifdef PLSLAM_MI_PROF
        prof_tl = PLSLAM_MI_TICK();
#endif
#if !(PLSLAM_MI_UNSCALED && (PLSLAM_MI_R5 & 4))                 // (else the window's first push writes the parked pairs)
#pragma unroll
        for (int s = 0; s < 16; ++s) park[s * 64] = u32x2_t{0xFFFFFFFFu, 0xFFFFFFFFu};   // wave-private: no barrier needed
#endif
        expand_store(raw_first, 0, wt0);           // wt0 is a multiple of 128: buffer parity restarts at 0
        ring_slot = 1024;                          // the slot of tile wt0 + 1
        u32x2_t rb[16];                            // the window's parked pairs, its last group merged in
        // the tile loops issue ahead of the other waves' per-item phases (prologue, row finish: long VALU and gather stretches
        // with no matrix instruction in them): measured 2.43-2.46 against 2.49-2.51 ms and 2.52 against 2.58 ms on two boxes;
        // the other way round (the row finish first, so that the workgroup's slot comes free sooner) and lowering the loops'
        // own per-item steps (group pushes, column combine) measured no better than no priorities at all
        // (profiles/r5_c_scan_wave_priorities.txt)
        if (PLSLAM_MI_PRIO) __builtin_amdgcn_s_setprio(1);
        pipeline(rb);
        if (PLSLAM_MI_PRIO) __builtin_amdgcn_s_setprio(0);
#ifdef PLSLAM_MI_PROF
        { const unsigned long long t = PLSLAM_MI_TICK(); prof_loop += t - prof_tl; prof_tl = t; }
#endif
        __syncthreads();                           // every wave is past its last operand read of the b tile, every column minimum is parked
        // the window's last block of column results (full or partial)
        combine_columns((wt1 - 1) >> 3);
        finish_rows(rb);
#ifdef PLSLAM_MI_PROF
        prof_fin += PLSLAM_MI_TICK() - prof_tl;
#endif
        if (wt1 == ntiles) break;
        __syncthreads();                           // smem becomes the b tile (+ parking area) again
        wt0 = wt1;
        wt1 = ntiles < wt0 + MH_WINDOW ? ntiles : wt0 + MH_WINDOW;
        raw_first = load_raw(wt0);
        load_raw_async(wt0 + 1, 1024);
        load_raw_async(wt0 + 2, 2048);
        load_raw_async(wt0 + 3, RING == 4 ? 3072 : 0);
    }
#if PLSLAM_MI_TAIL
    if (!DIRECTED && tail_counts) {
        __shared__ int s_last;
#if PLSLAM_MI_TAIL == 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the write-through stores of the partials have been acknowledged
#else
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // this workgroup's partials (and row results) leave its XCD's L2
#endif
        __syncthreads();
        const int nwb = (n1 + 255) >> 8;                            // the problem's workgroups: one per 256 rows of a
        if (tid == 0) s_last = atomicAdd(&tail_counts[bd.item], 1) == nwb - 1;
        __syncthreads();
        if (s_last) {
#if PLSLAM_MI_TAIL != 2
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // the other workgroups' partials are read from memory
#endif
            const gcu32_t part = (gcu32_t) sd.part21;
            auto wide = [](uint32_t k17, uint32_t wb) -> uint32_t { return ((k17 >> 8) << KEY_IDX_BITS) | ((k17 & 255u) + 256u * wb); };
            // (every load of a lane's slots goes out before the first is used: the loads bypass this XCD's L2, and a chain of them
            // keeps the workgroup's slot for microseconds per link -- measured with one slot at a time: scan + 0.18 ms per step)
            constexpr int TQ = 8, TWB = 8;                          // slots per lane and row blocks per round: 2048 columns, 2048 rows
            const int nslots = MH_TILE_N * ntiles;
            for (int s0_ = 0; s0_ < nslots; s0_ += 256 * TQ) {
                uint32_t b0[TQ], b1[TQ], sx[TQ];
#pragma unroll
                for (int q = 0; q < TQ; ++q) { b0[q] = b1[q] = KEY_NONE; sx[q] = 0xFFFFFFFFu; }
                for (int wb0 = 0; wb0 < nwb; wb0 += TWB) {
                    uint32_t e[TWB][TQ];
#pragma unroll
                    for (int u = 0; u < TWB; ++u)
#pragma unroll
                        for (int q = 0; q < TQ; ++q) {
                            const int slot = s0_ + tid + 256 * q;
                            const bool on = wb0 + u < nwb && slot < nslots;
#if PLSLAM_MI_TAIL == 2      // (an agent-scope load: past this XCD's L2)
                            e[u][q] = on ? __hip_atomic_load((const uint32_t*)(part + ((size_t)(wb0 + u) * n2p + slot)), __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT)
                                         : 0xFFFFFFFFu;
#else
                            e[u][q] = on ? part[(size_t)(wb0 + u) * n2p + slot] : 0xFFFFFFFFu;
#endif
                        }
#pragma unroll
                    for (int u = 0; u < TWB; ++u)
#pragma unroll
                        for (int q = 0; q < TQ; ++q) {
                            if (wb0 + u < nwb) {                    // (uniform)
                                const uint32_t k = wide(e[u][q] >> 9, (uint32_t)(wb0 + u)), e1 = ((e[u][q] & 511u) << 8) | 255u;
                                sx[q] = k < b0[q] ? e1 : sx[q];
                                b1[q] = umin_(b1[q], umax_(b0[q], k));
                                b0[q] = umin_(b0[q], k);
                            }
                        }
                }
#pragma unroll
                for (int q = 0; q < TQ; ++q) {
                    const int slot = s0_ + tid + 256 * q;
                    if (slot >= nslots) continue;
                    const int j = L.row_of(slot >> 5, slot & 31);
                    if (j >= n2) continue;
                    uint32_t r0 = b0[q], r1 = b1[q];
                    if (r0 < (257u << KEY_IDX_BITS)) {
                        r1 = umin_(r1, ((sx[q] >> 8) << KEY_IDX_BITS) | KEY_IDX_MASK);
                        if (r1 >= (257u << KEY_IDX_BITS)) r1 = KEY_NONE;
                    } else {
                        r0 = r1 = KEY_NONE;
                    }
                    ((gu2_t) reinterpret_cast<u32x2_t*>(sd.keys21))[j] = u32x2_t{r0, r1};
                }
            }
            if (tid == 0) tail_counts[bd.item] = 0;                 // (the next run of the plan counts afresh)
        }
    }
#endif
#if PLSLAM_MI_PERSIST
    }
#endif
#ifdef PLSLAM_MI_PROF
    if (threadIdx.x == 0 && blockIdx.x < MI_PROF_WGS) {
        g_mi_prof[4 * blockIdx.x + 0] = prof_t1 - prof_t0;
        g_mi_prof[4 * blockIdx.x + 1] = prof_loop;
        g_mi_prof[4 * blockIdx.x + 2] = prof_fin;
        g_mi_prof[4 * blockIdx.x + 3] = ((unsigned long long)ntiles << 40) | (PLSLAM_MI_TICK() - prof_t0);
    }
#endif
}

#ifdef PLSLAM_MI_PROF
extern "C" int plslam_debug_k1i_profile(unsigned long long* out, int nwg)
{
    if (nwg > MI_PROF_WGS) nwg = MI_PROF_WGS;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mi_prof), sizeof(unsigned long long) * 4 * (size_t)nwg) == hipSuccess ? nwg : -1;
}
#endif

bool k1i_tail_built() { return PLSLAM_MI_TAIL != 0; }

int launch_scan_sym_mfma_i(const SymDesc* d_sym, const BlockDesc* d_blocks, int nblocks, int32_t* d_zero, int nzero,
                           bool directed, hipStream_t s, int32_t* tail_counts)
{
    if (nblocks <= 0) return PLSLAM_OK;
    int grid = nblocks;
#if PLSLAM_MI_PERSIST
    if (grid > PLSLAM_MI_PERSIST) grid = PLSLAM_MI_PERSIST & ~7;
#endif
#if PLSLAM_MI_TAIL
    if (directed) hipLaunchKernelGGL((k_scan_sym_mfma_i<true>), dim3(grid), dim3(256), 0, s, d_sym, d_blocks, d_zero, nzero, nblocks, (int32_t*)nullptr);
    else hipLaunchKernelGGL((k_scan_sym_mfma_i<false>), dim3(grid), dim3(256), 0, s, d_sym, d_blocks, d_zero, nzero, nblocks, tail_counts);
#else
    (void)tail_counts;
    if (directed) hipLaunchKernelGGL((k_scan_sym_mfma_i<true>), dim3(grid), dim3(256), 0, s, d_sym, d_blocks, d_zero, nzero, nblocks);
    else hipLaunchKernelGGL((k_scan_sym_mfma_i<false>), dim3(grid), dim3(256), 0, s, d_sym, d_blocks, d_zero, nzero, nblocks);
#endif
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

}  // namespace plslam
