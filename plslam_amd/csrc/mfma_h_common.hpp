// mfma_h_common.hpp -- typedefs, constants and device helpers shared by the matrix-core scans with minimum-only bookkeeping
// (K1h, hamming_mfma_h.hip, and K1i, hamming_mfma_i.hip): fp4 operand codes, packed 16-bit key operations, the raw-row
// Hamming distance of the second-best recomputation.  Internal to libplslam_hip.so.
#pragma once

#include "common.hpp"

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
typedef const PLSLAM_GLOBAL u32x2_t* gu2c_t;
typedef PLSLAM_GLOBAL uint32_t* gu32_t;

namespace {

constexpr int MH_TILE_N = 32;                 // b rows per tile
constexpr int MH_KSTEPS = 4;                  // 256 bits = 4 x K 64
constexpr int MH_ROW_STRIDE = 144;            // bytes per expanded b row in LDS (128 + 16: 4-bank skew)
constexpr int MH_TILE_BYTES = MH_TILE_N * MH_ROW_STRIDE;
constexpr int MH_GROUP = 16;                  // tiles per group of the row direction (512 b rows)
constexpr int MH_GROUP_ROWS = MH_GROUP * MH_TILE_N;
constexpr int MH_WINDOW = 64;                 // tiles per window: the row keys' tag holds the group within the window (2 bits)
#ifndef PLSLAM_MERGE_SPT
#define PLSLAM_MERGE_SPT 4
#endif
constexpr int MH_MERGE_SPT = PLSLAM_MERGE_SPT;   // slots per lane of the merge kernel (PARTS == 1)
constexpr int MH_CGROUP = 8;                  // tiles whose column results are staged in LDS and stored together (256 slots)
#ifndef PLSLAM_NT_STREAMS
#define PLSLAM_NT_STREAMS 1
#endif
constexpr uint32_t FP4_NEG = 0x88888888u;
constexpr uint32_t FP4_ONE = 0x22222222u;
constexpr uint32_t FP4_FOUR = 0x66666666u;    // e2m1 code 0b0110 = 4.0
constexpr int SCALE_A = 133, SCALE_B = 127;   // E8M0: 2^6 on the a side, 2^0 on the b side
constexpr uint32_t ACC_BITS = 0x4B000000u + 16384u;   // float bits of 2^23 + 16384
constexpr uint32_t KEY16_MAX = 0x807Fu;       // 16-bit keys are (d << 7) | tag7; anything above is "none"
// a column that does not exist: zero codes (the contraction contributes nothing: "distance 128") + this in the seed
// = key 0xBF80 + tag: above KEY16_MAX, below the 16-bit wrap
constexpr uint32_t COL_PENALTY = 0x7F80u;

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
__device__ __forceinline__ void pk_push2(uint32_t& b0, uint32_t& b1, uint32_t key)
{
    b1 = pk_min16(b1, pk_max16(b0, key));
    b0 = pk_min16(b0, key);
}
// accumulators of the two M-tiles side by side: hi.lo16 << 16 | lo.lo16.  The BUILTIN, never inline asm: this is the one
// instruction that reads MFMA results ("K1e determinism")
__device__ __forceinline__ uint32_t pack_acc(float lo, float hi)
{
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x05040100u);
}
// 32 bits of a descriptor -> 32 fp4 codes of s(bit): dword s holds bits 4k + s, nibble k = 0x2 | bit << 3 (see K1f)
// MAG: the magnitude code in every nibble (FP4_ONE = 1.0; K1i's unscaled form: FP4_FOUR = 4.0)
template <bool A_SIDE, uint32_t MAG = FP4_ONE>
__device__ __forceinline__ i32x4 expand_dword_fp4(uint32_t x)
{
    uint32_t x1, x2, x3;
    asm("v_add_u32 %0, %1, %1" : "=v"(x1) : "v"(x));
    asm("v_add_u32 %0, %1, %1" : "=v"(x2) : "v"(x1));
    asm("v_add_u32 %0, %1, %1" : "=v"(x3) : "v"(x2));
    constexpr uint32_t base = A_SIDE ? (MAG ^ FP4_NEG) : MAG;               // a side: sign nibble-bit flipped
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
__device__ __forceinline__ uint32_t hamming256(u32x4_t a_lo, u32x4_t a_hi, u32x4_t b_lo, u32x4_t b_hi)
{
    uint32_t d = bcnt_acc_(a_lo.x ^ b_lo.x, 0u);
    d = bcnt_acc_(a_lo.y ^ b_lo.y, d);
    d = bcnt_acc_(a_lo.z ^ b_lo.z, d);
    d = bcnt_acc_(a_lo.w ^ b_lo.w, d);
    d = bcnt_acc_(a_hi.x ^ b_hi.x, d);
    d = bcnt_acc_(a_hi.y ^ b_hi.y, d);
    d = bcnt_acc_(a_hi.z ^ b_hi.z, d);
    d = bcnt_acc_(a_hi.w ^ b_hi.w, d);
    return d;
}
__device__ __forceinline__ int xcd_remap_(int orig, int nwg) { return (orig & 7) * (nwg >> 3) + (orig >> 3); }

}  // namespace

}  // namespace plslam
