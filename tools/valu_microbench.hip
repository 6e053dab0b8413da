// valu_microbench.hip -- measures the issue rate of the integer VALU / cross-lane / LDS instructions
// the Hamming scan is built from, on the device it runs on.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_microbench.hip -o build/valu_microbench
// Output: one line per instruction: cycles per wave64 instruction per SIMD at the max clock and
// the implied chip-wide lane-ops/s.  Used to calibrate the VALU roofline in DESIGN.md / bench.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// 8 independent chains, R repeats of one 8-instruction asm block per loop iteration.
#define REP8(X) X X X X X X X X
#define BODY(INS)                                                                        \
    asm volatile(INS(%0) INS(%1) INS(%2) INS(%3) INS(%4) INS(%5) INS(%6) INS(%7)         \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                 : "v"(b), "s"(s), "v"(c));

#define DEFINE_KERNEL(NAME, INS)                                                         \
    __global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t s, int iters) \
    {                                                                                    \
        uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11,  \
                 a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;                               \
        uint32_t b = a0 ^ 0x5a5a5a5a, c = a0 * 977 + 1;                                  \
        for (int i = 0; i < iters; ++i) { REP8(BODY(INS)) }                              \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
    }

// operand numbering inside BODY: %0..%7 chains, %8 = b (VGPR), %9 = s (SGPR), %10 = c (VGPR)
#define I_XOR(r)      "v_xor_b32 " #r ", %8, " #r "\n"
#define I_XOR_S(r)    "v_xor_b32 " #r ", %9, " #r "\n"
#define I_BCNT(r)     "v_bcnt_u32_b32 " #r ", %8, " #r "\n"
#define I_MIN(r)      "v_min_u32 " #r ", %8, " #r "\n"
#define I_MED3(r)     "v_med3_u32 " #r ", " #r ", %8, %10\n"
#define I_MIN3(r)     "v_min3_u32 " #r ", " #r ", %8, %10\n"
#define I_LSHLOR(r)   "v_lshl_or_b32 " #r ", " #r ", 3, %8\n"
#define I_LSHLOR_S(r) "v_lshl_or_b32 " #r ", " #r ", 3, %9\n"
#define I_ADD(r)      "v_add_u32 " #r ", %8, " #r "\n"
#define I_ADD3(r)     "v_add3_u32 " #r ", " #r ", %8, %10\n"
#define I_ANDOR(r)    "v_and_or_b32 " #r ", " #r ", %8, %10\n"
#define I_BFE(r)      "v_bfe_u32 " #r ", " #r ", 3, 9\n"
#define I_PERM(r)     "v_perm_b32 " #r ", " #r ", %8, %10\n"
#define I_MOV(r)      "v_mov_b32 " #r ", %8\n"
#define I_LSHL(r)     "v_lshlrev_b32 " #r ", 1, " #r "\n"
#define I_CNDMASK(r)  "v_cndmask_b32 " #r ", " #r ", %8, vcc\n"
#define I_FMA(r)      "v_fma_f32 " #r ", " #r ", %8, %10\n"
#define I_BITOP3(r)   "v_bitop3_b32 " #r ", " #r ", %8, %10 bitop3:0x96\n"
#define I_PKADD16(r)  "v_pk_add_u16 " #r ", " #r ", %8\n"
#define I_PKMIN16(r)  "v_pk_min_u16 " #r ", " #r ", %8\n"
#define I_SAD8(r)     "v_sad_u8 " #r ", " #r ", %8, %10\n"
#define I_DOT4(r)     "v_dot4_u32_u8 " #r ", " #r ", %8, %10\n"
#define I_MAD24(r)    "v_mad_u32_u24 " #r ", " #r ", %8, %10\n"
#define I_MIN_DPPQ(r) "v_min_u32_dpp " #r ", " #r ", " #r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_MIN_DPPR(r) "v_min_u32_dpp " #r ", " #r ", " #r " row_ror:8 row_mask:0xf bank_mask:0xf\n"
#define I_MAX_DPPM(r) "v_max_u32_dpp " #r ", " #r ", " #r " row_mirror row_mask:0xf bank_mask:0xf\n"
#define I_MOV_DPPB(r) "v_mov_b32_dpp " #r ", " #r " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_CMP(r)      "v_cmp_lt_u32 vcc, " #r ", %8\n"
#define I_READLANE(r) "v_readlane_b32 s20, " #r ", 5\n"

#define I_XOR_E64S(r) "v_xor_b32_e64 " #r ", " #r ", %9\n"
#define I_BITOP3_S(r) "v_bitop3_b32 " #r ", " #r ", %9, %9 bitop3:0x66\n"
#define I_BITOP3_S2(r) "v_bitop3_b32 " #r ", %9, " #r ", " #r " bitop3:0x3c\n"
#define I_ADD_S(r)    "v_add_u32 " #r ", %9, " #r "\n"
#define I_MOV_S(r)    "v_mov_b32 " #r ", %9\n"
#define I_XNOR_S(r)   "v_xnor_b32 " #r ", %9, " #r "\n"
#define I_AND_S(r)    "v_and_b32 " #r ", %9, " #r "\n"
#define I_XOR_LIT(r)  "v_xor_b32 " #r ", 0x12345678, " #r "\n"
#define I_XOR_INL(r)  "v_xor_b32 " #r ", 15, " #r "\n"
DEFINE_KERNEL(xor_e64_sgpr, I_XOR_E64S)
DEFINE_KERNEL(bitop3_sgpr_src12, I_BITOP3_S)
DEFINE_KERNEL(bitop3_sgpr_src0, I_BITOP3_S2)
DEFINE_KERNEL(add_u32_sgpr, I_ADD_S)
DEFINE_KERNEL(mov_from_sgpr, I_MOV_S)
DEFINE_KERNEL(xnor_sgpr, I_XNOR_S)
DEFINE_KERNEL(and_sgpr, I_AND_S)
DEFINE_KERNEL(xor_literal, I_XOR_LIT)
DEFINE_KERNEL(xor_inline_const, I_XOR_INL)
DEFINE_KERNEL(xor, I_XOR)
DEFINE_KERNEL(xor_sgpr, I_XOR_S)
DEFINE_KERNEL(bcnt, I_BCNT)
DEFINE_KERNEL(min_u32, I_MIN)
DEFINE_KERNEL(med3_u32, I_MED3)
DEFINE_KERNEL(min3_u32, I_MIN3)
DEFINE_KERNEL(lshl_or, I_LSHLOR)
DEFINE_KERNEL(lshl_or_sgpr, I_LSHLOR_S)
DEFINE_KERNEL(add_u32, I_ADD)
DEFINE_KERNEL(add3_u32, I_ADD3)
DEFINE_KERNEL(and_or, I_ANDOR)
DEFINE_KERNEL(bfe_u32, I_BFE)
DEFINE_KERNEL(perm_b32, I_PERM)
DEFINE_KERNEL(mov, I_MOV)
DEFINE_KERNEL(lshlrev, I_LSHL)
DEFINE_KERNEL(cndmask, I_CNDMASK)
DEFINE_KERNEL(fma_f32, I_FMA)
DEFINE_KERNEL(bitop3, I_BITOP3)
DEFINE_KERNEL(pk_add_u16, I_PKADD16)
DEFINE_KERNEL(pk_min_u16, I_PKMIN16)
DEFINE_KERNEL(sad_u8, I_SAD8)
DEFINE_KERNEL(dot4_u32_u8, I_DOT4)
DEFINE_KERNEL(mad_u32_u24, I_MAD24)
DEFINE_KERNEL(min_dpp_quad, I_MIN_DPPQ)
DEFINE_KERNEL(min_dpp_row_ror, I_MIN_DPPR)
DEFINE_KERNEL(max_dpp_row_mirror, I_MAX_DPPM)
DEFINE_KERNEL(mov_dpp_row_bcast, I_MOV_DPPB)
DEFINE_KERNEL(cmp_lt_u32, I_CMP)

// the real inner loop mix: 8 xor(sgpr) + 8 bcnt + lshl_or + med3 + min  (19 ops)
__global__ void __launch_bounds__(256) k_mix19(uint32_t* out, uint32_t s, int iters)
{
    uint32_t q0 = threadIdx.x, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 11, q5 = q0 * 13,
             q6 = q0 * 17, q7 = q0 * 19;
    uint32_t b0 = ~0u, b1 = ~0u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t t0, t1, t2, t3, t4, t5, t6, t7, d;
            asm volatile(
                "v_xor_b32 %0, %19, %10\n v_xor_b32 %1, %19, %11\n v_xor_b32 %2, %19, %12\n"
                "v_xor_b32 %3, %19, %13\n v_xor_b32 %4, %19, %14\n v_xor_b32 %5, %19, %15\n"
                "v_xor_b32 %6, %19, %16\n v_xor_b32 %7, %19, %17\n"
                "v_bcnt_u32_b32 %8, %0, 0\n v_bcnt_u32_b32 %8, %1, %8\n v_bcnt_u32_b32 %8, %2, %8\n"
                "v_bcnt_u32_b32 %8, %3, %8\n v_bcnt_u32_b32 %8, %4, %8\n v_bcnt_u32_b32 %8, %5, %8\n"
                "v_bcnt_u32_b32 %8, %6, %8\n v_bcnt_u32_b32 %8, %7, %8\n"
                "v_lshl_or_b32 %8, %8, 23, %19\n"
                : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7),
                  "=&v"(d), "+v"(b0)
                : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(b1), "s"(s + i + u));
            uint32_t nb1;
            asm volatile("v_med3_u32 %0, %1, %2, %3\n" : "=v"(nb1) : "v"(b0), "v"(b1), "v"(d));
            asm volatile("v_min_u32 %0, %0, %1\n" : "+v"(b0) : "v"(d));
            b1 = nb1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = b0 ^ b1;
}

// LDS: wave-private transposed u16 tile write + read (the symmetric scan's column path)
__global__ void __launch_bounds__(256) k_lds_w16(uint32_t* out, uint32_t s, int iters)
{
    __shared__ uint16_t tile[4][64 * 72];
    uint16_t* t = tile[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) t[j * 72 + lane] = (uint16_t)(s + j + i);
        acc += t[(lane * 72 + i) & 4095];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

// symmetric row-mode mix, train row from SGPRs (current kernel): 8 xor(s) + 8 bcnt + 3 + ds_write_b16
__global__ void __launch_bounds__(256) k_mix_sym_sgpr(uint32_t* out, uint32_t s, int iters)
{
    __shared__ uint16_t tile[4][64 * 72];
    uint16_t* wr = tile[threadIdx.x >> 6] + (threadIdx.x & 63);
    uint32_t q0 = threadIdx.x, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 11, q5 = q0 * 13,
             q6 = q0 * 17, q7 = q0 * 19;
    uint32_t b0 = ~0u, b1 = ~0u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            uint32_t t0, t1, t2, t3, t4, t5, t6, t7, d;
            asm volatile(
                "v_xor_b32 %0, %19, %10\n v_xor_b32 %1, %19, %11\n v_xor_b32 %2, %19, %12\n"
                "v_xor_b32 %3, %19, %13\n v_xor_b32 %4, %19, %14\n v_xor_b32 %5, %19, %15\n"
                "v_xor_b32 %6, %19, %16\n v_xor_b32 %7, %19, %17\n"
                "v_bcnt_u32_b32 %8, %0, 0\n v_bcnt_u32_b32 %8, %1, %8\n v_bcnt_u32_b32 %8, %2, %8\n"
                "v_bcnt_u32_b32 %8, %3, %8\n v_bcnt_u32_b32 %8, %4, %8\n v_bcnt_u32_b32 %8, %5, %8\n"
                "v_bcnt_u32_b32 %8, %6, %8\n v_bcnt_u32_b32 %8, %7, %8\n"
                : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7),
                  "=&v"(d), "+v"(b0)
                : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(b1), "s"(s + i + u));
            wr[((i * 4 + u) & 63) * 72] = (uint16_t)d;
            uint32_t key, nb1;
            asm volatile("v_lshl_or_b32 %0, %1, 23, %2\n" : "=v"(key) : "v"(d), "s"(s + i));
            asm volatile("v_med3_u32 %0, %1, %2, %3\n" : "=v"(nb1) : "v"(b0), "v"(b1), "v"(key));
            asm volatile("v_min_u32 %0, %0, %1\n" : "+v"(b0) : "v"(key));
            b1 = nb1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = b0 ^ b1;
}

// same, train row broadcast-read from LDS into VGPRs: 2 ds_read_b128 + 8 xor(v,v) + 8 bcnt + 3 + ds_write_b16
__global__ void __launch_bounds__(256) k_mix_sym_lds(uint32_t* out, uint32_t s, int iters)
{
    __shared__ __attribute__((aligned(16))) uint16_t tile[4][64 * 72];
    __shared__ __attribute__((aligned(16))) uint32_t brow[64 * 8];
    for (int i = threadIdx.x; i < 64 * 8; i += 256) brow[i] = i * 2654435761u + s;
    __syncthreads();
    uint16_t* wr = tile[threadIdx.x >> 6] + (threadIdx.x & 63);
    uint32_t q0 = threadIdx.x, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 11, q5 = q0 * 13,
             q6 = q0 * 17, q7 = q0 * 19;
    uint32_t b0 = ~0u, b1 = ~0u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint4* rp = reinterpret_cast<const uint4*>(brow + (((i * 4 + u) & 63) * 8));
            const uint4 ta = rp[0], tb = rp[1];
            uint32_t t0, t1, t2, t3, t4, t5, t6, t7, d;
            asm volatile(
                "v_xor_b32 %0, %19, %10\n v_xor_b32 %1, %20, %11\n v_xor_b32 %2, %21, %12\n"
                "v_xor_b32 %3, %22, %13\n v_xor_b32 %4, %23, %14\n v_xor_b32 %5, %24, %15\n"
                "v_xor_b32 %6, %25, %16\n v_xor_b32 %7, %26, %17\n"
                "v_bcnt_u32_b32 %8, %0, 0\n v_bcnt_u32_b32 %8, %1, %8\n v_bcnt_u32_b32 %8, %2, %8\n"
                "v_bcnt_u32_b32 %8, %3, %8\n v_bcnt_u32_b32 %8, %4, %8\n v_bcnt_u32_b32 %8, %5, %8\n"
                "v_bcnt_u32_b32 %8, %6, %8\n v_bcnt_u32_b32 %8, %7, %8\n"
                : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7),
                  "=&v"(d), "+v"(b0)
                : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(b1),
                  "v"(ta.x), "v"(ta.y), "v"(ta.z), "v"(ta.w), "v"(tb.x), "v"(tb.y), "v"(tb.z), "v"(tb.w));
            wr[((i * 4 + u) & 63) * 72] = (uint16_t)d;
            uint32_t key, nb1;
            asm volatile("v_lshl_or_b32 %0, %1, 23, %2\n" : "=v"(key) : "v"(d), "s"(s + i));
            asm volatile("v_med3_u32 %0, %1, %2, %3\n" : "=v"(nb1) : "v"(b0), "v"(b1), "v"(key));
            asm volatile("v_min_u32 %0, %0, %1\n" : "+v"(b0) : "v"(key));
            b1 = nb1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = b0 ^ b1;
}

// how do fast (xor v,v) and slow (bcnt) ops mix?  alternating vs blocked, 8 independent chains
#define I_XB_ALT(r)   "v_xor_b32 " #r ", %8, " #r "\n v_bcnt_u32_b32 " #r ", %10, " #r "\n"
__global__ void __launch_bounds__(256) k_alt_xor_bcnt(uint32_t* out, uint32_t s, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = a0 ^ 0x5a5a5a5a, c = a0 * 977 + 1;
    for (int i = 0; i < iters; ++i) { REP8(BODY(I_XB_ALT)) }
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
// xor into a temp then bcnt of the temp accumulating into the chain (the real dependency shape)
__global__ void __launch_bounds__(256) k_dep_xor_bcnt(uint32_t* out, uint32_t s, int iters)
{
    uint32_t q0 = threadIdx.x, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 11, q5 = q0 * 13, q6 = q0 * 17, q7 = q0 * 19;
    uint32_t t = q0 ^ 0x5a5a5a5a, acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t x0, x1, x2, x3, x4, x5, x6, x7;
            asm volatile(
                "v_xor_b32 %0, %17, %9\n v_xor_b32 %1, %17, %10\n v_xor_b32 %2, %17, %11\n v_xor_b32 %3, %17, %12\n"
                "v_xor_b32 %4, %17, %13\n v_xor_b32 %5, %17, %14\n v_xor_b32 %6, %17, %15\n v_xor_b32 %7, %17, %16\n"
                "v_bcnt_u32_b32 %8, %0, %8\n v_bcnt_u32_b32 %8, %1, %8\n v_bcnt_u32_b32 %8, %2, %8\n v_bcnt_u32_b32 %8, %3, %8\n"
                "v_bcnt_u32_b32 %8, %4, %8\n v_bcnt_u32_b32 %8, %5, %8\n v_bcnt_u32_b32 %8, %6, %8\n v_bcnt_u32_b32 %8, %7, %8\n"
                : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7), "+v"(acc)
                : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(t));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// same with 4 independent accumulators interleaved (xor block of 8, then 8 bcnt on 4 chains)
__global__ void __launch_bounds__(256) k_dep4_xor_bcnt(uint32_t* out, uint32_t s, int iters)
{
    uint32_t q0 = threadIdx.x, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7, q4 = q0 * 11, q5 = q0 * 13, q6 = q0 * 17, q7 = q0 * 19;
    uint32_t t = q0 ^ 0x5a5a5a5a, acc0 = 0, acc1 = 1, acc2 = 2, acc3 = 3;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t x0, x1, x2, x3, x4, x5, x6, x7;
            asm volatile(
                "v_xor_b32 %0, %20, %12\n v_xor_b32 %1, %20, %13\n v_xor_b32 %2, %20, %14\n v_xor_b32 %3, %20, %15\n"
                "v_xor_b32 %4, %20, %16\n v_xor_b32 %5, %20, %17\n v_xor_b32 %6, %20, %18\n v_xor_b32 %7, %20, %19\n"
                "v_bcnt_u32_b32 %8, %0, %8\n v_bcnt_u32_b32 %9, %1, %9\n v_bcnt_u32_b32 %10, %2, %10\n v_bcnt_u32_b32 %11, %3, %11\n"
                "v_bcnt_u32_b32 %8, %4, %8\n v_bcnt_u32_b32 %9, %5, %9\n v_bcnt_u32_b32 %10, %6, %10\n v_bcnt_u32_b32 %11, %7, %11\n"
                : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7),
                  "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3)
                : "v"(q0), "v"(q1), "v"(q2), "v"(q3), "v"(q4), "v"(q5), "v"(q6), "v"(q7), "v"(t));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc0 ^ acc1 ^ acc2 ^ acc3;
}

// runs of N independent fast xors followed by N slow bcnts (8 rotating chains): does run length matter?
template <int N>
__global__ void __launch_bounds__(256) k_runs(uint32_t* out, uint32_t s, int iters)
{
    uint32_t a[8], c[8];
    for (int k = 0; k < 8; ++k) { a[k] = threadIdx.x * (2 * k + 3); c[k] = k; }
    const uint32_t b = threadIdx.x ^ 0x5a5a5a5a;
    const int reps = 64 / N;   // 64 fast + 64 slow per iteration in total
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < reps; ++r) {
#pragma unroll
            for (int k = 0; k < N; ++k) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[k & 7]) : "v"(b));
#pragma unroll
            for (int k = 0; k < N; ++k) asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c[k & 7]) : "v"(a[k & 7]));
        }
    }
    uint32_t o = 0;
    for (int k = 0; k < 8; ++k) o ^= a[k] ^ c[k];
    out[blockIdx.x * 256 + threadIdx.x] = o;
}

typedef void (*kern_t)(uint32_t*, uint32_t, int);
struct Entry { const char* name; kern_t k; double ops_per_iter; };

int main(int argc, char** argv)
{
    int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    printf("device %s %s CUs=%d maxclk=%.0f MHz\n", prop.name, prop.gcnArchName, cus, clk / 1e6);
    const int wps = argc > 2 ? atoi(argv[2]) : 8;
    const int blocks = cus * wps;  // wps waves per SIMD (256-thread blocks, one wave per SIMD each)
    printf("waves per SIMD: %d\n", wps);
    uint32_t* out;
    CHECK(hipMalloc(&out, sizeof(uint32_t) * blocks * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
#define E(NAME) {#NAME, k_##NAME, 64.0}
    Entry es[] = {{"runs: 1 fast / 1 slow", k_runs<1>, 128.0}, {"runs: 2 fast / 2 slow", k_runs<2>, 128.0},
                  {"runs: 4 fast / 4 slow", k_runs<4>, 128.0}, {"runs: 8 fast / 8 slow", k_runs<8>, 128.0},
                  {"runs: 16 fast / 16 slow", k_runs<16>, 128.0}, {"runs: 32 fast / 32 slow", k_runs<32>, 128.0},
                  {"runs: 64 fast / 64 slow", k_runs<64>, 128.0},
                  {"alt xor(v,v)/bcnt, 8 indep chains", k_alt_xor_bcnt, 128.0},
                  {"8 xor(v,v) then 8 bcnt on ONE chain", k_dep_xor_bcnt, 128.0},
                  {"8 xor(v,v) then 8 bcnt on FOUR chains", k_dep4_xor_bcnt, 128.0},
                  E(xor_e64_sgpr), E(bitop3_sgpr_src12), E(bitop3_sgpr_src0), E(add_u32_sgpr), E(mov_from_sgpr),
                  E(xnor_sgpr), E(and_sgpr), E(xor_literal), E(xor_inline_const),
                  {"mix_sym_sgpr (19 valu + ds_write_b16)", k_mix_sym_sgpr, 76.0},
                  {"mix_sym_lds (2 ds_read_b128 bcast + 19 valu + ds_write_b16)", k_mix_sym_lds, 76.0},
                  E(xor), E(xor_sgpr), E(bcnt), E(min_u32), E(med3_u32), E(min3_u32), E(lshl_or),
                  E(lshl_or_sgpr), E(add_u32), E(add3_u32), E(and_or), E(bfe_u32), E(perm_b32), E(mov),
                  E(lshlrev), E(cndmask), E(fma_f32), E(bitop3), E(pk_add_u16), E(pk_min_u16), E(sad_u8),
                  E(dot4_u32_u8), E(mad_u32_u24), E(min_dpp_quad), E(min_dpp_row_ror),
                  E(max_dpp_row_mirror), E(mov_dpp_row_bcast), E(cmp_lt_u32),
                  {"mix19(xor8+bcnt8+lshl_or+med3+min)", k_mix19, 76.0},
                  {"lds_write_b16 x64 + 1 read", k_lds_w16, 65.0}};
    for (const Entry& e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u, 10);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 12345u, iters);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        const double wave_instr = (double)blocks * 4 * iters * e.ops_per_iter;  // 4 waves per block
        const double cyc = best * 1e-3 * clk * (cus * 4.0) / wave_instr;
        printf("%-62s %8.3f ms  %6.2f cyc/wave-instr/SIMD @maxclk  %7.2f T lane-ops/s\n", e.name, best,
               cyc, wave_instr * 64 / (best * 1e-3) / 1e12);
    }
    return 0;
}
