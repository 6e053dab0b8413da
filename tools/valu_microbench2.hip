// valu_microbench2.hip -- round 2: issue rates of the FLOAT min/max/med3 family and of the packed-f16 3-input ops on
// gfx950, and of the two candidate best-2 bookkeeping mixes of the matrix-core Hamming scan (K1e):
//   packed-u16 (round 1): v_perm + 3 v_pk_{max,min,min}_u16 (row) + 3 (column) per TWO distances
//   plain-f32  (round 2): v_med3_f32 + v_min_f32 (row) + v_med3_f32 + v_min_f32 (column) per ONE distance
// with and without MFMAs (v_mfma_scale_f32_32x32x64_f8f6f4, fp4 operands) in the same wave.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_microbench2.hip -o build/valu_microbench2
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(X) X X X X X X X X
#define BODY(INS)                                                                        \
    asm volatile(INS(%0) INS(%1) INS(%2) INS(%3) INS(%4) INS(%5) INS(%6) INS(%7)         \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                 : "v"(b), "s"(s), "v"(c));

// float-safe start values: 2^23 + small integers (the accumulator form of the scan), never NaN
#define DEFINE_KERNEL(NAME, INS)                                                         \
    __global__ void __launch_bounds__(256) k_##NAME(uint32_t* out, uint32_t s, int iters) \
    {                                                                                    \
        const uint32_t t = threadIdx.x;                                                  \
        uint32_t a0 = 0x4B000000u + t, a1 = 0x4B000000u + t * 3, a2 = 0x4B000000u + t * 5, a3 = 0x4B000000u + t * 7, \
                 a4 = 0x4B000000u + t * 11, a5 = 0x4B000000u + t * 13, a6 = 0x4B000000u + t * 17, a7 = 0x4B000000u + t * 19; \
        uint32_t b = 0x4B000000u + (t ^ 0x5a), c = 0x4B000000u + t * 977 + 1;           \
        for (int i = 0; i < iters; ++i) { REP8(BODY(INS)) }                              \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
    }

#define I_MIN_F32(r)    "v_min_f32 " #r ", %8, " #r "\n"
#define I_MAX_F32(r)    "v_max_f32 " #r ", %8, " #r "\n"
#define I_MED3_F32(r)   "v_med3_f32 " #r ", " #r ", %8, %10\n"
#define I_MIN3_F32(r)   "v_min3_f32 " #r ", " #r ", %8, %10\n"
#define I_MAX3_F32(r)   "v_max3_f32 " #r ", " #r ", %8, %10\n"
#define I_MINIMUM3_F32(r) "v_minimum3_f32 " #r ", " #r ", %8, %10\n"
#define I_MAXIMUM3_F32(r) "v_maximum3_f32 " #r ", " #r ", %8, %10\n"
#define I_PKMIN_F16(r)  "v_pk_min_f16 " #r ", " #r ", %8\n"
#define I_PKMAX_F16(r)  "v_pk_max_f16 " #r ", " #r ", %8\n"
#define I_PKMIN3_F16(r) "v_pk_minimum3_f16 " #r ", " #r ", %8, %10\n"
#define I_PKMAX3_F16(r) "v_pk_maximum3_f16 " #r ", " #r ", %8, %10\n"
#define I_PKMIN_U16(r)  "v_pk_min_u16 " #r ", " #r ", %8\n"
#define I_PKMIN_I16(r)  "v_pk_min_i16 " #r ", " #r ", %8\n"
#define I_MIN_U16(r)    "v_min_u16 " #r ", %8, " #r "\n"
#define I_MED3_U16(r)   "v_med3_u16 " #r ", " #r ", %8, %10\n"
#define I_MIN_I32(r)    "v_min_i32 " #r ", %8, " #r "\n"
#define I_MIN_U32(r)    "v_min_u32 " #r ", %8, " #r "\n"
#define I_AND(r)        "v_and_b32 " #r ", %8, " #r "\n"
#define I_OR(r)         "v_or_b32 " #r ", %8, " #r "\n"
#define I_SUB(r)        "v_sub_u32 " #r ", %8, " #r "\n"
#define I_ADD_F32(r)    "v_add_f32 " #r ", %8, " #r "\n"
#define I_MUL_F32(r)    "v_mul_f32 " #r ", %8, " #r "\n"
#define I_LSHLADD(r)    "v_lshl_add_u32 " #r ", " #r ", 1, %8\n"
#define I_ANDOR(r)      "v_and_or_b32 " #r ", " #r ", %8, %10\n"
#define I_PERM(r)       "v_perm_b32 " #r ", " #r ", %8, %10\n"
#define I_CVTPK(r)      "v_cvt_pk_u16_u32 " #r ", " #r ", %8\n"
#define I_MIN_F32_S(r)  "v_min_f32 " #r ", %9, " #r "\n"
#define I_MIN_F32_DPP(r) "v_min_f32_dpp " #r ", " #r ", " #r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define I_ROWPUSH_F32(r) "v_med3_f32 %10, " #r ", %10, %8\n v_min_f32 " #r ", %8, " #r "\n"
DEFINE_KERNEL(min_f32, I_MIN_F32)
DEFINE_KERNEL(max_f32, I_MAX_F32)
DEFINE_KERNEL(med3_f32, I_MED3_F32)
DEFINE_KERNEL(min3_f32, I_MIN3_F32)
DEFINE_KERNEL(max3_f32, I_MAX3_F32)
DEFINE_KERNEL(minimum3_f32, I_MINIMUM3_F32)
DEFINE_KERNEL(maximum3_f32, I_MAXIMUM3_F32)
DEFINE_KERNEL(pk_min_f16, I_PKMIN_F16)
DEFINE_KERNEL(pk_max_f16, I_PKMAX_F16)
DEFINE_KERNEL(pk_minimum3_f16, I_PKMIN3_F16)
DEFINE_KERNEL(pk_maximum3_f16, I_PKMAX3_F16)
DEFINE_KERNEL(pk_min_u16, I_PKMIN_U16)
DEFINE_KERNEL(pk_min_i16, I_PKMIN_I16)
DEFINE_KERNEL(min_u16, I_MIN_U16)
DEFINE_KERNEL(med3_u16, I_MED3_U16)
DEFINE_KERNEL(min_i32, I_MIN_I32)
DEFINE_KERNEL(min_u32, I_MIN_U32)
DEFINE_KERNEL(and_b32, I_AND)
DEFINE_KERNEL(or_b32, I_OR)
DEFINE_KERNEL(sub_u32, I_SUB)
DEFINE_KERNEL(add_f32, I_ADD_F32)
DEFINE_KERNEL(mul_f32, I_MUL_F32)
DEFINE_KERNEL(lshl_add_u32, I_LSHLADD)
DEFINE_KERNEL(and_or_b32, I_ANDOR)
DEFINE_KERNEL(perm_b32, I_PERM)
DEFINE_KERNEL(cvt_pk_u16_u32, I_CVTPK)
DEFINE_KERNEL(min_f32_sgpr, I_MIN_F32_S)
DEFINE_KERNEL(min_f32_dpp_quad, I_MIN_F32_DPP)

// 64-bit moves / packed f32 (register pairs)
__device__ __forceinline__ void pk_ops(uint32_t* out, uint32_t s, int iters, int which)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8], b = {1.0f + threadIdx.x, 2.0f};
    for (int k = 0; k < 8; ++k) a[k] = f2{(float)threadIdx.x * k, 1.0f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (which == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
                else if (which == 1) asm volatile("v_mov_b64 %0, %1" : "+v"(a[k]) : "v"(b));
                else asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(a[k]) : "v"(b));
            }
    }
    float o = 0;
    for (int k = 0; k < 8; ++k) o += a[k].x + a[k].y;
    out[blockIdx.x * 256 + threadIdx.x] = __builtin_bit_cast(uint32_t, o);
}
__global__ void __launch_bounds__(256) k_pk_add_f32(uint32_t* out, uint32_t s, int iters) { pk_ops(out, s, iters, 0); }
__global__ void __launch_bounds__(256) k_mov_b64(uint32_t* out, uint32_t s, int iters) { pk_ops(out, s, iters, 1); }
__global__ void __launch_bounds__(256) k_pk_mov_b32(uint32_t* out, uint32_t s, int iters) { pk_ops(out, s, iters, 2); }

// ---------------------------------------------------------------------------------------------------------
// Bookkeeping mixes over 32 "accumulator" registers per lane (two M-tiles x 16), as in the scan.
// NEW[r] stands for the accumulator (changes per iteration so nothing is hoisted).
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, bool WITH_MFMA>   // MODE 0: packed u16 keys, 1: plain f32 keys
__global__ void __launch_bounds__(256, 2) k_book(uint32_t* out, uint32_t s, int iters)
{
    const uint32_t t = threadIdx.x;
    f32x16 A0, A1, B0, B1;
    for (int r = 0; r < 16; ++r) {
        A0[r] = 8388608.0f + (float)(t * (r + 1) & 0xFFF); A1[r] = 8388608.0f + (float)(t * (r + 3) & 0xFFF);
        B0[r] = 8388608.0f + (float)(t * (r + 5) & 0xFFF); B1[r] = 8388608.0f + (float)(t * (r + 7) & 0xFFF);
    }
    const i32x8 av0 = {(int)(0x22222222u ^ (t * 0x9E3779B1u & 0x88888888u)), 0x2A2A2A2A, 0x22222222, 0x2222AAAA, 0, 0, 0, 0};
    const i32x8 av1 = {(int)(0x22222222u ^ (t * 0x85EBCA6Bu & 0x88888888u)), 0x2A2A2A22, 0x2222A222, 0x2A22AAAA, 0, 0, 0, 0};
    i32x8 bv = {(int)(0x22222222u ^ (t * 0x7F4A7C15u & 0x88888888u)), 0x22AA22AA, 0x2222A222, 0x22222222, 0, 0, 0, 0};
    f32x16 cinit;
    for (int r = 0; r < 16; ++r) cinit[r] = 8388608.0f + 16384.0f + (float)r;
    uint32_t colacc = 0;
    uint32_t prb0[16], prb1[16];
    float frb0[32], frb1[32];
    for (int r = 0; r < 16; ++r) prb0[r] = prb1[r] = 0xFFFFFFFFu;
    for (int r = 0; r < 32; ++r) frb0[r] = frb1[r] = 3.0e38f;
    // one step: the 8 MFMAs of a tile into (m0, m1) interleaved with the bookkeeping of the previous tile (acc0, acc1)
    auto step = [&](f32x16& m0, f32x16& m1, const f32x16& acc0, const f32x16& acc1) __attribute__((always_inline)) {
        uint32_t pcb0 = 0xFFFFFFFFu, pcb1 = 0xFFFFFFFFu;
        float fcb0 = 3.0e38f, fcb1 = 3.0e38f;
        bv.x ^= 0x80808080;                      // a new b tile per step
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (WITH_MFMA) m0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av0, bv, ks == 0 ? cinit : m0, 4, 4, 0, 133, 0, 127);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int r = 4 * ks + rr;
                const float f0 = acc0[r], f1 = acc1[r];
                if (MODE == 0) {
                    uint32_t kc = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, f1), __builtin_bit_cast(uint32_t, f0), 0x05040100u);
                    uint32_t x, y;
                    asm volatile("v_pk_max_u16 %0, %2, %3\n v_pk_min_u16 %1, %1, %0\n v_pk_min_u16 %2, %2, %3" : "=&v"(x), "+v"(prb1[r]), "+v"(prb0[r]) : "v"(kc));
                    asm volatile("v_pk_max_u16 %0, %2, %3\n v_pk_min_u16 %1, %1, %0\n v_pk_min_u16 %2, %2, %3" : "=&v"(y), "+v"(pcb1), "+v"(pcb0) : "v"(kc));
                } else {
                    asm volatile("v_med3_f32 %0, %1, %0, %2\n v_min_f32 %1, %1, %2" : "+v"(frb1[r]), "+v"(frb0[r]) : "v"(f0));
                    asm volatile("v_med3_f32 %0, %1, %0, %2\n v_min_f32 %1, %1, %2" : "+v"(fcb1), "+v"(fcb0) : "v"(f0));
                    asm volatile("v_med3_f32 %0, %1, %0, %2\n v_min_f32 %1, %1, %2" : "+v"(frb1[16 + r]), "+v"(frb0[16 + r]) : "v"(f1));
                    asm volatile("v_med3_f32 %0, %1, %0, %2\n v_min_f32 %1, %1, %2" : "+v"(fcb1), "+v"(fcb0) : "v"(f1));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (WITH_MFMA) m1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av1, bv, ks == 0 ? cinit : m1, 4, 4, 0, 133, 0, 127);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rr = 2; rr < 4; ++rr) {
                const int r = 4 * ks + rr;
                const float f0 = acc0[r], f1 = acc1[r];
                if (MODE == 0) {
                    uint32_t kc = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, f1), __builtin_bit_cast(uint32_t, f0), 0x05040100u);
                    uint32_t x, y;
                    asm volatile("v_pk_max_u16 %0, %2, %3\n v_pk_min_u16 %1, %1, %0\n v_pk_min_u16 %2, %2, %3" : "=&v"(x), "+v"(prb1[r]), "+v"(prb0[r]) : "v"(kc));
                    asm volatile("v_pk_max_u16 %0, %2, %3\n v_pk_min_u16 %1, %1, %0\n v_pk_min_u16 %2, %2, %3" : "=&v"(y), "+v"(pcb1), "+v"(pcb0) : "v"(kc));
                } else {
                    asm volatile("v_med3_f32 %0, %1, %0, %2\n v_min_f32 %1, %1, %2" : "+v"(frb1[r]), "+v"(frb0[r]) : "v"(f0));
                    asm volatile("v_med3_f32 %0, %1, %0, %2\n v_min_f32 %1, %1, %2" : "+v"(fcb1), "+v"(fcb0) : "v"(f0));
                    asm volatile("v_med3_f32 %0, %1, %0, %2\n v_min_f32 %1, %1, %2" : "+v"(frb1[16 + r]), "+v"(frb0[16 + r]) : "v"(f1));
                    asm volatile("v_med3_f32 %0, %1, %0, %2\n v_min_f32 %1, %1, %2" : "+v"(fcb1), "+v"(fcb0) : "v"(f1));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        colacc ^= pcb0 + pcb1 + __builtin_bit_cast(uint32_t, fcb0) + __builtin_bit_cast(uint32_t, fcb1);
        if (!WITH_MFMA) { asm volatile("" : "+v"(m0), "+v"(m1)); }
    };
    for (int i = 0; i < iters; i += 2) {
        step(B0, B1, A0, A1);
        step(A0, A1, B0, B1);
    }
    for (int r = 0; r < 16; ++r) colacc ^= prb0[r] ^ prb1[r];
    for (int r = 0; r < 32; ++r) colacc ^= __builtin_bit_cast(uint32_t, frb0[r]) ^ __builtin_bit_cast(uint32_t, frb1[r]);
    out[blockIdx.x * 256 + threadIdx.x] = colacc;
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
    uint32_t* out;
    CHECK(hipMalloc(&out, sizeof(uint32_t) * cus * 8 * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
#define E(NAME) {#NAME, k_##NAME, 64.0}
    // ops_per_iter for the bookkeeping mixes = VALU instructions per iteration (one 64 x 32 tile per wave):
    //   packed: 16 x (perm + 6) = 112;  f32: 32 x 4 = 128
    Entry es[] = {E(min_f32), E(max_f32), E(med3_f32), E(min3_f32), E(max3_f32), E(minimum3_f32), E(maximum3_f32),
                  E(pk_min_f16), E(pk_max_f16), E(pk_minimum3_f16), E(pk_maximum3_f16), E(pk_min_u16), E(pk_min_i16),
                  E(min_u16), E(med3_u16), E(min_i32), E(min_u32), E(and_b32), E(or_b32), E(sub_u32), E(add_f32),
                  E(mul_f32), E(lshl_add_u32), E(and_or_b32), E(perm_b32), E(cvt_pk_u16_u32), E(min_f32_sgpr),
                  E(min_f32_dpp_quad), E(pk_add_f32), E(mov_b64), E(pk_mov_b32),
                  {"book packed-u16, no mfma (112 valu/tile)", k_book<0, false>, 112.0},
                  {"book packed-u16 + 8 mfma    (112 valu/tile)", k_book<0, true>, 112.0},
                  {"book plain-f32, no mfma  (128 valu/tile)", k_book<1, false>, 128.0},
                  {"book plain-f32 + 8 mfma     (128 valu/tile)", k_book<1, true>, 128.0}};
    for (int wps : {1, 2, 3, 4, 8}) {
        const int blocks = cus * wps;
        printf("## %d waves per SIMD\n", wps);
        for (const Entry& e : es) {
            if (wps != 2 && wps != 8 && e.ops_per_iter == 64.0) continue;      // single-instruction rows at 2 and 8 only
            if (wps == 8 && e.ops_per_iter != 64.0) continue;
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
            const double wave_instr = (double)blocks * 4 * iters * e.ops_per_iter;
            const double cyc = best * 1e-3 * clk * (cus * 4.0) / wave_instr;
            printf("%-48s %8.3f ms  %6.2f cyc/wave-instr/SIMD  %8.1f cyc/iter/wave-slot\n", e.name, best, cyc,
                   best * 1e-3 * clk / iters / wps);
        }
    }
    return 0;
}
