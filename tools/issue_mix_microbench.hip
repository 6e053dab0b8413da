// What does a SIMD of gfx950 issue per cycle when three waves run the instruction MIX of the matrix-core scan's tile loop?
// Per loop iteration ("tile") a wave issues NM MFMAs (v_mfma[_scale]_f32_32x32x64_f8f6f4 with fp4 operands, two accumulator
// chains), NV packed VALU ops (v_pk_min_u16 on 8 independent registers), NS scalar ops (s_add_u32 on 4 independent registers)
// and NL LDS reads (ds_read_b128), spread evenly.  Time is taken with s_memtime inside the kernel (shader cycles, whatever the
// clock does) and with HIP events.  Prints shader cycles per wave-iteration per SIMD at 1, 2 and 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/issue_mix_microbench.hip -o build/issue_mix_microbench && build/issue_mix_microbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// MODE bit 0: scaled MFMA (else unscaled); bit 1: the two chains issued back to back (4 + 4) instead of alternating;
// bit 2: a workgroup barrier per iteration; bit 3: every VALU op depends on the one before it (ONE chain instead of eight)
template <int NM, int NV, int NS, int NL, int MODE>
__global__ void __launch_bounds__(256) k(int iters, unsigned long long* out, int* sink)
{
    __shared__ __attribute__((aligned(16))) int lds[256 * 8];
    for (int i = threadIdx.x; i < 256 * 8; i += 256) lds[i] = i;
    __syncthreads();
    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 1.f; }
    i32x8 a = {(int)threadIdx.x, 0x22222222, 0x2a2a2a2a, 3, 0, 0, 0, 0}, b = {0x22222222, 5, (int)threadIdx.x, 7, 0, 0, 0, 0};
    uint32_t x[8], y = threadIdx.x * 2654435761u;
    for (int i = 0; i < 8; ++i) x[i] = y + i;
    uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4;
    i32x4 ld = {0, 0, 0, 0}, pend = {0, 0, 0, 0};
    const int* lp = lds + 4 * (threadIdx.x & 63);
    constexpr int SLOTS = NM > 0 ? NM : 8;           // the iteration is cut into SLOTS equal parts, one MFMA at the head of each
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE & 4) __syncthreads();
#pragma unroll
        for (int m = 0; m < SLOTS; ++m) {
            if (NM > 0) {
                const bool first = (MODE & 2) ? (m < SLOTS / 2) : ((m & 1) == 0);
                if (MODE & 1) {
                    if (first) acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc0, 4, 4, 0, 133, 0, 127);
                    else acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc1, 4, 4, 0, 133, 0, 127);
                } else {
                    if (first) acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc0, 4, 4, 0, 0, 0, 0);
                    else acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc1, 4, 4, 0, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            constexpr int NVs = (NV + SLOTS - 1) / SLOTS, NSs = (NS + SLOTS - 1) / SLOTS;
#pragma unroll
            for (int v = 0; v < NVs; ++v) {
                if (m * NVs + v < NV) {
                    if (MODE & 8) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x[0]) : "v"(y));
                    else asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x[v & 7]) : "v"(y));
                }
                // scalar ops between the vector ops, as the compiler schedules them
                if (v < NSs && m * NSs + v < NS) {
                    switch (v & 3) {
                        case 0: asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc"); break;
                        case 1: asm volatile("s_add_u32 %0, %0, 1" : "+s"(s1) : : "scc"); break;
                        case 2: asm volatile("s_add_u32 %0, %0, 1" : "+s"(s2) : : "scc"); break;
                        default: asm volatile("s_add_u32 %0, %0, 1" : "+s"(s3) : : "scc"); break;
                    }
                }
            }
            if (NSs > NVs) {
#pragma unroll
                for (int v = NVs; v < NSs; ++v)
                    if (m * NSs + v < NS) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
            }
            if (m < NL) {
                // issued now, consumed a slot later (the compiler places the wait in front of the use)
                ld += pend;
                pend = *reinterpret_cast<const volatile i32x4*>(lp + 256 * (m & 7));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    ld += pend;
    int r = (int)s + ld.x + ld.y + ld.z + ld.w + (int)(s0 + s1 + s2 + s3);
    for (int i = 0; i < 8; ++i) r += x[i];
    sink[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

static int g_cus;
static unsigned long long* g_out;
static int* g_sink;

template <int NM, int NV, int NS, int NL, int MODE>
static void run(const char* name, int iters)
{
    printf("%-58s", name);
    for (int wps = 1; wps <= 3; ++wps) {
        const int nb = g_cus * wps;
        double best = 1e30, best_ms = 1e30;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL((k<NM, NV, NS, NL, MODE>), dim3(nb), dim3(256), 0, 0, iters, g_out, g_sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            static unsigned long long h[4096];
            CHECK(hipMemcpy(h, g_out, sizeof(unsigned long long) * nb, hipMemcpyDeviceToHost));
            double sum = 0;
            for (int i = 0; i < nb; ++i) sum += (double)h[i];
            best = std::min(best, sum / nb / iters);
            best_ms = std::min(best_ms, (double)ms);
        }
        // s_memtime counts at a fixed 100 MHz on this chip if it is the REFCLK: print both, the caller sees which it is
        printf("  wps%d: %7.1f tick/iter/wave = %7.1f per wave-iter/SIMD, %6.3f ms", wps, best, best / wps, best_ms);
    }
    printf("\n");
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    g_cus = p.multiProcessorCount;
    printf("device %s CUs=%d nominal clk=%d MHz, %d iterations; ms x nominal clk / iters = nominal cycles per iteration\n", p.gcnArchName, g_cus, p.clockRate / 1000, iters);
    CHECK(hipMalloc(&g_out, sizeof(unsigned long long) * 4096));
    CHECK(hipMalloc(&g_sink, sizeof(int) * 256 * 4096));
    run<0, 104, 0, 0, 0>("valu 104", iters);
    run<0, 104, 0, 0, 8>("valu 104, one dependent chain", iters);
    run<0, 104, 42, 0, 0>("valu 104 + salu 42", iters);
    run<0, 104, 104, 0, 0>("valu 104 + salu 104", iters);
    run<0, 104, 0, 8, 0>("valu 104 + 8 ds_read_b128", iters);
    run<8, 0, 0, 0, 1>("mfma 8 scaled (alternating chains)", iters);
    run<8, 0, 0, 0, 0>("mfma 8 unscaled (alternating chains)", iters);
    run<8, 0, 0, 0, 3>("mfma 8 scaled (4 + 4 back to back)", iters);
    run<8, 104, 0, 0, 1>("valu 104 + mfma 8 scaled", iters);
    run<8, 104, 0, 0, 0>("valu 104 + mfma 8 unscaled", iters);
    run<8, 104, 0, 0, 3>("valu 104 + mfma 8 scaled (4 + 4)", iters);
    run<8, 64, 0, 0, 1>("valu 64 + mfma 8 scaled", iters);
    run<8, 32, 0, 0, 1>("valu 32 + mfma 8 scaled", iters);
    run<8, 104, 42, 0, 1>("valu 104 + salu 42 + mfma 8 scaled", iters);
    run<8, 104, 42, 8, 1>("valu 104 + salu 42 + 8 lds + mfma 8 scaled", iters);
    run<8, 104, 42, 8, 5>("valu 104 + salu 42 + 8 lds + mfma 8 scaled + barrier", iters);
    run<8, 104, 42, 8, 4>("valu 104 + salu 42 + 8 lds + mfma 8 unscaled + barrier", iters);
    return 0;
}
