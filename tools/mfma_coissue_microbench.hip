// Does the VALU of a SIMD keep issuing while v_mfma_i32_32x32x32_i8 executes?  Per loop iteration a
// wave issues 16 MFMAs (two accumulator chains, as K1e) and NV integer VALU ops after each of them
// (v_pk_min_u16 on 8 independent registers).  Prints cycles per iteration per SIMD for MFMA only,
// VALU only and both, at 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 ... && ./a.out
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <algorithm>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int NM, int NV>
__global__ void __launch_bounds__(256) k(int iters, int* out)
{
    i32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0; acc1[i] = 1; }
    i32x4 a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    uint32_t x[8], y = threadIdx.x * 2654435761u;
    for (int i = 0; i < 8; ++i) x[i] = y + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (NM) {
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x[v & 7]) : "v"(y));
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
            }
#pragma unroll
            for (int v = 0; v < (NM ? NV : 2 * NV); ++v) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x[v & 7]) : "v"(y));
        }
    }
    int s = 0;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NM, int NV>
static void run(const char* name, int cus, double clk, int wps, int iters, int* d)
{
    float best = 1e30f;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<NM, NV>), dim3(cus * wps), dim3(256), 0, 0, iters, d);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    const double cyc = best * 1e-3 * clk / iters;      // wall cycles per iteration (all resident waves progress together)
    printf("%-34s wps %d: %8.3f ms  %8.1f cycles/iteration/SIMD (= per %d wave-iterations)  mfma %d valu %d per wave-iteration\n",
           name, wps, best, cyc, wps, NM ? 16 : 0, 16 * NV);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount; const double clk = p.clockRate * 1e3;
    printf("device %s CUs=%d clk=%.0f MHz\n", p.gcnArchName, cus, clk / 1e6);
    int* d; CHECK(hipMalloc(&d, sizeof(int) * 256 * cus * 4));
    for (int wps = 1; wps <= 2; ++wps) {
        run<1, 0>("mfma only (16)", cus, clk, wps, iters, d);
        run<0, 6>("valu only (96)", cus, clk, wps, iters, d);
        run<1, 6>("mfma 16 + valu 96 interleaved", cus, clk, wps, iters, d);
        run<0, 13>("valu only (208)", cus, clk, wps, iters, d);
        run<1, 13>("mfma 16 + valu 208 interleaved", cus, clk, wps, iters, d);
    }
    return 0;
}
