// What does rocprofv3's FETCH_SIZE report for GATHERS?  MI355X_MICROARCH.md: this rocprofv3 reports HALF the bytes of a wide
// coalesced streaming read and calls other access patterns uncalibrated; round 3 calibrated dword-per-lane streaming (factor 2).
// The finalize kernel's column-key reads are 8-byte gathers (k_finalize, hamming.hip): this launches kernels whose fetched LINES
// are known by construction, one launch each, over a table no cache has seen --
//   stream16   16 B per lane, contiguous                     bytes = n x 16, every byte used
//   stream8     8 B per lane, contiguous                     bytes = n x 8
//   gather8/64  8 B per lane at byte offset 64 i             one 8-byte word of every 64-byte half line
//   gather8/128 8 B per lane at byte offset 128 i            one 8-byte word of every 128-byte line
//   gather8/256 8 B per lane at byte offset 256 i            one word of every SECOND 128-byte line
//   gather8/rnd 8 B per lane at a random 8-byte slot of a 12 KB table per 1500 lanes (the finalize pattern: a problem's column table)
// -- and prints what each should move; run it under `rocprofv3 --pmc FETCH_SIZE` and compare per kernel (tools/rocpd_summary.py).
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_gather_calib.hip -o /tmp/fetch_gather_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_stream16(const u32x4* __restrict__ t, uint32_t* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32x4 v = t[i];
    if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345u) out[0] = 1;
}
__global__ void __launch_bounds__(256) k_stream8(const u32x2* __restrict__ t, uint32_t* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32x2 v = t[i];
    if ((v.x ^ v.y) == 0x12345u) out[0] = 1;
}
template <int STRIDE>
__global__ void __launch_bounds__(256) k_gather8(const char* __restrict__ t, uint32_t* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32x2 v = *reinterpret_cast<const u32x2*>(t + i * STRIDE);
    if ((v.x ^ v.y) == 0x12345u) out[0] = 1;
}
// lane i gathers slot idx[i] (< 1500) of table (i / 1500): 12 000-byte tables, every one read by 1500 lanes
__global__ void __launch_bounds__(256) k_gather8_rnd(const u32x2* __restrict__ t, const uint16_t* __restrict__ idx, uint32_t* __restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const u32x2 v = t[(i / 1500) * 1500 + idx[i]];
    if ((v.x ^ v.y) == 0x12345u) out[0] = 1;
}

int main()
{
    const size_t n = (size_t)1 << 22;                          // 4 M lanes per launch
    const size_t bytes = n * 256 + 4096;                       // the widest stride's table: 1 GB
    char* t;
    uint32_t* out;
    uint16_t* idx;
    CHECK(hipMalloc(&t, bytes)); CHECK(hipMalloc(&out, 64)); CHECK(hipMalloc(&idx, n * 2));
    CHECK(hipMemset(t, 0, bytes)); CHECK(hipMemset(out, 0, 64));
    std::vector<uint16_t> h(n);
    uint32_t s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (uint16_t)((s >> 8) % 1500u); }
    CHECK(hipMemcpy(idx, h.data(), n * 2, hipMemcpyHostToDevice));
    CHECK(hipDeviceSynchronize());
    const unsigned nb = (unsigned)(n / 256);
    // between the launches a 1 GB memset pushes everything out of L2 and the memory-side cache
    auto flush = [&]() { CHECK(hipMemset(t + bytes / 2, 0, bytes / 2)); CHECK(hipMemset(t, 0, bytes / 2)); CHECK(hipDeviceSynchronize()); };
    flush(); hipLaunchKernelGGL(k_stream16, dim3(nb), dim3(256), 0, 0, (const u32x4*)t, out, n); CHECK(hipDeviceSynchronize());
    flush(); hipLaunchKernelGGL(k_stream8, dim3(nb), dim3(256), 0, 0, (const u32x2*)t, out, n); CHECK(hipDeviceSynchronize());
    flush(); hipLaunchKernelGGL((k_gather8<64>), dim3(nb), dim3(256), 0, 0, (const char*)t, out, n); CHECK(hipDeviceSynchronize());
    flush(); hipLaunchKernelGGL((k_gather8<128>), dim3(nb), dim3(256), 0, 0, (const char*)t, out, n); CHECK(hipDeviceSynchronize());
    flush(); hipLaunchKernelGGL((k_gather8<256>), dim3(nb), dim3(256), 0, 0, (const char*)t, out, n); CHECK(hipDeviceSynchronize());
    flush(); hipLaunchKernelGGL(k_gather8_rnd, dim3(nb), dim3(256), 0, 0, (const u32x2*)t, idx, out, n); CHECK(hipDeviceSynchronize());
    const double MiB = 1024.0 * 1024.0;
    printf("lanes per launch: %zu\n", n);
    printf("k_stream16       used %8.1f MiB  lines touched x 128 B %8.1f MiB\n", n * 16 / MiB, n * 16 / MiB);
    printf("k_stream8        used %8.1f MiB  lines touched x 128 B %8.1f MiB\n", n * 8 / MiB, n * 8 / MiB);
    printf("k_gather8<64>    used %8.1f MiB  half lines x 64 B %8.1f MiB  lines x 128 B %8.1f MiB\n", n * 8 / MiB, n * 64 / MiB, n * 64 / MiB);
    printf("k_gather8<128>   used %8.1f MiB  half lines x 64 B %8.1f MiB  lines x 128 B %8.1f MiB\n", n * 8 / MiB, n * 64 / MiB, n * 128 / MiB);
    printf("k_gather8<256>   used %8.1f MiB  half lines x 64 B %8.1f MiB  lines x 128 B %8.1f MiB\n", n * 8 / MiB, n * 64 / MiB, n * 128 / MiB);
    printf("k_gather8_rnd    used %8.1f MiB  tables %8.1f MiB (each read once per XCD that runs one of its 5.9 workgroups) + indices %6.1f MiB\n",
           n * 8 / MiB, n * 8 / MiB, n * 2 / MiB);
    return 0;
}
