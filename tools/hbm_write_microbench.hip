// What does HBM take when a kernel mostly WRITES?  The LBA row kernels (K3 / K4: k_point_rows, k_line_rows) write 3/4 of their
// bytes -- 88 of 112 B per point row -- and reach 0.64 - 0.67 of the 8 TB/s the data sheet gives; this measures the same byte
// mix with NO arithmetic: each workgroup reads R bytes per 1024 B it writes (linear, 16 B per lane per instruction, nontemporal
// or plain), 1.43 GB per launch like the K3 stream of bench.py.  Prints GB/s (read + written bytes) per variant.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_write_microbench.hip -o build/hbm_write_microbench && build/hbm_write_microbench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// a workgroup of 256 lanes per 16 KB of output (4 x 16 B per lane x 4 rounds); READ16 sixteen-byte loads per lane per 4 stores
template <int READ16, bool NT>
__global__ void __launch_bounds__(256) k(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n16)
{
    const size_t base = (size_t)blockIdx.x * 1024 + threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < READ16; ++r) {
        const size_t i = (size_t)blockIdx.x * 256 * READ16 + r * 256 + threadIdx.x;
        acc += NT ? __builtin_nontemporal_load(in + i) : in[i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const size_t i = base + r * 256;
        if (i < n16) {
            if (NT) __builtin_nontemporal_store(acc, out + i);
            else out[i] = acc;
        }
    }
}

template <int READ16, bool NT>
static void run(const char* name, const f32x4* in, f32x4* out, size_t bytes_out)
{
    const size_t n16 = bytes_out / 16;
    const unsigned nb = (unsigned)((n16 + 1023) / 1024);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k<READ16, NT>), dim3(nb), dim3(256), 0, 0, in, out, n16);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, (double)ms / 10);
    }
    const double total = (double)bytes_out * (1.0 + READ16 / 4.0);
    printf("%-44s %7.3f ms per launch  %7.1f GB/s  (%.3f of 8000)\n", name, best, total / best / 1e6, total / best / 1e6 / 8000.0);
}

int main()
{
    const size_t bytes_out = (size_t)1072 << 20;              // 1.07 GB written (+ 1/3 of it read = 1.43 GB: K3's stream)
    f32x4 *in, *out;
    CHECK(hipMalloc(&in, bytes_out)); CHECK(hipMalloc(&out, bytes_out));
    CHECK(hipMemset(in, 0, bytes_out)); CHECK(hipMemset(out, 0, bytes_out));
    run<0, false>("write only, plain stores", in, out, bytes_out);
    run<0, true>("write only, nontemporal stores", in, out, bytes_out);
    run<1, false>("read 1/4 + write (K3 is 24 B in, 88 B out)", in, out, bytes_out);
    run<1, true>("read 1/4 + write, nontemporal both", in, out, bytes_out);
    run<4, true>("copy (read 1 : write 1), nontemporal", in, out, bytes_out);
    run<4, false>("copy (read 1 : write 1), plain", in, out, bytes_out);
    return 0;
}
