// v_pk_minimum3_f16 / v_pk_maximum3_f16 on 16-bit INTEGER keys below 0x7C00 (all positive finite half floats, denormals
// included): is the result the integer minimum / maximum for every operand combination, and what does the instruction cost
// beside v_pk_min_u16?   hipcc --offload-arch=gfx950 -O3 tools/pk_min3_f16_check.hip -o build/pk_min3_f16_check
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_check(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* mn, uint32_t* mx, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r, s;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    asm("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(s) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    mn[i] = r;
    mx[i] = s;
}

// round 6: the same instruction with op_sel -- low result = min3(a.lo, b.lo, c.HI), high result = min3(a.hi, b.HI, c.lo): with
// b, c two UNPACKED accumulators (float bits 0x4B00kkkk: key in the low half, 0x4B00 in the high half) and a the packed running
// minima of two rows, ONE instruction is "a.lo = min(a.lo, key(b)), a.hi = min(a.hi, key(c))", capped at 0x4B00 -- no v_perm
__global__ void k_check_opsel(const uint32_t* a, const uint32_t* b, const uint32_t* c, uint32_t* mn, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t r;
    asm("v_pk_minimum3_f16 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    mn[i] = r;
}

template <int KIND>      // 0: v_pk_min_u16 (2 inputs), 1: v_pk_minimum3_f16 (3 inputs), 2: v_min3_u32, 3: v_perm_b32, 4: pk_minimum3 with op_sel, 5: v_min3_f32
__global__ void __launch_bounds__(256) k_rate(int iters, unsigned long long* out, uint32_t* sink)
{
    uint32_t x[8], y = (threadIdx.x * 2654435761u) & 0x3FFF3FFFu, z = (y >> 1) | 0x04000400u;
    for (int i = 0; i < 8; ++i) x[i] = (y + 0x00010001u * i) & 0x3FFF3FFFu;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int v = 0; v < 104; ++v) {
            if (KIND == 0) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x[v & 7]) : "v"(y));
            if (KIND == 1) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(y), "v"(z));
            if (KIND == 2) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(y), "v"(z));
            if (KIND == 3) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(y), "v"(z));
            if (KIND == 4) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2 op_sel:[0,0,1] op_sel_hi:[1,1,0]" : "+v"(x[v & 7]) : "v"(y), "v"(z));
            if (KIND == 5) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x[v & 7]) : "v"(y | 0x4B000000u), "v"(z | 0x4B000000u));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r += x[i];
    sink[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int KIND>
static void rate(const char* name, int cus, unsigned long long* d_out, uint32_t* d_sink)
{
    const int iters = 2000;
    printf("%-22s", name);
    for (int wps = 1; wps <= 3; ++wps) {
        const int nb = cus * wps;
        double best = 1e30, bms = 1e30;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL((k_rate<KIND>), dim3(nb), dim3(256), 0, 0, iters, d_out, d_sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(nb);
            CHECK(hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * nb, hipMemcpyDeviceToHost));
            double sum = 0;
            for (auto v : h) sum += (double)v;
            best = std::min(best, sum / nb / iters / 104.0);
            bms = std::min(bms, (double)ms);
        }
        printf("  wps%d: %5.2f ticks/instr/wave = %5.2f per SIMD, %6.3f ms (%5.2f ns/instr/SIMD)", wps, best, best / wps, bms, bms * 1e6 / (iters * 104.0 * wps));
    }
    printf("\n");
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    // every pair of 16-bit keys below 0x7C00 would be 2^30 cases; the third operand makes it 2^45: sample -- all (lo, hi)
    // pairs of a dense set of edge values, plus random triples
    std::vector<uint32_t> edge;
    for (uint32_t v = 0; v < 0x7C00; v += 1) if (v < 0x0440 || (v & 0x3FF) < 3 || (v & 0x3FF) > 0x3FC || v % 97 == 0) edge.push_back(v);
    std::vector<uint32_t> a, b, c;
    uint64_t st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (uint32_t)(st >> 16); };
    for (size_t i = 0; i < edge.size(); ++i)
        for (int k = 0; k < 64; ++k) {
            const uint32_t e2 = edge[rnd() % edge.size()], e3 = edge[rnd() % edge.size()];
            const uint32_t r1 = rnd() % 0x7C00, r2 = rnd() % 0x7C00, r3 = rnd() % 0x7C00;
            a.push_back(edge[i] | (r1 << 16)); b.push_back(e2 | (r2 << 16)); c.push_back(e3 | (r3 << 16));
            a.push_back(r1 | (edge[i] << 16)); b.push_back(r2 | (e3 << 16)); c.push_back(r3 | (e2 << 16));
        }
    const int n = (int)a.size();
    uint32_t *da, *db, *dc, *dmn, *dmx;
    CHECK(hipMalloc(&da, 4 * n)); CHECK(hipMalloc(&db, 4 * n)); CHECK(hipMalloc(&dc, 4 * n)); CHECK(hipMalloc(&dmn, 4 * n)); CHECK(hipMalloc(&dmx, 4 * n));
    CHECK(hipMemcpy(da, a.data(), 4 * n, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(db, b.data(), 4 * n, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dc, c.data(), 4 * n, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_check, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dc, dmn, dmx, n);
    std::vector<uint32_t> mn(n), mx(n);
    CHECK(hipMemcpy(mn.data(), dmn, 4 * n, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(mx.data(), dmx, 4 * n, hipMemcpyDeviceToHost));
    long bad_min = 0, bad_max = 0;
    for (int i = 0; i < n; ++i) {
        auto lo = [](uint32_t v) { return v & 0xFFFFu; };
        auto hi = [](uint32_t v) { return v >> 16; };
        const uint32_t emn = std::min({lo(a[i]), lo(b[i]), lo(c[i])}) | (std::min({hi(a[i]), hi(b[i]), hi(c[i])}) << 16);
        const uint32_t emx = std::max({lo(a[i]), lo(b[i]), lo(c[i])}) | (std::max({hi(a[i]), hi(b[i]), hi(c[i])}) << 16);
        if (mn[i] != emn) { if (bad_min < 5) printf("min3 %08x %08x %08x -> %08x expected %08x\n", a[i], b[i], c[i], mn[i], emn); ++bad_min; }
        if (mx[i] != emx) { if (bad_max < 5) printf("max3 %08x %08x %08x -> %08x expected %08x\n", a[i], b[i], c[i], mx[i], emx); ++bad_max; }
    }
    printf("v_pk_minimum3_f16 / v_pk_maximum3_f16 as integer min / max over %d packed triples of keys < 0x7C00: %ld / %ld wrong\n", n, bad_min, bad_max);
    {
        // op_sel form: b, c as accumulators (0x4B00 in the high half), a any packed pair of keys
        std::vector<uint32_t> b2(n), c2(n);
        for (int i = 0; i < n; ++i) { b2[i] = 0x4B000000u | (b[i] & 0xFFFFu); c2[i] = 0x4B000000u | (c[i] & 0xFFFFu); }
        CHECK(hipMemcpy(db, b2.data(), 4 * n, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dc, c2.data(), 4 * n, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_check_opsel, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, dc, dmn, n);
        CHECK(hipMemcpy(mn.data(), dmn, 4 * n, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int i = 0; i < n; ++i) {
            const uint32_t e = std::min({a[i] & 0xFFFFu, b2[i] & 0xFFFFu, 0x4B00u}) | (std::min({a[i] >> 16, 0x4B00u, c2[i] & 0xFFFFu}) << 16);
            if (mn[i] != e) { if (bad < 5) printf("opsel %08x %08x %08x -> %08x expected %08x\n", a[i], b2[i], c2[i], mn[i], e); ++bad; }
        }
        printf("v_pk_minimum3_f16 op_sel:[0,0,1] op_sel_hi:[1,1,0] on (packed minima, accumulator, accumulator) = (min(a.lo, b.lo), min(a.hi, c.lo)) capped at 0x4B00, %d triples: %ld wrong\n", n, bad);
    }
    unsigned long long* d_out; uint32_t* d_sink;
    CHECK(hipMalloc(&d_out, 8 * 4096)); CHECK(hipMalloc(&d_sink, 4 * 256 * 4096));
    rate<0>("v_pk_min_u16", p.multiProcessorCount, d_out, d_sink);
    rate<1>("v_pk_minimum3_f16", p.multiProcessorCount, d_out, d_sink);
    rate<2>("v_min3_u32", p.multiProcessorCount, d_out, d_sink);
    rate<3>("v_perm_b32", p.multiProcessorCount, d_out, d_sink);
    rate<4>("v_pk_minimum3_f16 op_sel", p.multiProcessorCount, d_out, d_sink);
    rate<5>("v_min3_f32", p.multiProcessorCount, d_out, d_sink);
    return 0;
}
