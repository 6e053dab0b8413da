// K12/K13 -- representative ("median") descriptor of every landmark, batched.
//
// Reference: MapPoint::updateAverageDescDir, src/mapFeatures.cpp:51-81, and the identical
// MapLine::updateAverageDescDir, :121-157 (descriptor part; the direction average :86-91 sums into an
// uninitialised vector upstream and is not reproduced):
//   conf(i,j) = Hamming(desc_i, desc_j), conf(i,i) = 0                       :57-67
//   per row i: sort the n entries, take element int(1+0.5*(n-1))            :74-77
//   the row with the smallest such value wins, strict '<' => first row wins  :78-82
//   med_desc = desc_list[winner]                                            :84
// The reference runs this per landmark each time an observation is added (:47); here all landmarks
// of a map go in one launch, the observation lists concatenated CSR-style, and the result rows are
// directly the query matrix of the map<->keyframe matcher (mapHandler.cpp:545-561).
//
// K12 k_row_medians: one lane per (landmark, observation) = one row of conf.  The k-th order
//   statistic of the row is found by bisection on the value range [0,256] (lists of up to 16
//   observations keep the row in a lane-private LDS column; longer ones recompute the distances
//   per probe: 9 x n XOR+popcount distances per lane, rows come from L1/L2) and the
//   row's key (value << 23 | i) is folded into the landmark's slot with atomicMin -- min over keys
//   is exactly "smallest value, then smallest i", independent of execution order.
// K13 k_select_median: decodes the slot in place (idempotent, so no ordering between the lanes that
//   read it is needed) and gathers the winner's 32 bytes.
#include "common.hpp"

namespace plslam {
namespace {

constexpr uint32_t IDX_BITS = 23, IDX_MASK = (1u << IDX_BITS) - 1u;
constexpr int ROW_CACHE = 16;      // lists up to this long keep their distance row in LDS

__device__ __forceinline__ int hamming256(const uint32_t (&q)[8], const uint32_t* __restrict__ t)
{
    int d = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) d += __popc(q[w] ^ t[w]);
    return d;
}

__global__ __launch_bounds__(256) void k_row_medians(const uint32_t* __restrict__ desc,
                                                     const int32_t* __restrict__ off, int32_t n_lm,
                                                     int32_t total, uint32_t* __restrict__ slot)
{
    __shared__ uint16_t row_cache[ROW_CACHE][256];      // 8 KB
    const int32_t t = (int32_t)(blockIdx.x * 256u + threadIdx.x);
    if (t >= total) return;
    // landmark of observation t: the last l with off[l] <= t (skips empty lists)
    int32_t lo = 0, hi = n_lm;
    while (hi - lo > 1) {
        const int32_t mid = (lo + hi) >> 1;
        if (off[mid] <= t) lo = mid; else hi = mid;
    }
    const int32_t s = off[lo], n = off[lo + 1] - s;
    if (n <= 1) {                       // a single observation is its own representative (ctor :28-38)
        slot[lo] = 0u;
        return;
    }
    const int32_t k = 1 + ((n - 1) >> 1);          // == int(1 + 0.5*(n-1)) for n >= 1
    uint32_t q[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) q[w] = desc[(int64_t)t * 8 + w];
    const uint32_t* rows = desc + (int64_t)s * 8;
    // smallest v with #{j : d(i,j) <= v} >= k+1  ==  sorted(row)[k]
    int32_t vlo = 0, vhi = 256;
    if (n <= ROW_CACHE) {
        // the usual case: the row of conf lives in this lane's LDS column (lane-private, no barrier)
        for (int32_t j = 0; j < n; ++j) row_cache[j][threadIdx.x] = (uint16_t)hamming256(q, rows + (int64_t)j * 8);
        while (vlo < vhi) {
            const int32_t mid = (vlo + vhi) >> 1;
            int32_t cnt = 0;
            for (int32_t j = 0; j < n; ++j) cnt += (int32_t)row_cache[j][threadIdx.x] <= mid;
            if (cnt >= k + 1) vhi = mid; else vlo = mid + 1;
        }
    } else {
        // long lists: recompute the distances per probe instead of storing n of them
        while (vlo < vhi) {
            const int32_t mid = (vlo + vhi) >> 1;
            int32_t cnt = 0;
            for (int32_t j = 0; j < n; ++j) cnt += hamming256(q, rows + (int64_t)j * 8) <= mid;
            if (cnt >= k + 1) vhi = mid; else vlo = mid + 1;
        }
    }
    atomicMin(slot + lo, ((uint32_t)vlo << IDX_BITS) | (uint32_t)(t - s));
}

__global__ __launch_bounds__(256) void k_select_median(const uint32_t* __restrict__ desc,
                                                       const int32_t* __restrict__ off, int32_t n_lm,
                                                       int32_t* __restrict__ slot,
                                                       uint32_t* __restrict__ med_desc)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (landmark, word)
    if (t >= (int64_t)n_lm * 8) return;
    const int32_t l = (int32_t)(t >> 3), w = (int32_t)(t & 7);
    const int32_t s = off[l], n = off[l + 1] - s;
    // key -> index; applying it to an already decoded slot gives the same value
    const int32_t idx = n > 0 ? (int32_t)((uint32_t)slot[l] & IDX_MASK) : -1;
    if (w == 0) slot[l] = idx;
    if (med_desc) med_desc[t] = n > 0 ? desc[((int64_t)s + idx) * 8 + w] : 0u;
}

}  // namespace

int launch_median_desc(const uint8_t* desc, const int32_t* off, int32_t n_lm, int32_t total,
                       int32_t* med_idx, uint8_t* med_desc, hipStream_t s)
{
    if (n_lm <= 0) return PLSLAM_OK;
    PLSLAM_HIP_CHECK(hipMemsetAsync(med_idx, 0xFF, (size_t)n_lm * 4, s));
    if (total > 0) {
        hipLaunchKernelGGL(k_row_medians, dim3((unsigned)(((int64_t)total + 255) / 256)), dim3(256), 0, s,
                           reinterpret_cast<const uint32_t*>(desc), off, n_lm, total,
                           reinterpret_cast<uint32_t*>(med_idx));
        PLSLAM_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL(k_select_median, dim3((unsigned)(((int64_t)n_lm * 8 + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint32_t*>(desc), off, n_lm, med_idx,
                       reinterpret_cast<uint32_t*>(med_desc));
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

}  // namespace plslam
