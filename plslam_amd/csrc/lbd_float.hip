// K18 -- the LBD float descriptor of a line: BinaryDescriptor::computeLBD,
// 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:1026-1372 (the producer of the 72 floats that K11 turns
// into the 256-bit rows StVO::match consumes for lines).  Upstream, per line: a support region of 9 bands x
// widthOfBand (7) rows along the line; every row sums the image gradient projected on the line direction dL and on its
// normal dO, split by sign (:1140-1187); rows are weighted with the global Gaussian gaussCoefG_ (:1188-1196) and added
// to their own band and its two neighbours with the local weights gaussCoefL_ (:1201-1239); per band mean and standard
// deviation of the four sums (:1253-1277), normalisation of the means and of the deviations (:1279-1312), clamp at 0.4
// (:1318-1325), re-normalisation (:1327-1338).  All in fp32 with SEQUENTIAL sums.
//
// Mapping (round 2): ONE WORKGROUP OF FOUR WAVES PER LINE; the rows of the support region are the lanes of every wave.
// What is sequential by contract -- the fp32 row sums must see their terms in pixel order, and a row's pixel coordinates
// are a chain of fp32 additions -- is kept sequential; everything around it is spread:
//   * gather: the pixels of a row are cut into rounds of 32; in a round each wave visits 8 pixels of its lane's row
//     (nearest-pixel cell, border clamp, the two 2-byte gradient reads, projection on the line direction and its normal)
//     and parks the two projections in LDS.  A wave reaches its pixels by running the coordinate chain through the
//     pixels of the other waves too (two additions per pixel, no memory).  The 16 reads of a round are independent and in
//     flight together -- round 1's single wave paid one memory latency per pixel.
//   * sums: wave q owns ONE of the four sign-split sums (positive / negative part of either projection) of every row and
//     adds the round's 32 terms in pixel order from LDS while the next round's reads are in flight (two LDS buffers, one
//     workgroup barrier per round).
//   * bands: 72 threads = 9 bands x 8 quantities, each ONE chain over the <= 3 x widthOfBand rows that feed its band, in
//     row order (round 1: 9 lanes x 8 chains x 63 rows).
//   * normalisation: the two 36-term chains in two lanes, the clamp and the scaling across 72 threads, the final
//     72-term chain in one lane.
// Built with -ffp-contract=off like the checker: no FMA, same operation order => the result is the checker's bit for
// bit (the reference binary itself is built -O3 -march=native and may contract, so the CONTRACT is 1e-5 relative).
// cos / sin of the direction and the two Gaussian tables come from the host (libm), as upstream's come from its
// constructor.  The line records are read by the kernel straight from page-locked host memory (a ring of slots in the
// context): no copy, no synchronisation inside the call.
// Work per line = rows x numOfPixels pixel visits, 4 B gathered per visit (L1 / L2 hits: neighbouring rows share cache
// lines), 288 B out.
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace plslam {
namespace {

constexpr int LBD_BANDS = 9;
constexpr int LBD_MAX_W = 7;          // widthOfBand: 9 * 7 = 63 rows <= one wave
constexpr int LBD_WAVES = 4;          // waves per line
constexpr int LBD_CHUNK = 8;          // pixels of a row one wave visits per round
constexpr int LBD_ROUND = LBD_WAVES * LBD_CHUNK;
constexpr int LBD_DIM = LBD_BANDS * 8;

struct LbdLineDev {                   // plslam_lbd_line + the host-evaluated direction cosines
    int32_t num_pixels;
    float sx, sy, ex, ey;
    float dl0, dl1;
    float pad;
};

struct LbdTables {                    // (float) of the constructor's double tables
    float coef_l[3 * LBD_MAX_W];
    float coef_g[LBD_BANDS * LBD_MAX_W];
};

// nearest pixel as the source takes it: round half away from zero (exact here: v - trunc(v) has no rounding error in
// fp32), through a 16-bit integer, clamped to [0, last]
__device__ __forceinline__ int nearest_cell(float v, int last)
{
    const float whole = truncf(v);
    const float nearest = fabsf(v - whole) >= 0.5f ? whole + copysignf(1.0f, v) : whole;
    const int as16 = (int)(short)(int)nearest;
    return min(max(as16, 0), last);
}

__global__ void __launch_bounds__(64 * LBD_WAVES)
k_lbd_rows(const int16_t* __restrict__ grad_x, const int16_t* __restrict__ grad_y, int32_t width, int32_t height,
           const LbdLineDev* __restrict__ lines, int32_t band_w, LbdTables tab, float* __restrict__ lbd)
{
    __shared__ float s_proj[2][2][LBD_ROUND][64];        // [buffer][along | across][pixel of the round][row]
    __shared__ float s_rows[8][64];                      // the weighted row sums and their squares
    __shared__ float s_band[LBD_BANDS][8];
    __shared__ float s_vec[LBD_DIM];
    __shared__ float s_scale[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, tid = threadIdx.x;
    const LbdLineDev L = lines[blockIdx.x];
    const int rows = band_w * LBD_BANDS;
    const int npix = (short)L.num_pixels;
    const float along_x = L.dl0, along_y = L.dl1;        // unit vector of the line; its normal is (-along_y, along_x)
    const float across_x = -along_y, across_y = along_x;

    // ---- where this lane's row starts: the region's corner, then one step along the normal per row -- a chain of fp32
    // additions (not a product); every lane runs the same chain and keeps the value of its own row
    float row_x, row_y;
    {
        const short half_len = (short)((npix - 1) / 2), half_rows = (short)((rows - 1) / 2);
        const float mid_x = (float)(0.5 * (L.sx + L.ex)), mid_y = (float)(0.5 * (L.sy + L.ey));
        float cx = -along_x * half_len + along_y * half_rows + mid_x;
        float cy = -along_y * half_len - along_x * half_rows + mid_y;
        row_x = cx;
        row_y = cy;
        for (int r = 1; r < rows; ++r) {
            cx -= along_y;
            cy += along_x;
            if (lane == r) {
                row_x = cx;
                row_y = cy;
            }
        }
    }

    // ---- rounds of 32 pixels: gather (8 per wave) | barrier | one sign-split sum per wave ------------------------
    float walk_x = row_x, walk_y = row_y;                // coordinate chain of this lane's row ...
    int walk_at = 0;                                     // ... standing at this pixel
    const int last_x = width - 1, last_y = height - 1;
    auto gather = [&](int round, int buf) {
        const int first = round * LBD_ROUND + wave * LBD_CHUNK;
        const int mine = min(LBD_CHUNK, npix - first);
        if (mine <= 0) return;
        for (; walk_at < first; ++walk_at) {             // the pixels the other waves visit
            walk_x += along_x;
            walk_y += along_y;
        }
        int cell[LBD_CHUNK];
#pragma unroll
        for (int k = 0; k < LBD_CHUNK; ++k) {
            cell[k] = nearest_cell(walk_y, last_y) * width + nearest_cell(walk_x, last_x);
            walk_x += along_x;
            walk_y += along_y;
        }
        walk_at += LBD_CHUNK;
        float gx[LBD_CHUNK], gy[LBD_CHUNK];
#pragma unroll
        for (int k = 0; k < LBD_CHUNK; ++k) {
            gx[k] = (float)grad_x[cell[k]];
            gy[k] = (float)grad_y[cell[k]];
        }
#pragma unroll
        for (int k = 0; k < LBD_CHUNK; ++k)
            if (k < mine) {
                s_proj[buf][0][wave * LBD_CHUNK + k][lane] = gx[k] * along_x + gy[k] * along_y;
                s_proj[buf][1][wave * LBD_CHUNK + k][lane] = gx[k] * across_x + gy[k] * across_y;
            }
    };
    // wave 0: positive part along, 1: negative part along, 2 / 3: the same across the line.  A term enters its sum only
    // on its own side of `> 0` (zero and NaN count as negative), the negative parts as acc - term = acc + (-term).
    const bool neg_part = (wave & 1) != 0;
    const int which = wave >> 1;
    float acc = 0.0f;
    auto add_round = [&](int round, int buf) {
        const int cnt = min(LBD_ROUND, npix - round * LBD_ROUND);
        const float* term = &s_proj[buf][which][0][lane];
#pragma unroll 8
        for (int k = 0; k < cnt; ++k) {
            const float t = term[k * 64];
            const float next = acc + (neg_part ? -t : t);
            acc = ((t > 0.0f) != neg_part) ? next : acc;
        }
    };
    const int nround = (npix + LBD_ROUND - 1) / LBD_ROUND;
    gather(0, 0);
    __syncthreads();
    for (int r = 0; r < nround; ++r) {
        if (r + 1 < nround) gather(r + 1, (r + 1) & 1);
        add_round(r, r & 1);
        __syncthreads();
    }
    if (lane < rows) {
        const float weighted = tab.coef_g[lane] * acc;
        const int slot = (wave & 1) + 4 * which;          // pos along, neg along, (squares), pos across, neg across, (squares)
        s_rows[slot][lane] = weighted;
        s_rows[slot + 2][lane] = weighted * weighted;
    }
    __syncthreads();

    // ---- bands: thread (band, quantity) adds the rows of bands band-1, band, band+1 in row order; the local weight of a
    // row depends only on its distance from the first row of band-1
    if (tid < LBD_DIM) {
        const int band = tid >> 3, q = tid & 7;
        const int origin = (band - 1) * band_w;
        const int from = max(origin, 0), to = min(origin + 3 * band_w, rows);
        const bool squares = (q & 2) != 0;
        float sum = 0.0f;
        for (int r = from; r < to; ++r) {
            const float c = tab.coef_l[r - origin];
            sum += (squares ? c * c : c) * s_rows[q][r];
        }
        s_band[band][q] = sum;
    }
    __syncthreads();
    // per band: the four means, then the four standard deviations
    if (tid < LBD_DIM) {
        const int band = tid >> 3, o = tid & 7;
        const float inv_n = (band == 0 || band == LBD_BANDS - 1) ? (float)(1.0 / (band_w * 2.0)) : (float)(1.0 / (band_w * 3.0));
        const int src = (o & 1) + ((o & 2) ? 4 : 0);
        const float mean = s_band[band][src] * inv_n;
        s_vec[tid] = o < 4 ? mean : sqrtf(s_band[band][src + 2] * inv_n - mean * mean);
    }
    __syncthreads();

    // ---- the 72-vector: means and deviations normalised separately, clamped at 0.4, normalised again -------------------
    if (tid < 2) {                                        // thread 0: the 36 means, thread 1: the 36 deviations, band by band
        float ss = 0.0f;
        for (int b = 0; b < LBD_BANDS; ++b)
            for (int k = 0; k < 4; ++k) {
                const float v = s_vec[b * 8 + 4 * tid + k];
                ss += v * v;
            }
        s_scale[tid] = 1 / sqrtf(ss);
    }
    __syncthreads();
    if (tid < LBD_DIM) {
        float v = s_vec[tid] * s_scale[(tid >> 2) & 1];
        if ((double)v > 0.4) v = (float)0.4;
        s_vec[tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float ss = 0.0f;
        for (int i = 0; i < LBD_DIM; ++i) ss += s_vec[i] * s_vec[i];
        s_scale[2] = 1 / sqrtf(ss);
    }
    __syncthreads();
    if (tid < LBD_DIM) lbd[(size_t)blockIdx.x * LBD_DIM + tid] = s_vec[tid] * s_scale[2];
}

LbdTables make_tables(int32_t w)
{
    // the constructor of BinaryDescriptor, :146-176 (integer divisions as written)
    LbdTables t;
    memset(&t, 0, sizeof(t));
    double u = (w * 3 - 1) / 2;
    double sigma = (w * 2 + 1) / 2;
    double invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < w * 3; i++) {
        const double dis = i - u;
        t.coef_l[i] = (float)std::exp(dis * dis * invsigma2);
    }
    u = (LBD_BANDS * w - 1) / 2;
    sigma = u;
    invsigma2 = -1 / (2 * sigma * sigma);
    for (int i = 0; i < LBD_BANDS * w; i++) {
        const double dis = i - u;
        t.coef_g[i] = (float)std::exp(dis * dis * invsigma2);
    }
    return t;
}

int lbd_check(int32_t width, int32_t height, int32_t n, int32_t w)
{
    PLSLAM_REQUIRE(n >= 0 && width >= 1 && height >= 1 && width <= 32767 && height <= 32767, PLSLAM_EINVAL);
    PLSLAM_REQUIRE((int64_t)width * height <= 32767LL * 32767LL, PLSLAM_ERANGE);
    PLSLAM_REQUIRE(w >= 1 && w <= LBD_MAX_W, PLSLAM_ENOTSUP);
    return PLSLAM_OK;
}

int launch_lbd_compute(const int16_t* dx, const int16_t* dy, int32_t width, int32_t height, const LbdLineDev* d_lines,
                       int32_t n, int32_t w, float* lbd, hipStream_t s)
{
    if (n <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_lbd_rows, dim3((unsigned)n), dim3(64 * LBD_WAVES), 0, s, dx, dy, width, height, d_lines, w,
                       make_tables(w), lbd);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

// The line records of a call (the detector's output, host data) go into the next slot of the context's ring of
// page-locked buffers; the kernel reads them through the slot's device address.  A slot is rewritten only after the
// kernel that read it has finished (its event; four calls later, so that wait is normally over before it starts).
int stage_line_records(plslam_ctx* ctx, const plslam_lbd_line* lines, int32_t n, int* slot, const LbdLineDev** d_lines)
{
    LineRing& ring = ctx->lbd_ring;
    const int k = ring.next;
    ring.next = (k + 1) % LineRing::SLOTS;
    if (ring.busy[k]) {
        PLSLAM_HIP_CHECK(hipEventSynchronize(ring.done[k]));
        ring.busy[k] = false;
    }
    if (!ring.done[k]) PLSLAM_HIP_CHECK(hipEventCreateWithFlags(&ring.done[k], hipEventDisableTiming));
    int rc;
    if ((rc = ring.rec[k].reserve((size_t)n * sizeof(LbdLineDev)))) return rc;
    LbdLineDev* out = ring.rec[k].as<LbdLineDev>();
    for (int32_t i = 0; i < n; ++i) {
        out[i].num_pixels = lines[i].num_pixels;
        out[i].sx = lines[i].sx; out[i].sy = lines[i].sy; out[i].ex = lines[i].ex; out[i].ey = lines[i].ey;
        // dL[0] = cos(direction), :1117-1118 -- the FLOAT overload (cosf), as the reference binds it under libstdc++
        out[i].dl0 = std::cos(lines[i].direction);
        out[i].dl1 = std::sin(lines[i].direction);
        out[i].pad = 0.f;
    }
    void* dev = mapped_device_pointer(out);
    if (!dev) {
        set_last_error("%s:%d: page-locked line records have no device address", __FILE__, __LINE__);
        return PLSLAM_EHIP;
    }
    *slot = k;
    *d_lines = static_cast<const LbdLineDev*>(dev);
    return PLSLAM_OK;
}

int lbd_enqueue(plslam_ctx* ctx, const int16_t* dx, const int16_t* dy, int32_t width, int32_t height,
                const plslam_lbd_line* lines, int32_t n, int32_t w, float* lbd, hipStream_t s)
{
    int slot = 0, rc;
    const LbdLineDev* d_lines = nullptr;
    if ((rc = stage_line_records(ctx, lines, n, &slot, &d_lines))) return rc;
    if ((rc = launch_lbd_compute(dx, dy, width, height, d_lines, n, w, lbd, s))) return rc;
    PLSLAM_HIP_CHECK(hipEventRecord(ctx->lbd_ring.done[slot], s));
    ctx->lbd_ring.busy[slot] = true;
    return PLSLAM_OK;
}

}  // namespace
}  // namespace plslam

extern "C" {

int plslam_lbd_compute(plslam_ctx* ctx, const int16_t* dx_img, const int16_t* dy_img, int32_t width, int32_t height,
                       const plslam_lbd_line* lines, int32_t n, int32_t width_of_band, float* lbd_f32)
{
    using namespace plslam;
    PLSLAM_REQUIRE(ctx != nullptr, PLSLAM_EINVAL);
    int rc;
    if ((rc = lbd_check(width, height, n, width_of_band))) return rc;
    if (n == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(dx_img && dy_img && lines && lbd_f32, PLSLAM_EINVAL);
    for (int32_t i = 0; i < n; ++i) PLSLAM_REQUIRE(lines[i].num_pixels >= 0 && lines[i].num_pixels <= 32767, PLSLAM_EINVAL);
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard(ctx->device);
    const size_t img = (size_t)width * height * 2;
    Carver ci;
    const size_t oX = ci.take(img), oY = ci.take(img);
    const size_t out_bytes = (size_t)n * 72 * 4;
    if ((rc = ctx->in_a.reserve(ci.off))) return rc;
    if ((rc = ctx->out_a.reserve(out_bytes))) return rc;
    char* d = ctx->in_a.as<char>();
    hipStream_t s = ctx->stream;
    StreamSyncOnError sg(s);
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oX, dx_img, img, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oY, dy_img, img, hipMemcpyHostToDevice, s));
    if ((rc = lbd_enqueue(ctx, (const int16_t*)(d + oX), (const int16_t*)(d + oY), width, height, lines, n, width_of_band,
                          ctx->out_a.as<float>(), s)))
        return rc;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(lbd_f32, ctx->out_a.p, out_bytes, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    sg.dismiss();
    return PLSLAM_OK;
}

int plslam_lbd_compute_dev(plslam_ctx* ctx, const int16_t* dx_img, const int16_t* dy_img, int32_t width, int32_t height,
                           const plslam_lbd_line* lines_host, int32_t n, int32_t width_of_band, float* lbd_f32,
                           void* stream)
{
    using namespace plslam;
    PLSLAM_REQUIRE(ctx != nullptr, PLSLAM_EINVAL);
    int rc;
    if ((rc = lbd_check(width, height, n, width_of_band))) return rc;
    if (n == 0) return PLSLAM_OK;
    PLSLAM_REQUIRE(dx_img && dy_img && lines_host && lbd_f32, PLSLAM_EINVAL);
    for (int32_t i = 0; i < n; ++i) PLSLAM_REQUIRE(lines_host[i].num_pixels >= 0 && lines_host[i].num_pixels <= 32767, PLSLAM_EINVAL);
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard guard(ctx->device);
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : ctx->stream;
    // asynchronous: the records sit in a page-locked slot of the context until the kernel has read them
    return lbd_enqueue(ctx, dx_img, dy_img, width, height, lines_host, n, width_of_band, lbd_f32, s);
}

}  // extern "C"
