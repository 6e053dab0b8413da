// K18 -- the LBD float descriptor of a line: BinaryDescriptor::computeLBD,
// 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:1026-1372 (the producer of the 72 floats that K11 turns
// into the 256-bit rows StVO::match consumes for lines).  Upstream, per line: a support region of 9 bands x
// widthOfBand (7) rows along the line; every row sums the image gradient projected on the line direction dL and on its
// normal dO, split by sign (:1140-1187); rows are weighted with the global Gaussian gaussCoefG_ (:1188-1196) and added
// to their own band and its two neighbours with the local weights gaussCoefL_ (:1201-1239); per band mean and standard
// deviation of the four sums (:1253-1277), normalisation of the means and of the deviations (:1279-1312), clamp at 0.4
// (:1318-1325), re-normalisation (:1327-1338).  All in fp32 with SEQUENTIAL sums.
//
// Mapping: ONE WAVE PER LINE -- the 63 rows of the region are the lanes.  Lane h walks its row pixel by pixel in the
// source's order (row start by h repeated float subtractions, `sCorX += dL[0]` per pixel, nearest-pixel rounding,
// clamping to the image), so the four row sums are the same fp32 sequences as upstream's; the rows meet in LDS; lanes
// 0..8 each build one band by adding its up to 3 x widthOfBand contributions in row order and derive the band's eight
// values; one lane finishes the 72-vector with the source's two normalisation loops.  Built with -ffp-contract=off like
// the checker: no FMA, same operation order => the result is the checker's bit for bit (the reference binary itself is
// built -O3 -march=native and may contract, so the CONTRACT is 1e-5 relative).  cos / sin of the direction and the two
// Gaussian tables come from the host (libm), as upstream's come from its constructor.
// The gradient reads are scattered 2-byte gathers along the line (L1 / L2 hits: neighbouring rows share cache lines);
// work per line = 63 x numOfPixels pixel visits, so a frame's 200 lines are launch-bound and a batch is gather-bound.
#include <cmath>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace plslam {
namespace {

constexpr int LBD_BANDS = 9;
constexpr int LBD_MAX_W = 7;          // widthOfBand: 9 * 7 = 63 rows <= one wave
constexpr int LBD_LINES_PER_WG = 4;

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

__global__ void __launch_bounds__(64 * LBD_LINES_PER_WG)
k_lbd_compute(const int16_t* __restrict__ pdxImg, const int16_t* __restrict__ pdyImg, int32_t width, int32_t height,
              const LbdLineDev* __restrict__ lines, int32_t n, int32_t w, LbdTables tab, float* __restrict__ lbd)
{
    __shared__ float s_row[LBD_LINES_PER_WG][8][64];     // per line: 8 row quantities x 63 rows
    __shared__ float s_des[LBD_LINES_PER_WG][LBD_BANDS * 8];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = blockIdx.x * LBD_LINES_PER_WG + wv;
    if (li >= n) return;                                  // wave-uniform; no workgroup barriers below
    const LbdLineDev L = lines[li];
    const short heightOfLSP = (short)(w * LBD_BANDS);
    const short halfHeight = (heightOfLSP - 1) / 2;
    const short lengthOfLSP = (short)L.num_pixels;
    const short halfWidth = (lengthOfLSP - 1) / 2;
    const short realWidth = (short)width, imageWidth = realWidth - 1, imageHeight = (short)(height - 1);
    const float lineMiddlePointX = (float)(0.5 * (L.sx + L.ex));
    const float lineMiddlePointY = (float)(0.5 * (L.sy + L.ey));
    const float dL0 = L.dl0, dL1 = L.dl1, dO0 = -dL1, dO1 = dL0;
    if (lane < heightOfLSP) {
        float sCorX0 = -dL0 * halfWidth + dL1 * halfHeight + lineMiddlePointX;
        float sCorY0 = -dL1 * halfWidth - dL0 * halfHeight + lineMiddlePointY;
        for (int h = 0; h < lane; ++h) {                  // upstream reaches row h by h repeated updates
            sCorX0 -= dL1;
            sCorY0 += dL0;
        }
        float sCorX = sCorX0, sCorY = sCorY0;
        float pgdLRowSum = 0, ngdLRowSum = 0, pgdORowSum = 0, ngdORowSum = 0;
        for (short wID = 0; wID < lengthOfLSP; wID++) {
            short tempCor = (short)(int)round((double)sCorX);
            const short xCor = (tempCor < 0) ? (short)0 : (tempCor > imageWidth) ? imageWidth : tempCor;
            tempCor = (short)(int)round((double)sCorY);
            const short yCor = (tempCor < 0) ? (short)0 : (tempCor > imageHeight) ? imageHeight : tempCor;
            const short dx = pdxImg[yCor * realWidth + xCor];
            const short dy = pdyImg[yCor * realWidth + xCor];
            const float gDL = dx * dL0 + dy * dL1;
            const float gDO = dx * dO0 + dy * dO1;
            if (gDL > 0) pgdLRowSum += gDL; else ngdLRowSum -= gDL;
            if (gDO > 0) pgdORowSum += gDO; else ngdORowSum -= gDO;
            sCorX += dL0;
            sCorY += dL1;
        }
        const float cg = tab.coef_g[lane];
        pgdLRowSum = cg * pgdLRowSum;
        ngdLRowSum = cg * ngdLRowSum;
        pgdORowSum = cg * pgdORowSum;
        ngdORowSum = cg * ngdORowSum;
        s_row[wv][0][lane] = pgdLRowSum;
        s_row[wv][1][lane] = ngdLRowSum;
        s_row[wv][2][lane] = pgdLRowSum * pgdLRowSum;
        s_row[wv][3][lane] = ngdLRowSum * ngdLRowSum;
        s_row[wv][4][lane] = pgdORowSum;
        s_row[wv][5][lane] = ngdORowSum;
        s_row[wv][6][lane] = pgdORowSum * pgdORowSum;
        s_row[wv][7][lane] = ngdORowSum * ngdORowSum;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane < LBD_BANDS) {
        // band `lane`: contributions in row order; a row of band b0 adds to b0 (weights [w, 2w)), b0 - 1 ([2w, 3w)), b0 + 1 ([0, w))
        float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (short hID = 0; hID < heightOfLSP; hID++) {
            const int b0 = hID / w;
            int off;
            if (lane == b0) off = w;
            else if (lane == b0 - 1) off = 2 * w;
            else if (lane == b0 + 1) off = 0;
            else continue;
            const float c = tab.coef_l[hID % w + off];
            sum[0] += c * s_row[wv][0][hID];
            sum[1] += c * s_row[wv][1][hID];
            sum[2] += c * c * s_row[wv][2][hID];
            sum[3] += c * c * s_row[wv][3][hID];
            sum[4] += c * s_row[wv][4][hID];
            sum[5] += c * s_row[wv][5][hID];
            sum[6] += c * c * s_row[wv][6][hID];
            sum[7] += c * c * s_row[wv][7][hID];
        }
        const float invN2 = (float)(1.0 / (w * 2.0)), invN3 = (float)(1.0 / (w * 3.0));
        const float invN = (lane == 0 || lane == LBD_BANDS - 1) ? invN2 : invN3;
        float* d = &s_des[wv][lane * 8];
        float temp = sum[0] * invN;
        d[0] = temp;
        d[4] = sqrtf(sum[2] * invN - temp * temp);
        temp = sum[1] * invN;
        d[1] = temp;
        d[5] = sqrtf(sum[3] * invN - temp * temp);
        temp = sum[4] * invN;
        d[2] = temp;
        d[6] = sqrtf(sum[6] * invN - temp * temp);
        temp = sum[5] * invN;
        d[3] = temp;
        d[7] = sqrtf(sum[7] * invN - temp * temp);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (lane == 0) {
        float* desVec = s_des[wv];
        float tempM = 0, tempS = 0;
        for (int i = 0; i < LBD_BANDS * 8; i += 8) {
            tempM += desVec[i] * desVec[i];
            tempM += desVec[i + 1] * desVec[i + 1];
            tempM += desVec[i + 2] * desVec[i + 2];
            tempM += desVec[i + 3] * desVec[i + 3];
            tempS += desVec[i + 4] * desVec[i + 4];
            tempS += desVec[i + 5] * desVec[i + 5];
            tempS += desVec[i + 6] * desVec[i + 6];
            tempS += desVec[i + 7] * desVec[i + 7];
        }
        tempM = 1 / sqrtf(tempM);
        tempS = 1 / sqrtf(tempS);
        float temp = 0;
        for (int i = 0; i < LBD_BANDS * 8; ++i) {
            float v = desVec[i] * ((i & 4) ? tempS : tempM);
            if ((double)v > 0.4) v = (float)0.4;
            desVec[i] = v;
        }
        for (int i = 0; i < LBD_BANDS * 8; ++i) temp += desVec[i] * desVec[i];
        temp = 1 / sqrtf(temp);
        for (int i = 0; i < LBD_BANDS * 8; ++i) desVec[i] = desVec[i] * temp;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int i = lane; i < LBD_BANDS * 8; i += 64) lbd[(size_t)li * (LBD_BANDS * 8) + i] = s_des[wv][i];
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

void lbd_lines_dev(const plslam_lbd_line* lines, int32_t n, LbdLineDev* out)
{
    for (int32_t i = 0; i < n; ++i) {
        out[i].num_pixels = lines[i].num_pixels;
        out[i].sx = lines[i].sx; out[i].sy = lines[i].sy; out[i].ex = lines[i].ex; out[i].ey = lines[i].ey;
        // dL[0] = cos(direction), :1117-1118 -- the FLOAT overload (cosf), as the reference binds it under libstdc++
        out[i].dl0 = std::cos(lines[i].direction);
        out[i].dl1 = std::sin(lines[i].direction);
        out[i].pad = 0.f;
    }
}

int launch_lbd_compute(const int16_t* dx, const int16_t* dy, int32_t width, int32_t height, const LbdLineDev* d_lines,
                       int32_t n, int32_t w, float* lbd, hipStream_t s)
{
    if (n <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_lbd_compute, dim3((unsigned)((n + LBD_LINES_PER_WG - 1) / LBD_LINES_PER_WG)),
                       dim3(64 * LBD_LINES_PER_WG), 0, s, dx, dy, width, height, d_lines, n, w, make_tables(w), lbd);
    PLSLAM_HIP_CHECK(hipGetLastError());
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
    const size_t oX = ci.take(img), oY = ci.take(img), oL = ci.take((size_t)n * sizeof(LbdLineDev));
    const size_t out_bytes = (size_t)n * 72 * 4;
    if ((rc = ctx->in_a.reserve(ci.off))) return rc;
    if ((rc = ctx->out_a.reserve(out_bytes))) return rc;
    if ((rc = ctx->pin_in.reserve((size_t)n * sizeof(LbdLineDev)))) return rc;
    lbd_lines_dev(lines, n, ctx->pin_in.as<LbdLineDev>());
    char* d = ctx->in_a.as<char>();
    hipStream_t s = ctx->stream;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oX, dx_img, img, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oY, dy_img, img, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + oL, ctx->pin_in.p, (size_t)n * sizeof(LbdLineDev), hipMemcpyHostToDevice, s));
    if ((rc = launch_lbd_compute((const int16_t*)(d + oX), (const int16_t*)(d + oY), width, height,
                                 (const LbdLineDev*)(d + oL), n, width_of_band, ctx->out_a.as<float>(), s)))
        return rc;
    PLSLAM_HIP_CHECK(hipMemcpyAsync(lbd_f32, ctx->out_a.p, out_bytes, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
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
    // the line records are small host data (the detector's output): staged through the context, ordered on `s`
    if ((rc = ctx->misc_b.reserve((size_t)n * sizeof(LbdLineDev)))) return rc;
    std::vector<LbdLineDev> tmp((size_t)n);
    lbd_lines_dev(lines_host, n, tmp.data());
    PLSLAM_HIP_CHECK(hipMemcpyAsync(ctx->misc_b.p, tmp.data(), (size_t)n * sizeof(LbdLineDev), hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));            // tmp goes out of scope
    return launch_lbd_compute(dx_img, dy_img, width, height, ctx->misc_b.as<LbdLineDev>(), n, width_of_band, lbd_f32, s);
}

}  // extern "C"
