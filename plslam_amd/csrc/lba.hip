// lba.hip -- K3/K4 (local-BA residual + Jacobian rows) and K5/K6 (map<->keyframe gates), gfx950.
//
// fp64 elementwise work, one lane per observation.  HBM-bound: 152 B (point) / 208 B (line) of
// compulsory traffic per row against ~100 / ~220 flops.  The translation unit is compiled with
// -ffp-contract=off and every expression keeps the reference's source order
// (src/mapHandler.cpp:1358-1407, :1436-1516, :605-613, :720-729) so that thresholded results
// (inlier masks) are reproducible bit for bit against the CPU restatement.
#include "lba_rows_dev.hpp"

namespace plslam {

// K3: point rows
__global__ void __launch_bounds__(256)
k_point_rows(CamD K, double th, const double* __restrict__ T, const double* __restrict__ Xw,
             const double* __restrict__ uv, const int32_t* __restrict__ lm,
             const int32_t* __restrict__ kf, int32_t nobs, double* __restrict__ Jp,
             double* __restrict__ Jl, double* __restrict__ r, double* __restrict__ w, int32_t n_pose_slots)
{
    __shared__ __attribute__((aligned(16))) double slabs[4][64 * 6];
    __shared__ PoseCache<PLSLAM_POSE_LINES> poses;
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o0 = o - lane;                                 // first observation of this wave
    const int oc = o < nobs ? o : nobs - 1;                  // clamp: tail lanes recompute the last row
    // every load a row needs goes out before the workgroup's one barrier: slot number, landmark index, observation, the copy of
    // the pose matrices into LDS, the landmark
    const int32_t slot = kf[oc], l = lm[oc];
    const double2 ob = reinterpret_cast<const double2*>(uv)[oc];
    pose_cache_fill(poses, T, n_pose_slots);
    double X3[3], T12[12];
    load3(Xw + 3 * (size_t)l, X3);
    pose12_take(poses, T, n_pose_slots, slot, T12);          // (every wave of the workgroup: a barrier inside)
    if (o0 >= nobs) return;                                  // whole wave out of range (wave-uniform)
    const int valid = nobs - o0 < 64 ? nobs - o0 : 64;
    double out6[6], out3[3], nrm, wgt;
    point_row(K, th, T12, X3, ob, out6, out3, nrm, wgt);
    wave_store_rows<6>(Jp + 6 * (size_t)o0, out6, slabs[wave], lane, valid);
    wave_store_rows<3>(Jl + 3 * (size_t)o0, out3, slabs[wave], lane, valid);
    if (o < nobs) {
        r[o] = nrm;
        w[o] = wgt;
    }
}

// K4: line rows.  The pose matrices are NOT taken through LDS here (n_pose_slots is accepted and unused): measured on one box
// (tools/r6_rows_ab.sh) 0.344 ms per 10.2 M rows with the copy against 0.326 without -- a line row's loads (9 doubles of landmark
// and observation more than a point row's) hide the gather.  A register budget of 80 / 72 instead of the compiler's 84 (six / seven
// waves per SIMD instead of five) changed nothing; 64 spills and doubles the time.
__global__ void __launch_bounds__(256)
k_line_rows(CamD K, double th, int compat, const double* __restrict__ T,
            const double* __restrict__ Lw, const double* __restrict__ lobs,
            const int32_t* __restrict__ lm, const int32_t* __restrict__ kf, int32_t nobs,
            double* __restrict__ Jp, double* __restrict__ Jl, double* __restrict__ r,
            double* __restrict__ w, int32_t n_pose_slots)
{
    __shared__ __attribute__((aligned(16))) double slabs[4][64 * 6];
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int o0 = o - lane;
    const int oc = o < nobs ? o : nobs - 1;
    const int32_t slot = kf[oc];
    const size_t l0 = (size_t)lm[oc];
    double PQ[6], lo[3], T12[12];
    load3(lobs + 3 * (size_t)oc, lo);
    (void)n_pose_slots;
#pragma unroll
    for (int e = 0; e < 12; ++e) T12[e] = g_(T)[(size_t)slot * 16 + e];
    if (compat) {                                            // (the iteration pass's quirk: both end points = the 3 doubles at 3 l0)
        double P3[3];
        load3(Lw + 3 * l0, P3);
#pragma unroll
        for (int a = 0; a < 3; ++a) PQ[a] = PQ[3 + a] = P3[a];
    } else {
        load6(Lw + 6 * l0, PQ);
    }
    if (o0 >= nobs) return;
    const int valid = nobs - o0 < 64 ? nobs - o0 : 64;
    double outl[6], outp[6], nrm, wgt;
    line_row(K, th, T12, PQ, PQ + 3, lo[0], lo[1], lo[2], outl, outp, nrm, wgt);
    wave_store_rows<6>(Jl + 6 * (size_t)o0, outl, slabs[wave], lane, valid);
    wave_store_rows<6>(Jp + 6 * (size_t)o0, outp, slabs[wave], lane, valid);
    if (o < nobs) {
        r[o] = nrm;
        w[o] = wgt;
    }
}

struct Pose12 { double m[12]; };  // rows 0..2 of the row-major 4x4

__device__ __forceinline__ void xform44(const Pose12& T, const double* X, double o[3])
{
    const double x = X[0], y = X[1], z = X[2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o[i] = (T.m[4 * i] * x + T.m[4 * i + 1] * y + T.m[4 * i + 2] * z) + T.m[4 * i + 3];
}

// The gates as the LAST kernel of a one-synchronisation driver call (map2kf.hip): the workgroup that finishes last copies the call's
// counters (device words, some of them this kernel's own atomic counts) into the page-locked block the host reads -- a launch less
// at the end of a call that is bound by its launches.  Every wave WAITS for its count atomic (a returning one) before the
// workgroup's barrier, one lane then counts the workgroup in; the last one reads the counters with agent-scope loads.
struct GatePublish { int32_t* done; const int32_t* src; int32_t* dst; int32_t n; };     // done = nullptr: nothing to publish
__device__ __forceinline__ void gate_count_and_publish(int ok, int32_t* __restrict__ count, const GatePublish& pub)
{
    if (count) {
        const unsigned long long b = __ballot(ok);
        if ((threadIdx.x & 63) == 0 && b) {
            if (pub.done) {
                const int old = atomicAdd(count, (int)__popcll(b));
                asm volatile("" ::"v"(old));              // (the atomic has been performed when the wave passes here)
            } else {
                atomicAdd(count, (int)__popcll(b));
            }
        }
    }
    if (pub.done) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const int prev = atomicAdd(pub.done, 1);
            if (prev == (int)gridDim.x - 1)
                for (int w = 0; w < pub.n; ++w) pub.dst[w] = __hip_atomic_load(pub.src + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// K5: point gate  (:601-613)
__global__ void __launch_bounds__(256)
k_point_gate(CamD K, Pose12 Twf, const double* __restrict__ Xw, const int32_t* __restrict__ m12,
             int32_t nq, const double* __restrict__ pl, double th, uint8_t* __restrict__ mask,
             int32_t* __restrict__ count, const int32_t* __restrict__ nq_dev, const int32_t* __restrict__ idx,
             const int32_t* __restrict__ ti, int32_t* __restrict__ map_to_kf, GatePublish pub)
{
    if (nq_dev) nq = *nq_dev;                  // (the row count lives on the device: the launch covers an upper bound)
    const int i = blockIdx.x * 256 + threadIdx.x;
    int ok = 0;
    if (i < nq) {
        const int i2 = m12[i];
        if (i2 >= 0) {
            double Pf[3], u, v;
            xform44(Twf, Xw + 3 * (size_t)i, Pf);
            project(K, Pf, u, v);
            const double ex = u - pl[2 * (size_t)i2], ey = v - pl[2 * (size_t)i2 + 1];
            ok = sqrt(ex * ex + ey * ey) < th;
        }
        mask[i] = (uint8_t)ok;
        if (ok && map_to_kf) map_to_kf[idx[i]] = ti[i2];          // :614-619, the association (table pre-filled with -1)
    }
    gate_count_and_publish(ok, count, pub);
}

// K6: line gate  (:716-729), signed test on both endpoints
__global__ void __launch_bounds__(256)
k_line_gate(CamD K, Pose12 Twf, const double* __restrict__ Lw, const int32_t* __restrict__ m12,
            int32_t nq, const double* __restrict__ le, double th, uint8_t* __restrict__ mask,
            int32_t* __restrict__ count, const int32_t* __restrict__ nq_dev, const int32_t* __restrict__ idx,
            const int32_t* __restrict__ ti, int32_t* __restrict__ map_to_kf, GatePublish pub)
{
    if (nq_dev) nq = *nq_dev;
    const int i = blockIdx.x * 256 + threadIdx.x;
    int ok = 0;
    if (i < nq) {
        const int i2 = m12[i];
        if (i2 >= 0) {
            double sP[3], eP[3], su, sv, eu, ev;
            xform44(Twf, Lw + 6 * (size_t)i, sP);
            project(K, sP, su, sv);
            xform44(Twf, Lw + 6 * (size_t)i + 3, eP);
            project(K, eP, eu, ev);
            const double lx = le[3 * (size_t)i2], ly = le[3 * (size_t)i2 + 1], lz = le[3 * (size_t)i2 + 2];
            const double e0 = lx * su + ly * sv + lz;
            const double e1 = lx * eu + ly * ev + lz;
            ok = (e0 < th) && (e1 < th);
        }
        mask[i] = (uint8_t)ok;
        if (ok && map_to_kf) map_to_kf[idx[i]] = ti[i2];
    }
    gate_count_and_publish(ok, count, pub);
}

__device__ __forceinline__ int inside(const CamD& K, const double P[3])
{
    double u, v;
    project(K, P, u, v);
    return u > 0 && u < K.width && v > 0 && v < K.height && P[2] > 0.0;
}

// candidate pre-filter (:549-551, :650-655)
__global__ void __launch_bounds__(256)
k_visible(CamD K, Pose12 Twf, const double* __restrict__ X, int32_t n, int lines,
          uint8_t* __restrict__ vis, const uint8_t* __restrict__ cand)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const bool c = cand == nullptr || cand[i] != 0;      // (cand: the caller's candidate flags live on the device -- folded in)
    double P[3];
    if (!lines) {
        xform44(Twf, X + 3 * (size_t)i, P);
        vis[i] = (uint8_t)(c && inside(K, P));
    } else {
        double E[3];
        xform44(Twf, X + 6 * (size_t)i, P);
        xform44(Twf, X + 6 * (size_t)i + 3, E);
        vis[i] = (uint8_t)(c && inside(K, P) && inside(K, E));
    }
}

static CamD cam_d(const plslam_cam& K)
{
    return CamD{K.fx, K.fy, K.cx, K.cy, (double)K.width, (double)K.height};
}
static Pose12 pose12(const double* T16)
{
    Pose12 p;
    for (int i = 0; i < 12; ++i) p.m[i] = T16[i];
    return p;
}

int launch_point_rows(const plslam_cam& K, double th, const double* T, const double* Xw,
                      const double* uv, const int32_t* lm, const int32_t* kf, int32_t nobs,
                      double* Jp, double* Jl, double* r, double* w, hipStream_t s, int32_t n_pose_slots)
{
    if (nobs <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_point_rows, dim3((nobs + 255) / 256), dim3(256), 0, s, cam_d(K), th, T, Xw,
                       uv, lm, kf, nobs, Jp, Jl, r, w, n_pose_slots);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int launch_line_rows(const plslam_cam& K, double th, int compat, const double* T, const double* Lw,
                     const double* lobs, const int32_t* lm, const int32_t* kf, int32_t nobs,
                     double* Jp, double* Jl, double* r, double* w, hipStream_t s, int32_t n_pose_slots)
{
    if (nobs <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_line_rows, dim3((nobs + 255) / 256), dim3(256), 0, s, cam_d(K), th, compat, T,
                       Lw, lobs, lm, kf, nobs, Jp, Jl, r, w, n_pose_slots);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int launch_point_gate(const plslam_cam& K, const double* Twf16, const double* Xw, const int32_t* m12,
                      int32_t nq, const double* pl, double th, uint8_t* mask, int32_t* count,
                      hipStream_t s)
{
    if (count) PLSLAM_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    if (nq <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_point_gate, dim3((nq + 255) / 256), dim3(256), 0, s, cam_d(K), pose12(Twf16),
                       Xw, m12, nq, pl, th, mask, count, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr,
                       (int32_t*)nullptr, GatePublish{nullptr, nullptr, nullptr, 0});
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int launch_line_gate(const plslam_cam& K, const double* Twf16, const double* Lw, const int32_t* m12,
                     int32_t nq, const double* le, double th, uint8_t* mask, int32_t* count,
                     hipStream_t s)
{
    if (count) PLSLAM_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(int32_t), s));
    if (nq <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_line_gate, dim3((nq + 255) / 256), dim3(256), 0, s, cam_d(K), pose12(Twf16),
                       Lw, m12, nq, le, th, mask, count, (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr,
                       (int32_t*)nullptr, GatePublish{nullptr, nullptr, nullptr, 0});
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

int launch_visible(const plslam_cam& K, const double* Twf16, const double* X, int32_t n, int lines,
                   uint8_t* vis, hipStream_t s)
{
    if (n <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_visible, dim3((n + 255) / 256), dim3(256), 0, s, cam_d(K), pose12(Twf16), X, n,
                       lines, vis, (const uint8_t*)nullptr);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}
// ... AND the candidate flags (device): what the drivers with a device-resident map read back
int launch_visible_cand(const plslam_cam& K, const double* Twf16, const double* X, const uint8_t* cand, int32_t n, int lines,
                        uint8_t* vis, hipStream_t s)
{
    if (n <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_visible, dim3((n + 255) / 256), dim3(256), 0, s, cam_d(K), pose12(Twf16), X, n,
                       lines, vis, cand);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

// pj_points / pj_lines of the fast_matching drivers (:553-555, :656-661): projection of the (already gathered)
// candidate landmarks in grid units, truncated toward zero as std::make_pair<int,int>(double, double) does; for
// lines also the unit direction matchGrid derives from the two INTEGER end points (zero vector -> NaN)
__global__ void __launch_bounds__(256)
k_project_cells(CamD K, Pose12 Twf, const double* __restrict__ X, int32_t n, int lines, double inv_w, double inv_h,
                int32_t* __restrict__ cells, double* __restrict__ dir1, const int32_t* __restrict__ n_dev)
{
    if (n_dev) n = *n_dev;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int nc = lines ? 2 : 1;
    int32_t c[4];
    // double -> int as the reference's x86 build does it (cvttsd2si): truncation; NaN / out of range -> INT_MIN
    auto cvtt = [](double v) -> int32_t { return (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN; };
    for (int e = 0; e < nc; ++e) {
        double P[3], u, v;
        xform44(Twf, X + (size_t)i * 3 * nc + 3 * e, P);
        project(K, P, u, v);
        c[2 * e] = cvtt(u * inv_w);
        c[2 * e + 1] = cvtt(v * inv_h);
        cells[((size_t)i * nc + e) * 2] = c[2 * e];
        cells[((size_t)i * nc + e) * 2 + 1] = c[2 * e + 1];
    }
    if (lines) {
        const double vx = (double)c[2] - (double)c[0], vy = (double)c[3] - (double)c[1];
        const double magnitude = sqrt(vx * vx + vy * vy);
        dir1[2 * (size_t)i] = vx / magnitude;
        dir1[2 * (size_t)i + 1] = vy / magnitude;
    }
}

int launch_project_cells(const plslam_cam& K, const double* Twf16, const double* X, int32_t n, int lines, double inv_w,
                         double inv_h, int32_t* cells, double* dir1, hipStream_t s)
{
    if (n <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_project_cells, dim3((n + 255) / 256), dim3(256), 0, s, cam_d(K), pose12(Twf16), X, n, lines,
                       inv_w, inv_h, cells, dir1, (const int32_t*)nullptr);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

// The gates with the row count ON THE DEVICE (*n_dev <= n_max; the launch covers n_max rows): the drivers' one-synchronisation
// form builds its candidate list on the device and never learns its length before the results are back.  The gate's counter is
// NOT cleared here (the caller's image holds the zero); idx / ti / map_to_kf: the association of the rows that pass.
// publish_*: the call's counters (publish_n device words at publish_src) go to publish_dst -- page-locked, mapped -- from the last
// workgroup to finish; publish_done: a ZERO device word the workgroups count themselves into (nullptr: nothing is published)
int launch_gate_n(int lines, const plslam_cam& K, const double* Twf16, const double* LM, const int32_t* m12, const int32_t* n_dev,
                  int32_t n_max, const double* feat, double th, uint8_t* mask, int32_t* count, const int32_t* idx, const int32_t* ti,
                  int32_t* map_to_kf, hipStream_t s, int32_t* publish_done, const int32_t* publish_src, int32_t* publish_dst,
                  int32_t publish_n)
{
    if (n_max <= 0) {
        // (no gate workgroup will run: the counters still have to come down)
        PLSLAM_REQUIRE(!publish_done, PLSLAM_EINVAL);
        return PLSLAM_OK;
    }
    const GatePublish pub{publish_done, publish_src, publish_dst, publish_n};
    if (lines)
        hipLaunchKernelGGL(k_line_gate, dim3((n_max + 255) / 256), dim3(256), 0, s, cam_d(K), pose12(Twf16), LM, m12, n_max, feat,
                           th, mask, count, n_dev, idx, ti, map_to_kf, pub);
    else
        hipLaunchKernelGGL(k_point_gate, dim3((n_max + 255) / 256), dim3(256), 0, s, cam_d(K), pose12(Twf16), LM, m12, n_max, feat,
                           th, mask, count, n_dev, idx, ti, map_to_kf, pub);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}


// ---- the map<->keyframe drivers' one-synchronisation form (map2kf.hip): its small kernels fused -- a launch of a
// microsecond's work costs 4-5 us on the timeline ------------------------------------------------------------------------------
// candidate pre-filter (:549-551, :650-655) x candidate flags -> the STABLE list of the landmarks that pass (ascending) + its
// length, the association table's -1 start, and the row count of the matchGrid problem that follows (its uploaded descriptor).
// A workgroup per 256 landmarks, all of them at once (a single workgroup evaluated 10 000 landmarks in 17 - 28 us: fp64 on one
// CU); a workgroup's survivors are counted by ballot and the list comes out ascending: a workgroup's first slot is the number of
// survivors in front of it, found by DECOUPLED LOOK-BACK (round 6; rounds 3-5 summed ALL predecessors' counts, b words per
// workgroup b: fine at C3's 40 workgroups, 7.6 M polls at a 1 M-landmark map).  part[b] is one word: bit 30 = "my own count is
// here", bit 31 = "the count of everything up to and including me is here".  A workgroup publishes its own count at once; its
// first wave then looks back 64 predecessors at a time -- the nearest one that already knows its inclusive sum ends the walk,
// the ones in between contribute their own counts -- and publishes its inclusive sum.  Normally one or two loads per lane.  It
// waits only for workgroups dispatched before it (a word without either bit), which were started earlier: the chain cannot
// wait on itself.  part: (n + 255) / 256 words, ZERO when the kernel starts (the drivers' upload image holds them).
constexpr int VC_NT = 256;
constexpr uint32_t VC_INCL = 0x80000000u, VC_AGG = 0x40000000u, VC_VAL = 0x3FFFFFFFu;
__global__ void __launch_bounds__(VC_NT)
k_visible_compact(CamD K, Pose12 Twf, const double* __restrict__ X, const uint8_t* __restrict__ cand, int32_t n, int lines,
                  int32_t* __restrict__ idx, int32_t* __restrict__ n_out, int32_t* __restrict__ fill, GridDesc* __restrict__ desc,
                  uint32_t* __restrict__ part)
{
    constexpr int NW = VC_NT / 64;
    __shared__ uint32_t s_w[NW], s_before[NW];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6, b = (int)blockIdx.x;
    const int32_t i = b * VC_NT + tid;
    bool v = false;
    if (i < n) {                                            // (the landmark is read whether it is a candidate or not: one round trip)
        const uint8_t c = cand[i];
        double P[3], E[3];
        if (!lines) {
            xform44(Twf, X + 3 * (size_t)i, P);
            v = c != 0 && inside(K, P) != 0;
        } else {
            xform44(Twf, X + 6 * (size_t)i, P);
            xform44(Twf, X + 6 * (size_t)i + 3, E);
            v = c != 0 && inside(K, P) && inside(K, E);
        }
        fill[i] = -1;
    }
    const uint64_t m = __ballot(v);
    if (lane == 0) s_w[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t own = 0, inside_wg = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        inside_wg += w < wv ? s_w[w] : 0u;
        own += s_w[w];
    }
    if (tid == 0) __hip_atomic_store(part + b, own | (b == 0 ? VC_INCL : VC_AGG), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wv == 0) {
        uint32_t before = 0;                                // survivors in front of this workgroup
        for (int base = b - 1; base >= 0; base -= 64) {
            const int p = base - lane;                      // lane 0 looks at the nearest predecessor
            uint32_t x = VC_INCL;                           // (in front of workgroup 0: an inclusive sum of nothing)
            if (p >= 0)
                while (!((x = __hip_atomic_load(part + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & (VC_INCL | VC_AGG))) __builtin_amdgcn_s_sleep(1);
            const uint64_t incl = __ballot((x & VC_INCL) != 0);
            const int first = incl ? (int)__builtin_ctzll(incl) : 64;        // the nearest predecessor that knows its inclusive sum
            uint32_t t = lane <= first ? (x & VC_VAL) : 0u;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) t += (uint32_t)__shfl_xor((int)t, o);
            before += t;
            if (incl) break;
        }
        if (lane == 0) {
            s_before[0] = before;
            if (b > 0) __hip_atomic_store(part + b, (before + own) | VC_INCL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    const uint32_t before = s_before[0];
    if (v) idx[before + inside_wg + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
    if (b == (int)gridDim.x - 1 && tid == 0) {
        *n_out = (int32_t)(before + own);
        if (desc) desc->n1 = (int32_t)(before + own);
    }
}

// Q matrix construction (:555, :567) + pj_points / pj_lines (k_project_cells): row a of Q = med_desc[idx[a]], of QL = LM[idx[a]],
// its window centre(s) from QL; a lane per listed landmark, *n_dev of them
__global__ void __launch_bounds__(256)
k_prepare_rows(CamD K, Pose12 Twf, const uint64_t* __restrict__ md, const double* __restrict__ lm, const int32_t* __restrict__ idx,
               const int32_t* __restrict__ n_dev, int lines, double inv_w, double inv_h, uint64_t* __restrict__ Q,
               double* __restrict__ QL, int32_t* __restrict__ cells, double* __restrict__ dir1)
{
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= *n_dev) return;
    const int64_t src = idx[a];
    const int nc = lines ? 2 : 1, lw = 3 * nc;
#pragma unroll
    for (int w = 0; w < 4; ++w) Q[(int64_t)a * 4 + w] = md[src * 4 + w];
    double X[6];
    for (int w = 0; w < lw; ++w) {
        X[w] = lm[src * lw + w];
        QL[(int64_t)a * lw + w] = X[w];
    }
    if (!cells) return;                                     // (the brute-force driver: no window centres)
    auto cvtt = [](double v) -> int32_t { return (v > -2147483649.0 && v < 2147483648.0) ? (int32_t)v : INT32_MIN; };
    int32_t c[4];
    for (int e = 0; e < nc; ++e) {
        double P[3], u, v;
        xform44(Twf, X + 3 * e, P);
        project(K, P, u, v);
        c[2 * e] = cvtt(u * inv_w);
        c[2 * e + 1] = cvtt(v * inv_h);
        cells[((size_t)a * nc + e) * 2] = c[2 * e];
        cells[((size_t)a * nc + e) * 2 + 1] = c[2 * e + 1];
    }
    if (lines) {
        const double vx = (double)c[2] - (double)c[0], vy = (double)c[3] - (double)c[1];
        const double magnitude = sqrt(vx * vx + vy * vy);
        dir1[2 * (size_t)a] = vx / magnitude;
        dir1[2 * (size_t)a + 1] = vy / magnitude;
    }
}

// part: visible_compact_part_words(n) device words that must be ZERO when the kernel starts (its workgroups chain their counts
// through them; a stale word gives wrong offsets).  part_zeroed = true: the caller's upload image has just written the zeros
// on this stream (the drivers: no extra operation on their latency path); false: they are cleared here.
size_t visible_compact_part_words(int32_t n) { return (size_t)(n > 0 ? (n + VC_NT - 1) / VC_NT : 1); }
int launch_visible_compact(const plslam_cam& K, const double* Twf16, const double* X, const uint8_t* cand, int32_t n, int lines,
                           int32_t* idx, int32_t* n_out, int32_t* fill, GridDesc* desc, uint32_t* part, bool part_zeroed,
                           hipStream_t s)
{
    const unsigned nb = (unsigned)visible_compact_part_words(n);
    if (!part_zeroed) PLSLAM_HIP_CHECK(hipMemsetAsync(part, 0, sizeof(uint32_t) * nb, s));
    hipLaunchKernelGGL(k_visible_compact, dim3(nb), dim3(VC_NT), 0, s, cam_d(K), pose12(Twf16), X, cand, n, lines, idx, n_out, fill,
                       desc, part);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}
int launch_prepare_rows(const plslam_cam& K, const double* Twf16, const void* md, const double* lm, const int32_t* idx,
                        const int32_t* n_dev, int32_t n_max, int lines, double inv_w, double inv_h, void* Q, double* QL, int32_t* cells,
                        double* dir1, hipStream_t s)
{
    if (n_max <= 0) return PLSLAM_OK;
    hipLaunchKernelGGL(k_prepare_rows, dim3((n_max + 255) / 256), dim3(256), 0, s, cam_d(K), pose12(Twf16),
                       static_cast<const uint64_t*>(md), lm, idx, n_dev, lines, inv_w, inv_h, static_cast<uint64_t*>(Q), QL, cells, dir1);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

}  // namespace plslam
