// lba_rows_dev.hpp -- device-side arithmetic of the local-BA rows (K3 / K4), shared by the row kernels (lba.hip) and the fused
// iteration kernels of the LBA plan (lba_assemble.hip).  fp64; every translation unit that includes this is compiled with
// -ffp-contract=off, and every expression keeps the reference's source order (src/mapHandler.cpp:1358-1407, :1436-1516).
#pragma once

#include "common.hpp"

namespace plslam {

struct CamD { double fx, fy, cx, cy; double width, height; };

__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }  // std::max

// stvo-pl PinholeStereoCamera::projection: u = cx + fx*X/Z, v = cy + fy*Y/Z
__device__ __forceinline__ void project(const CamD& K, const double P[3], double& u, double& v)
{
    u = K.cx + K.fx * P[0] / P[2];
    v = K.cy + K.fy * P[1] / P[2];
}

// inverse_se3 (stvo-pl): Tiw = [R^T, -R^T t] of the row-major 4x4 at T  (:1372)
__device__ __forceinline__ void inv_pose(const double* __restrict__ T, double R[9], double t[3])
{
    double m[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) m[i] = T[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        R[3 * i] = m[i];
        R[3 * i + 1] = m[4 + i];
        R[3 * i + 2] = m[8 + i];
        t[i] = (-m[i]) * m[3] + (-m[4 + i]) * m[7] + (-m[8 + i]) * m[11];
    }
}

// A landmark's 3 (or 6) doubles with the fewest load instructions: a 16-byte load needs dword alignment only on this hardware,
// so a row of 24 bytes at an 8-byte aligned address is one 16-byte load and one of 8 -- gathers touch 64 cache lines per
// instruction whatever their width, and the vector memory path is what the row kernels wait for.
typedef double f64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
__device__ __forceinline__ void load3(const double* __restrict__ p, double (&v)[3])
{
    const f64x2_a8 a = *reinterpret_cast<PLSLAM_AS1 const f64x2_a8*>(g_(p));
    v[0] = a.x; v[1] = a.y; v[2] = g_(p)[2];
}
__device__ __forceinline__ void load6(const double* __restrict__ p, double (&v)[6])
{
    const PLSLAM_AS1 f64x2_a8* q = reinterpret_cast<PLSLAM_AS1 const f64x2_a8*>(g_(p));
    const f64x2_a8 a = q[0], b = q[1], c = q[2];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
}

// The pose matrices of a workgroup's rows through LDS (round 6).  A row reads the first 12 doubles of its keyframe's 4 x 4; the
// lanes of a wave name a handful of keyframes between them, and 12 loads of 64 addresses each, nearly all equal, kept the vector
// memory path as busy as the row stores did (K3 at C3 sizes, same box: 0.345 -> 0.276 ms per launch with the matrices in LDS).
// The workgroup copies the first min(n_pose_slots, NC) matrices into LDS -- loads that wait for nothing, beside the loads of the
// rows' slot numbers -- and a row whose slot is among them reads LDS, any other one global memory as before.  n_pose_slots = how
// many matrices the caller says the array holds (0: unknown -- nothing is copied).  A first form that cached the slots the
// workgroup's rows NAME (no count needed) chained slot numbers -> tags -> matrices -> rows through two barriers and was no
// faster than the global loads.  Every thread of the workgroup calls pose_cache_fill and pose12_take (one barrier).
#ifndef PLSLAM_POSE_LINES
#define PLSLAM_POSE_LINES 32
#endif
template <int NC>
struct PoseCache {
    double T[NC * 16];
};
// pose_cache_fill: the copy's loads and LDS stores (no barrier) -- the caller issues its rows' own loads (slot numbers, landmark
// indices, observations, the landmark gather) AROUND it, so that the matrices arrive beside them and the one barrier
// (pose12_take) waits for nothing a row would not have waited for anyway.  A first placement had the landmark index loaded behind
// the barrier: three dependent round trips per row instead of two, and no gain over the global loads.
template <int NC>
__device__ __forceinline__ void pose_cache_fill(PoseCache<NC>& c, const double* __restrict__ Tg, int32_t n_pose_slots)
{
    static_assert(NC * 16 <= 512, "two loads per lane of a 256-lane workgroup");
    const int n16 = (n_pose_slots < NC ? n_pose_slots : NC) * 16, i0 = threadIdx.x, i1 = threadIdx.x + blockDim.x;
    double a = 0.0, b = 0.0;
    if (i0 < n16) a = g_(Tg)[i0];
    if (i1 < n16) b = g_(Tg)[i1];
    if (i0 < n16) c.T[i0] = a;
    if (i1 < n16) c.T[i1] = b;
}
template <int NC>
__device__ __forceinline__ void pose12_take(PoseCache<NC>& c, const double* __restrict__ Tg, int32_t n_pose_slots, int32_t slot,
                                            double (&T12)[12])
{
    const int ncache = n_pose_slots < NC ? n_pose_slots : NC;
    __syncthreads();
    // (the two address spaces spelt out: one pointer chosen between them would be a generic one -- FLAT loads)
    typedef __attribute__((address_space(3))) const double* lds_f64;
    if (slot < ncache) {
        const lds_f64 lt = (lds_f64)c.T + slot * 16;
#pragma unroll
        for (int e = 0; e < 12; ++e) T12[e] = lt[e];
    } else {
#pragma unroll
        for (int e = 0; e < 12; ++e) T12[e] = g_(Tg)[(size_t)slot * 16 + e];
    }
}

__device__ __forceinline__ void xform(const double R[9], const double t[3], const double* X,
                                      double o[3])
{
    const double x = X[0], y = X[1], z = X[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = (R[3 * i] * x + R[3 * i + 1] * y + R[3 * i + 2] * z) + t[i];
}

// the 6-vector of :1392-1397 / :1475-1480
__device__ __forceinline__ void jac6(double a, double b, double k, const double G[3], double J[6])
{
    const double gx = G[0], gy = G[1], gz = G[2];
    J[0] = +k * a * gz;
    J[1] = +k * b * gz;
    J[2] = -k * (a * gx + b * gy);
    J[3] = -k * (a * gx * gy + b * gy * gy + b * gz * gz);
    J[4] = +k * (a * gx * gx + a * gz * gz + b * gx * gy);
    J[5] = +k * (b * gx * gz - a * gy * gz);
}

// Coalesced row store: every lane holds NW doubles of its own output row (row-major [obs][NW]).
// Written straight to global memory, one instruction touches 16 B out of every 8*NW B; instead the
// wave drops its 64 rows into a private LDS slab and streams the slab out linearly, 16 B per lane
// per instruction (1 KB contiguous per instruction).  `valid` = rows of this wave inside nobs.
template <int NW>
__device__ __forceinline__ void wave_store_rows(double* __restrict__ gbase /* row 0 of this wave */,
                                                const double (&v)[NW], double* __restrict__ slab, int lane,
                                                int valid)
{
#pragma unroll
    for (int c = 0; c < NW; ++c) slab[lane * NW + c] = v[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (reinterpret_cast<uintptr_t>(gbase) & 15) {          // (wave-uniform) a base that is only 8-byte aligned: word by word, in runs
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            const int i = k * 64 + lane;
            if (i < valid * NW) __builtin_nontemporal_store(slab[i], gbase + i);
        }
        __builtin_amdgcn_wave_barrier();
        return;
    }
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const int total2 = valid * NW / 2;                      // number of double2 chunks (NW*valid is even
    const f64x2* s2 = reinterpret_cast<const f64x2*>(slab);      //  unless NW and valid are odd)
    f64x2* g2 = reinterpret_cast<f64x2*>(gbase);
#pragma unroll
    for (int k = 0; k < (NW + 1) / 2; ++k) {
        const int i = k * 64 + lane;
        if (i < total2) __builtin_nontemporal_store(s2[i], g2 + i);     // rows are written once and read by a later kernel
    }
    if ((valid * NW) & 1) {                                  // odd tail double
        if (lane == 0) gbase[valid * NW - 1] = slab[valid * NW - 1];
    }
    __builtin_amdgcn_wave_barrier();
}

// The same for a PART of a wider row: every lane holds NW (even) doubles that belong at columns [col0, col0 + NW) of its row of
// STRIDE doubles (gbase = row 0 of this wave, column col0; 16-byte aligned).  The slab streams out in runs of NW doubles: 8 * NW
// contiguous bytes per row instead of one 8-byte store per lane and column.
template <int NW, int STRIDE>
__device__ __forceinline__ void wave_store_row_parts(double* __restrict__ gbase, const double (&v)[NW], double* __restrict__ slab,
                                                     int lane, int valid)
{
    static_assert(NW % 2 == 0 && STRIDE % 2 == 0, "double2 chunks");
#pragma unroll
    for (int c = 0; c < NW; ++c) slab[lane * NW + c] = v[c];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const int total2 = valid * (NW / 2);
    const f64x2* s2 = reinterpret_cast<const f64x2*>(slab);
    f64x2* g2 = reinterpret_cast<f64x2*>(gbase);
#pragma unroll
    for (int k = 0; k < NW / 2; ++k) {
        const int i = k * 64 + lane;
        if (i < total2) {
            const int row = i / (NW / 2), c2 = i - row * (NW / 2);
            __builtin_nontemporal_store(s2[i], g2 + (size_t)row * (STRIDE / 2) + c2);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// One point row (src/mapHandler.cpp:1358-1407): J_pose (6), J_lm (3), r, w from the keyframe's pose T (row-major 4x4), the
// landmark Xw and the observation ob.
__device__ __forceinline__ void point_row(const CamD& K, double th, const double* __restrict__ T, const double* __restrict__ X,
                                          const double2 ob, double (&out6)[6], double (&out3)[3], double& nrm_out, double& w_out)
{
    double R[9], t[3], G[3], Jc[6];
    inv_pose(T, R, t);
    xform(R, t, X, G);
    double pu, pv;
    project(K, G, pu, pv);
    const double dx = ob.x - pu, dy = ob.y - pv;
    const double nrm = sqrt(dx * dx + dy * dy);
    const double k = 1.0 / dmax(th, G[2] * G[2]);
    const double a = K.fx * dx, b = K.fy * dy;
    jac6(a, b, k, G, Jc);
    const double den = dmax(th, nrm);
#pragma unroll
    for (int c = 0; c < 6; ++c) out6[c] = Jc[c] / den;
#pragma unroll
    for (int j = 0; j < 3; ++j) out3[j] = (Jc[0] * R[j] + Jc[1] * R[3 + j] + Jc[2] * R[6 + j]) / den;
    nrm_out = nrm;
    w_out = 1.0 / (1.0 + nrm * nrm);
}

// One line row (:1436-1516; compat = the iteration pass's quirks, :1668-1772): J_lm (6), J_pose (6), r, w.
__device__ __forceinline__ void line_row(const CamD& K, double th, const double* __restrict__ T, const double* __restrict__ Pw,
                                         const double* __restrict__ Qw, double lx, double ly, double lz, double (&outl)[6],
                                         double (&outp)[6], double& nrm_out, double& w_out)
{
    double R[9], t[3], P[3], Q[3], JP[6], JQ[6];
    inv_pose(T, R, t);
    xform(R, t, Pw, P);
    xform(R, t, Qw, Q);
    double pu, pv, qu, qv;
    project(K, P, pu, pv);
    project(K, Q, qu, qv);
    const double e0 = lx * pu + ly * pv + lz;
    const double e1 = lx * qu + ly * qv + lz;
    const double nrm = sqrt(e0 * e0 + e1 * e1);
    const double a = K.fx * e0, b = K.fy * e1;  // sic: the reference multiplies by l_err (:1469-1472)
    const double kP = 1.0 / dmax(th, P[2] * P[2]);
    const double kQ = 1.0 / dmax(th, Q[2] * Q[2]);
    jac6(a, b, kP, P, JP);
    jac6(a, b, kQ, Q, JQ);
    const double den = dmax(th, nrm);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double vp = JP[0] * R[j] + JP[1] * R[3 + j] + JP[2] * R[6 + j];
        const double vq = JQ[0] * R[j] + JQ[1] * R[3 + j] + JQ[2] * R[6 + j];
        outl[j] = vp * e0 / den;
        outl[3 + j] = vq * e1 / den;
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) outp[c] = (JP[c] * e0 + JQ[c] * e1) / den;
    nrm_out = nrm;
    w_out = 1.0 / (1.0 + nrm * nrm);
}

}  // namespace plslam
