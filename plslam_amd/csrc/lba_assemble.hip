// lba_assemble.hip -- device-side assembly of the local-BA normal equations in BLOCK form.
//
// The reference adds every observation's blocks into a dense MatrixXd H(N,N) and VectorXd g
// (src/mapHandler.cpp:1410-1429 for points, :1519-1538 for lines) and then takes H.sparseView()
// (:1555).  H has a fixed block structure -- one 6x6 block per optimised keyframe, one 3x3 / 6x6
// block per landmark, one 3x6 / 6x6 cross block per observation -- so this file produces exactly
// those blocks (the Schur-complement-ready layout) and g.  Landmark and cross blocks keep the
// reference's accumulation ORDER (sequential over the landmark's observations in list order), so
// they are bit-exact against the dense accumulation; the keyframe blocks, which sum thousands of
// observations, use a fixed-shape two-level sum (deterministic, equal up to rounding).  No atomics
// (they would make the sums depend on scheduling).
//   K7  k_landmark_blocks<DL>   one lane per landmark: H_ll (DLxDL), g_l (DL) over its observations
//   K8  k_cross_blocks<DL>      one lane per observation: W = J_lm * J_pose^T * w (DLx6)
//   K9  k_pose_partials/_blocks one lane per (keyframe, chunk, entry) then per (keyframe, entry):
//                                H_pp (6x6) and g_p (6) over the keyframe's observations (points
//                                first, then lines, list order; 64-observation chunks)
//   K10 k_weighted_error_*      err = sum r^2 w (fixed-shape two-level tree: deterministic, not sequential)
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "lba_rows_dev.hpp"

namespace plslam {

template <int DL>
__global__ void __launch_bounds__(256)
k_landmark_blocks(const int32_t* __restrict__ lm_ptr, const int32_t* __restrict__ lm_obs, int32_t nlm,
                  const double* __restrict__ Jl, const double* __restrict__ r, const double* __restrict__ w,
                  double* __restrict__ Hll, double* __restrict__ gl)
{
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= nlm) return;
    double H[DL * DL], g[DL];
#pragma unroll
    for (int i = 0; i < DL * DL; ++i) H[i] = 0.0;
#pragma unroll
    for (int i = 0; i < DL; ++i) g[i] = 0.0;
    for (int k = lm_ptr[l]; k < lm_ptr[l + 1]; ++k) {
        const int o = lm_obs[k];
        double J[DL];
#pragma unroll
        for (int a = 0; a < DL; ++a) J[a] = Jl[(size_t)o * DL + a];
        const double rr = r[o], ww = w[o];
#pragma unroll
        for (int a = 0; a < DL; ++a) g[a] += J[a] * rr * ww;
#pragma unroll
        for (int a = 0; a < DL; ++a)
#pragma unroll
            for (int b = 0; b < DL; ++b) H[a * DL + b] += J[a] * J[b] * ww;
    }
#pragma unroll
    for (int i = 0; i < DL * DL; ++i) Hll[(size_t)l * DL * DL + i] = H[i];
#pragma unroll
    for (int i = 0; i < DL; ++i) gl[(size_t)l * DL + i] = g[i];
}

// TRANSPOSED (DL == 6 only): element (a, b) receives Jl[b] * Jp[a] * w -- the block as levMarquardtOptimizationGBA writes
// it for lines (src/mapHandler.cpp:2341-2352 against :1531-1532 of the local BA; a reference defect that callers after
// the reference's GBA numbers reproduce with PLSLAM_LBA_COMPAT_GBA)
template <int DL, bool TRANSPOSED = false>
__global__ void __launch_bounds__(256)
k_cross_blocks(const int32_t* __restrict__ kf_loc, int32_t nobs, const double* __restrict__ Jp,
               const double* __restrict__ Jl, const double* __restrict__ w, double* __restrict__ W)
{
    static_assert(!TRANSPOSED || DL == 6, "only the square line block can be transposed in place");
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= nobs) return;
    const bool opt = kf_loc[o] >= 0;   // kf_loc == -1: the keyframe is not optimised, no cross block
    double P[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) P[b] = Jp[(size_t)o * 6 + b];
    const double ww = w[o];
#pragma unroll
    for (int a = 0; a < DL; ++a) {
        const double ja = Jl[(size_t)o * DL + a];
#pragma unroll
        for (int b = 0; b < 6; ++b) W[TRANSPOSED ? ((size_t)o * 6 + b) * 6 + a : ((size_t)o * DL + a) * 6 + b] = opt ? ja * P[b] * ww : 0.0;
    }
}

// entry e of keyframe k: e < 36 -> H_pp[k][e/6][e%6]; e >= 36 -> g_p[k][e-36].
// Two-level, fixed-shape (deterministic) summation: chunk c of keyframe k sums observations
// [c*POSE_CHUNK, (c+1)*POSE_CHUNK) of the keyframe's list sequentially in list order, then the chunk
// partials are summed sequentially in chunk order.  (A single sequential chain over the ~6700
// observations of a C3 keyframe is bit-identical to the reference's dense accumulation but takes
// 1.9 ms on 9 workgroups; this takes microseconds and differs from it by rounding only.)
constexpr int POSE_CHUNK = 64;

__global__ void __launch_bounds__(64)
k_pose_partials(const int32_t* __restrict__ kf_ptr, const int32_t* __restrict__ kf_obs, int32_t n_pt_obs,
                const double* __restrict__ Jp_pt, const double* __restrict__ r_pt, const double* __restrict__ w_pt,
                const double* __restrict__ Jp_ls, const double* __restrict__ r_ls, const double* __restrict__ w_ls,
                int32_t max_chunks, double* __restrict__ part /* [nkf][max_chunks][42] */)
{
    const int k = blockIdx.x, c = blockIdx.y, e = threadIdx.x;
    if (e >= 42) return;
    const int beg = kf_ptr[k] + c * POSE_CHUNK;
    const int end = beg + POSE_CHUNK < kf_ptr[k + 1] ? beg + POSE_CHUNK : kf_ptr[k + 1];
    const int a = e < 36 ? e / 6 : e - 36, b = e < 36 ? e % 6 : 0;
    double acc = 0.0;
    // Same terms, same order -- the loads of PB observations are issued together (the loop was a chain of two dependent round
    // trips per observation: 128 of them = 48 us for work that takes microseconds)
    constexpr int PB = 16;
    for (int i0 = beg; i0 < end; i0 += PB) {
        int oo[PB];
        bool pt[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int o = kf_obs[i0 + j < end ? i0 + j : end - 1];      // global observation id: points [0, n_pt_obs), then lines
            pt[j] = o < n_pt_obs;
            oo[j] = pt[j] ? o : o - n_pt_obs;
        }
        double ja[PB], jb[PB], ww[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const double* J = (pt[j] ? Jp_pt : Jp_ls) + (size_t)oo[j] * 6;
            ja[j] = J[a];
            jb[j] = e < 36 ? J[b] : (pt[j] ? r_pt : r_ls)[oo[j]];
            ww[j] = (pt[j] ? w_pt : w_ls)[oo[j]];
        }
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (i0 + j < end) acc += ja[j] * jb[j] * ww[j];
    }
    part[((size_t)k * max_chunks + c) * 42 + e] = acc;   // empty chunks write 0
}

__global__ void __launch_bounds__(64)
k_pose_blocks(const int32_t* __restrict__ kf_ptr, int32_t max_chunks, const double* __restrict__ part,
              double* __restrict__ Hpp, double* __restrict__ gp)
{
    const int k = blockIdx.x, e = threadIdx.x;
    if (e >= 42) return;
    const int nchunks = (kf_ptr[k + 1] - kf_ptr[k] + POSE_CHUNK - 1) / POSE_CHUNK;
    double acc = 0.0;
    constexpr int PB = 8;                       // (same order of additions; the loads in batches)
    for (int c0 = 0; c0 < nchunks; c0 += PB) {
        double v[PB];
#pragma unroll
        for (int j = 0; j < PB; ++j) v[j] = part[((size_t)k * max_chunks + (c0 + j < nchunks ? c0 + j : nchunks - 1)) * 42 + e];
#pragma unroll
        for (int j = 0; j < PB; ++j)
            if (c0 + j < nchunks) acc += v[j];
    }
    if (e < 36) Hpp[(size_t)k * 36 + e] = acc;
    else gp[(size_t)k * 6 + (e - 36)] = acc;
}

// err = sum r^2 w in a fixed shape: ERR_BLOCKS workgroups of 256 lanes -- lane g of all ERR_BLOCKS x 256 sums observations g,
// g + ERR_BLOCKS x 256, ... (points, then lines) in that order, a tree over the workgroup's lanes -> err[1 + block] -- then one
// wave's tree over the ERR_BLOCKS partials -> err[0].  (One workgroup alone read the 0.96 MB of a C3 pass at a single CU's
// 60 GB/s: 58 us with one load per loop trip, 16 us with the loads batched.)
constexpr int ERR_BLOCKS = 64;
__global__ void __launch_bounds__(256)
k_weighted_error_partials(const double* __restrict__ r_pt, const double* __restrict__ w_pt, int32_t n_pt,
                          const double* __restrict__ r_ls, const double* __restrict__ w_ls, int32_t n_ls,
                          double* __restrict__ err)
{
    __shared__ double red[256];
    constexpr int G = ERR_BLOCKS * 256, PB = 4;
    const int g = blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    auto sum = [&](const double* __restrict__ r, const double* __restrict__ w, int32_t n) {
        for (int o0 = g; o0 < n; o0 += G * PB) {
            double rr[PB], ww[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int o = o0 + j * G;
                rr[j] = o < n ? r[o] : 0.0;
                ww[j] = o < n ? w[o] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < PB; ++j)
                if (o0 + j * G < n) acc += rr[j] * rr[j] * ww[j];
        }
    };
    sum(r_pt, w_pt, n_pt);
    sum(r_ls, w_ls, n_ls);
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) err[1 + blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(ERR_BLOCKS)
k_weighted_error_final(double* __restrict__ err)
{
    __shared__ double red[ERR_BLOCKS];
    red[threadIdx.x] = err[1 + threadIdx.x];
    __syncthreads();
    for (int s = ERR_BLOCKS / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) err[0] = red[0];
}

// ---------------------------------------------------------------------------------------------------------------------
// The iteration of an LBA plan in THREE launches (round 4; round 3: ten launches of 4-13 us each, i.e. launch granularity):
//   F1 k_lba_rows_cross   one lane per observation (points, then lines): the row (K3 / K4), its cross block (K8) from the
//                         values still in registers, and the workgroup's share of err = sum r^2 w
//   F2 k_lba_blocks       workgroups [0, nb3): point landmark blocks (K7<3>); [.., +nb6): line landmark blocks (K7<6>); the
//                         rest: the keyframes' chunk partials (K9, four 64-lane chunks per workgroup)
//   F3 k_lba_finish       workgroup k < nkf: keyframe k's blocks from its chunk partials (K9); workgroup nkf: err
// Every block keeps the summation ORDER of the separate kernels (landmark and cross blocks: the reference's, bit for bit;
// keyframe blocks: chunks of 64 observations in list order, then the chunks in order); err is the sum of the row workgroups'
// trees in workgroup order -- a fixed shape, deterministic, equal to the reference's sequential sum up to rounding.
// ---------------------------------------------------------------------------------------------------------------------
struct LbaIterArgs {
    CamD K;
    double th;
    int32_t compat_iter, transpose_ls_cross;
    const double *T, *Xw, *Lw, *uv, *lobs;
    const int32_t *pt_lm, *pt_slot, *pt_kf_loc, *ls_lm, *ls_slot, *ls_kf_loc;
    int32_t np, nl, nbp, nbl;                 // observations and row workgroups (points, lines)
    int32_t n_slots;                          // pose slots behind T (the first of them ride in LDS: pose12_cached)
    double *pJp, *pJl, *pr, *pw, *lJp, *lJl, *lr, *lw, *Wp, *Wl, *err_part;   // err_part[nbp + nbl]
};

__global__ void __launch_bounds__(256)
k_lba_rows_cross(const LbaIterArgs A)
{
    // (a slab holds a wave's 64 rows of up to 18 doubles: the cross blocks leave through it too -- written lane by lane they were 18 /
    // 36 instructions of 64 scattered 8-byte stores each)
    __shared__ __attribute__((aligned(16))) double slabs[4][64 * 18];
    __shared__ double red[256];
    __shared__ PoseCache<PLSLAM_POSE_LINES> poses;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool lines = (int)blockIdx.x >= A.nbp;
    const int nobs = lines ? A.nl : A.np;
    const int o = (lines ? (int)blockIdx.x - A.nbp : (int)blockIdx.x) * 256 + (int)threadIdx.x;
    const int o0 = o - lane;
    // every load a row needs goes out before the workgroup's one barrier (lba_rows_dev.hpp: pose_cache_fill): slot number, landmark
    // index, observation, the pose matrices' copy into LDS, the landmark
    double T12[12], LM6[6], ob3[3];
    {
        const int oq = o < nobs ? o : nobs - 1;
        const int32_t slot = (lines ? A.ls_slot : A.pt_slot)[oq];
        const size_t l0 = (size_t)(lines ? A.ls_lm : A.pt_lm)[oq];
        if (!lines) {
            const double2 ob = reinterpret_cast<const double2*>(A.uv)[oq];
            ob3[0] = ob.x; ob3[1] = ob.y; ob3[2] = 0.0;
        } else {
            load3(A.lobs + 3 * (size_t)oq, ob3);
        }
        pose_cache_fill(poses, A.T, A.n_slots);
        if (!lines || A.compat_iter) {                       // (the iteration pass's quirk: both end points = the 3 doubles at 3 l0)
            double P3[3];
            load3((lines ? A.Lw : A.Xw) + 3 * l0, P3);
#pragma unroll
            for (int a = 0; a < 3; ++a) LM6[a] = LM6[3 + a] = P3[a];
        } else {
            load6(A.Lw + 6 * l0, LM6);
        }
        pose12_take(poses, A.T, A.n_slots, slot, T12);       // (every wave of the workgroup: a barrier inside)
    }
    double e2w = 0.0;
    if (o0 < nobs) {                                         // (wave-uniform)
        const int valid = nobs - o0 < 64 ? nobs - o0 : 64;
        const int oc = o < nobs ? o : nobs - 1;              // tail lanes recompute the last row
        double nrm, wgt;
        if (!lines) {
            double out6[6], out3[3];
            double2 ob;
            ob.x = ob3[0]; ob.y = ob3[1];
            point_row(A.K, A.th, T12, LM6, ob, out6, out3, nrm, wgt);
            wave_store_rows<6>(A.pJp + 6 * (size_t)o0, out6, slabs[wave], lane, valid);
            wave_store_rows<3>(A.pJl + 3 * (size_t)o0, out3, slabs[wave], lane, valid);
            if (o < nobs) {
                A.pr[o] = nrm;
                A.pw[o] = wgt;
            }
            const bool opt = A.pt_kf_loc[oc] >= 0;             // kf_loc == -1: the keyframe is not optimised, no cross block
            double w18[18];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 6; ++b) w18[a * 6 + b] = opt ? out3[a] * out6[b] * wgt : 0.0;
            wave_store_rows<18>(A.Wp + 18 * (size_t)o0, w18, slabs[wave], lane, valid);
        } else {
            double outl[6], outp[6];
            line_row(A.K, A.th, T12, LM6, LM6 + 3, ob3[0], ob3[1], ob3[2], outl, outp, nrm, wgt);
            wave_store_rows<6>(A.lJl + 6 * (size_t)o0, outl, slabs[wave], lane, valid);
            wave_store_rows<6>(A.lJp + 6 * (size_t)o0, outp, slabs[wave], lane, valid);
            if (o < nobs) {
                A.lr[o] = nrm;
                A.lw[o] = wgt;
            }
            const bool opt = A.ls_kf_loc[oc] >= 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {             // rows 0-2, then rows 3-5 of the 6 x 6 block
                double w18[18];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 6; ++b) {
                        const int r = 3 * half + a;            // (transposed: entry (r, b) of what is stored is outl[b] * outp[r])
                        w18[a * 6 + b] = !opt ? 0.0 : A.transpose_ls_cross ? outl[b] * outp[r] * wgt : outl[r] * outp[b] * wgt;
                    }
                wave_store_row_parts<18, 36>(A.Wl + 36 * (size_t)o0 + 18 * half, w18, slabs[wave], lane, valid);
            }
        }
        if (o < nobs) e2w = nrm * nrm * wgt;
    }
    red[threadIdx.x] = e2w;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) A.err_part[blockIdx.x] = red[0];
}

template <int DL>
__device__ __forceinline__ void schur_landmark(const double* __restrict__ H, const double* __restrict__ g, int j, double lambda,
                                               double* __restrict__ Vinv, double* __restrict__ t, int32_t* __restrict__ nsing);

// One lane per landmark, 64 landmarks per workgroup (a wave: the 2 000 lines of C3 were 8 workgroups of 256 on 8 CUs).  The blocks
// leave through an LDS slab in runs (wave_store_rows): a lane storing its own 36 doubles word by word is 36 instructions of 64
// scattered 8-byte stores.  Lanes past the last landmark replay it and store nothing.
template <int DL>
__device__ __forceinline__ void landmark_block(int l0 /* the wave's first landmark */, int32_t nlm, const int32_t* __restrict__ lm_ptr,
                                               const int32_t* __restrict__ lm_obs, const double* __restrict__ Jl,
                                               const double* __restrict__ r, const double* __restrict__ w, double* __restrict__ Hll,
                                               double* __restrict__ gl, double* __restrict__ slab /* [64 * DL * DL] */, double lambda,
                                               double* __restrict__ Vinv, double* __restrict__ t, int32_t* __restrict__ nsing)
{
    const int lane = threadIdx.x;
    const int valid = nlm - l0 < 64 ? nlm - l0 : 64;
    const bool live = lane < valid;
    const int l = live ? l0 + lane : nlm - 1;
    double H[DL * DL], g[DL];
#pragma unroll
    for (int i = 0; i < DL * DL; ++i) H[i] = 0.0;
#pragma unroll
    for (int i = 0; i < DL; ++i) g[i] = 0.0;
    // (the observations go PB at a time: their ids in one round trip, their rows in a second -- one by one a landmark's list was a
    // chain of two dependent round trips per observation; the sums keep their order)
    constexpr int PB = DL == 3 ? 8 : 4;
    const int kbeg = lm_ptr[l], kend = lm_ptr[l + 1];
    for (int k0 = kbeg; k0 < kend; k0 += PB) {
        int oo[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) oo[u] = lm_obs[k0 + u < kend ? k0 + u : kend - 1];
        double J[PB][DL], rr[PB], ww[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
#pragma unroll
            for (int a = 0; a < DL; ++a) J[u][a] = Jl[(size_t)oo[u] * DL + a];
            rr[u] = r[oo[u]];
            ww[u] = w[oo[u]];
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            if (k0 + u < kend) {
#pragma unroll
                for (int a = 0; a < DL; ++a) g[a] += J[u][a] * rr[u] * ww[u];
#pragma unroll
                for (int a = 0; a < DL; ++a)
#pragma unroll
                    for (int b = 0; b < DL; ++b) H[a * DL + b] += J[u][a] * J[u][b] * ww[u];
            }
        }
    }
    wave_store_rows<DL * DL>(Hll + (size_t)l0 * DL * DL, H, slab, lane, valid);
    wave_store_rows<DL>(gl + (size_t)l0 * DL, g, slab, lane, valid);
    // plslam_lba_plan_iterate_schur: lambda is known when the blocks are built, so the landmark's damped inverse and t = Vinv g
    // (K20's work) follow from the registers -- the same arithmetic on the same words, one launch and one round trip less
    if (Vinv) {                                                  // (kernel argument: uniform)
        double Vo[DL * DL], to[DL];
        schur_landmark<DL>(H, g, 0, lambda, Vo, to, live ? nsing : nullptr);
        wave_store_rows<DL * DL>(Vinv + (size_t)l0 * DL * DL, Vo, slab, lane, valid);
        wave_store_rows<DL>(t + (size_t)l0 * DL, to, slab, lane, valid);
    }
}

struct LbaBlockArgs {
    const int32_t *pt_ptr, *pt_ids, *ls_ptr, *ls_ids, *kf_ptr, *kf_ids;
    const double *pJp, *pJl, *pr, *pw, *lJp, *lJl, *lr, *lw;
    double *H_pt, *g_pt, *H_ls, *g_ls, *pose_part, *H_pose, *g_pose, *err;
    const double* err_part;
    int32_t npt, nls, nkf, np, nb3, nb6, max_chunks, nerr;
    // the Schur step's landmark inverses in the same launch (Vp = nullptr: not asked for)
    double lambda = 0.0;
    double *Vp = nullptr, *tp = nullptr, *Vl = nullptr, *tl = nullptr;
    int32_t* nsing = nullptr;
};

#ifndef PLSLAM_CHUNK_PB
#define PLSLAM_CHUNK_PB 32
#endif
__global__ void __launch_bounds__(256)
k_lba_blocks(const LbaBlockArgs A)
{
    __shared__ __attribute__((aligned(16))) double slab[64 * 36];
    const int b = blockIdx.x;
#ifdef PLSLAM_BLOCKS_X            // timing experiments (tools/r6_blocks_knockouts.sh): 1 no point landmarks, 2 no line landmarks, 4 no keyframe chunk
                                  // partials.  At C3 the launch is 16.3 us; points alone 5.1, lines alone 8.9, the chunk partials alone 13.9 --
                                  // the launch's critical path, and not a matter of round trips (16 / 32 observations in flight: 16.3 / 16.1 us)
    if ((PLSLAM_BLOCKS_X & 1) && b < A.nb3) return;
    if ((PLSLAM_BLOCKS_X & 2) && b >= A.nb3 && b < A.nb3 + A.nb6) return;
    if ((PLSLAM_BLOCKS_X & 4) && b >= A.nb3 + A.nb6) return;
#endif
    if (b < A.nb3) {                                   // (64 landmarks: the workgroup's first wave; the others leave)
        if (threadIdx.x >= 64) return;
        landmark_block<3>(b * 64, A.npt, A.pt_ptr, A.pt_ids, A.pJl, A.pr, A.pw, A.H_pt, A.g_pt, slab, A.lambda, A.Vp, A.tp, A.nsing);
    } else if (b < A.nb3 + A.nb6) {
        if (threadIdx.x >= 64) return;
        landmark_block<6>((b - A.nb3) * 64, A.nls, A.ls_ptr, A.ls_ids, A.lJl, A.lr, A.lw, A.H_ls, A.g_ls, slab, A.lambda, A.Vl, A.tl,
                          A.nsing);
    } else {
        // chunk partials of the keyframes (K9): item = keyframe * max_chunks + chunk, one wave per item, lane e < 42 an entry
        // (the item is the wave's: said so, the observation ids are scalar loads)
        const int item = __builtin_amdgcn_readfirstlane((b - A.nb3 - A.nb6) * 4 + ((int)threadIdx.x >> 6)), e = (int)threadIdx.x & 63;
        if (item >= A.nkf * A.max_chunks || e >= 42) return;
        const int k = item / A.max_chunks, c = item - k * A.max_chunks;
        const int beg = A.kf_ptr[k] + c * POSE_CHUNK;
        const int end = beg + POSE_CHUNK < A.kf_ptr[k + 1] ? beg + POSE_CHUNK : A.kf_ptr[k + 1];
        const int a = e < 36 ? e / 6 : e - 36, bb = e < 36 ? e % 6 : 0;
        double acc = 0.0;
        constexpr int PB = PLSLAM_CHUNK_PB;   // (observations whose loads are in flight together: 32 = two round trips per 64-observation chunk; 16 was four, and this part of the launch its critical path: 13.9 us alone)
        for (int i0 = beg; i0 < end; i0 += PB) {
            int oo[PB];
            bool pt[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int o = A.kf_ids[i0 + j < end ? i0 + j : end - 1];
                pt[j] = o < A.np;
                oo[j] = pt[j] ? o : o - A.np;
            }
            double ja[PB], jb[PB], ww[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const double* J = (pt[j] ? A.pJp : A.lJp) + (size_t)oo[j] * 6;
                ja[j] = J[a];
                jb[j] = e < 36 ? J[bb] : (pt[j] ? A.pr : A.lr)[oo[j]];
                ww[j] = (pt[j] ? A.pw : A.lw)[oo[j]];
            }
#pragma unroll
            for (int j = 0; j < PB; ++j)
                if (i0 + j < end) acc += ja[j] * jb[j] * ww[j];
        }
        A.pose_part[((size_t)k * A.max_chunks + c) * 42 + e] = acc;   // empty chunks write 0
    }
}

#ifndef PLSLAM_SCH_WAVES
#define PLSLAM_SCH_WAVES 1                 // chunks per workgroup of the Schur partials' launch (see wave_sync)
#endif
// NT = 256: a lane per partial-sum slot; NT = 64 (inside the Schur partials' launch): a lane plays the four lanes e, e + 64,
// e + 128, e + 192 of the 256-lane form and adds them as its tree's first two levels do -- the same sums in the same order
template <int NT>
__device__ __forceinline__ void lba_finish_wg(const LbaBlockArgs& A, int k, double* __restrict__ red /* [NT] */)
{
    static_assert(NT == 256 || NT == 64, "256 lanes, or 64 lanes playing four each");
    const int e = threadIdx.x;
    if (k < A.nkf) {
        if (e >= 42) return;
        const int nchunks = (A.kf_ptr[k + 1] - A.kf_ptr[k] + POSE_CHUNK - 1) / POSE_CHUNK;
        double acc = 0.0;
        constexpr int PB = 32;                     // (loads in flight per round trip: C3's 105 chunks were 14 round trips at 8)
        for (int c0 = 0; c0 < nchunks; c0 += PB) {
            double v[PB];
#pragma unroll
            for (int j = 0; j < PB; ++j) v[j] = A.pose_part[((size_t)k * A.max_chunks + (c0 + j < nchunks ? c0 + j : nchunks - 1)) * 42 + e];
#pragma unroll
            for (int j = 0; j < PB; ++j)
                if (c0 + j < nchunks) acc += v[j];
        }
        if (e < 36) A.H_pose[(size_t)k * 36 + e] = acc;
        else A.g_pose[(size_t)k * 6 + (e - 36)] = acc;
        return;
    }
    // err: lane e sums the row workgroups' partials e, e + 256, ... in that order, then a tree over the lanes
    constexpr int Q = 256 / NT;
    double acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        acc[q] = 0.0;
        for (int i = e + q * NT; i < A.nerr; i += 256) acc[q] += A.err_part[i];
    }
    if constexpr (Q == 4) { acc[0] += acc[2]; acc[1] += acc[3]; acc[0] += acc[1]; }      // the tree's levels 128 and 64
    auto sync = [] {
        if (NT == 64 && PLSLAM_SCH_WAVES != 1) {   // (one wave, inside a workgroup whose other waves are elsewhere)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
    };
    red[e] = acc[0];
    sync();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (e < s) red[e] += red[e + s];
        sync();
    }
    if (e == 0) A.err[0] = red[0];
}

__global__ void __launch_bounds__(256)
k_lba_finish(const LbaBlockArgs A)
{
    __shared__ double red[256];
    lba_finish_wg<256>(A, (int)blockIdx.x, red);
}

namespace {
struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off += (bytes + 255) & ~size_t(255); return o; }
};
}  // namespace
}  // namespace plslam

namespace plslam {

// all pointers on the device; CSR lists as built by build_csr().  Asynchronous on `s`.
struct AssembleDev {
    const int32_t *pt_kf_loc, *ls_kf_loc, *pt_ptr, *pt_ids, *ls_ptr, *ls_ids, *kf_ptr, *kf_ids;
    const double *pt_Jp, *pt_Jl, *pt_r, *pt_w, *ls_Jp, *ls_Jl, *ls_r, *ls_w;
    double *g, *H_pose, *H_pt, *H_ls, *W_pt, *W_ls, *err;
    double* pose_part;      // [nkf][max_chunks][42] scratch
    int32_t max_chunks;     // max over keyframes of ceil(#observations / POSE_CHUNK)
    int32_t transpose_ls_cross = 0;   // PLSLAM_LBA_COMPAT_GBA: the pose x line cross blocks as the reference's GBA writes them
};

static int assemble_on_device(const AssembleDev& a, int32_t nkf, int32_t npt, int32_t nls, int32_t n_pt_obs,
                              int32_t n_ls_obs, hipStream_t s)
{
    // g layout = the reference's X layout: [6*nkf poses | 3*npt points | 6*nls lines]
    if (npt)
        hipLaunchKernelGGL(k_landmark_blocks<3>, dim3((npt + 255) / 256), dim3(256), 0, s, a.pt_ptr, a.pt_ids, npt,
                           a.pt_Jl, a.pt_r, a.pt_w, a.H_pt, a.g + 6 * (size_t)nkf);
    if (nls)
        hipLaunchKernelGGL(k_landmark_blocks<6>, dim3((nls + 255) / 256), dim3(256), 0, s, a.ls_ptr, a.ls_ids, nls,
                           a.ls_Jl, a.ls_r, a.ls_w, a.H_ls, a.g + 6 * (size_t)nkf + 3 * (size_t)npt);
    if (n_pt_obs)
        hipLaunchKernelGGL(k_cross_blocks<3>, dim3((n_pt_obs + 255) / 256), dim3(256), 0, s, a.pt_kf_loc, n_pt_obs,
                           a.pt_Jp, a.pt_Jl, a.pt_w, a.W_pt);
    if (n_ls_obs && !a.transpose_ls_cross)
        hipLaunchKernelGGL(k_cross_blocks<6>, dim3((n_ls_obs + 255) / 256), dim3(256), 0, s, a.ls_kf_loc, n_ls_obs,
                           a.ls_Jp, a.ls_Jl, a.ls_w, a.W_ls);
    if (n_ls_obs && a.transpose_ls_cross)
        hipLaunchKernelGGL((k_cross_blocks<6, true>), dim3((n_ls_obs + 255) / 256), dim3(256), 0, s, a.ls_kf_loc, n_ls_obs,
                           a.ls_Jp, a.ls_Jl, a.ls_w, a.W_ls);
    if (nkf) {
        if (a.max_chunks > 0)
            hipLaunchKernelGGL(k_pose_partials, dim3(nkf, a.max_chunks), dim3(64), 0, s, a.kf_ptr, a.kf_ids, n_pt_obs,
                               a.pt_Jp, a.pt_r, a.pt_w, a.ls_Jp, a.ls_r, a.ls_w, a.max_chunks, a.pose_part);
        hipLaunchKernelGGL(k_pose_blocks, dim3(nkf), dim3(64), 0, s, a.kf_ptr, a.max_chunks, a.pose_part, a.H_pose, a.g);
    }
    hipLaunchKernelGGL(k_weighted_error_partials, dim3(ERR_BLOCKS), dim3(256), 0, s, a.pt_r, a.pt_w, n_pt_obs, a.ls_r, a.ls_w,
                       n_ls_obs, a.err);
    hipLaunchKernelGGL(k_weighted_error_final, dim3(1), dim3(ERR_BLOCKS), 0, s, a.err);
    PLSLAM_HIP_CHECK(hipGetLastError());
    return PLSLAM_OK;
}

// stable CSR lists (host, O(nobs)): observations per landmark and per keyframe, list order; keyframe
// lists hold points first, then lines (global ids: points [0, n_pt_obs), lines n_pt_obs + o)
struct CsrLists { std::vector<int32_t> ptp, pti, lsp, lsi, kfp, kfi; };
static void build_csr(const int32_t* pt_lm, const int32_t* pt_kf, int32_t np, const int32_t* ls_lm,
                      const int32_t* ls_kf, int32_t nl, int32_t nkf, int32_t npt, int32_t nls, CsrLists& c)
{
    auto by = [](const int32_t* key, int32_t n, int32_t nkeys, std::vector<int32_t>& ptr, std::vector<int32_t>& ids) {
        ptr.assign((size_t)nkeys + 1, 0);
        for (int32_t o = 0; o < n; ++o) ++ptr[key[o] + 1];
        for (int32_t k = 0; k < nkeys; ++k) ptr[k + 1] += ptr[k];
        ids.assign((size_t)ptr[nkeys], 0);
        std::vector<int32_t> pos(ptr.begin(), ptr.end() - 1);
        for (int32_t o = 0; o < n; ++o) ids[pos[key[o]]++] = o;
    };
    by(pt_lm, np, npt, c.ptp, c.pti);
    by(ls_lm, nl, nls, c.lsp, c.lsi);
    c.kfp.assign((size_t)nkf + 1, 0);
    for (int32_t o = 0; o < np; ++o) if (pt_kf[o] >= 0) ++c.kfp[pt_kf[o] + 1];
    for (int32_t o = 0; o < nl; ++o) if (ls_kf[o] >= 0) ++c.kfp[ls_kf[o] + 1];
    for (int32_t k = 0; k < nkf; ++k) c.kfp[k + 1] += c.kfp[k];
    c.kfi.assign((size_t)c.kfp[nkf], 0);
    std::vector<int32_t> pos(c.kfp.begin(), c.kfp.end() - 1);
    for (int32_t o = 0; o < np; ++o) if (pt_kf[o] >= 0) c.kfi[pos[pt_kf[o]]++] = o;
    for (int32_t o = 0; o < nl; ++o) if (ls_kf[o] >= 0) c.kfi[pos[ls_kf[o]]++] = np + o;
}

static int32_t pose_max_chunks(const std::vector<int32_t>& kfp)
{
    int32_t m = 0;
    for (size_t k = 0; k + 1 < kfp.size(); ++k) m = std::max(m, (kfp[k + 1] - kfp[k] + POSE_CHUNK - 1) / POSE_CHUNK);
    return m;
}

}  // namespace plslam

using namespace plslam;

// ---------------------------------------------------------------------------------------------
// LBA plan: the LM loop of levMarquardtOptimizationLBA rebuilds rows + H/g up to max_iters_lba = 15
// times per call (src/mapHandler.cpp:1358-1540 once, :1587-1772 per iteration) while only X (poses,
// landmarks) changes.  The plan uploads the observation lists, the observations and the CSR lists
// once; iterate() uploads X, runs K3/K4 and K7-K10 device-resident and downloads the blocks.
// ---------------------------------------------------------------------------------------------
struct plslam_lba_plan {
    plslam_ctx* ctx = nullptr;
    plslam_cam cam{};
    double th = 0;
    int32_t n_slots = 0, nkf = 0, npt = 0, nls = 0, np = 0, nl = 0;
    DevBuf stat, dyn, rows, out;   // static lists / X / row arrays / blocks
    // offsets
    size_t oPlm = 0, oPslot = 0, oPkf = 0, oPuv = 0, oLlm = 0, oLslot = 0, oLkf = 0, oLobs = 0, oPtp = 0, oPti = 0,
           oLsp = 0, oLsi = 0, oKfp = 0, oKfi = 0;
    size_t oT = 0, oX = 0, oL = 0;
    size_t oPJp = 0, oPJl = 0, oPr = 0, oPw = 0, oLJp = 0, oLJl = 0, oLr = 0, oLw = 0;
    size_t oG = 0, oHp = 0, oHpt = 0, oHls = 0, oWp = 0, oWl = 0, oErr = 0, oPart = 0, oErrPart = 0;
    int32_t max_chunks = 0;
    // one iteration = one upload, three launches, one download: the poses and landmarks are packed into a page-locked image
    // of `dyn` (ONE copy instead of three from pageable memory), err sits right behind g (ONE copy back)
    HostBuf pin_in, pin_out;
    size_t dyn_bytes = 0;
    bool state_valid = false;      // T / Xw / Lw have been uploaded at least once (iterate_resident needs them)
    bool blocks_valid = false;     // an iteration has left H / g / W on the device (the Schur step consumes them)
    bool blocks_gba = false;       // ... with the pose x line cross blocks transposed (PLSLAM_LBA_COMPAT_GBA): not what the Schur step reads
    // ---- the Schur step (round 5): pair lists built on first use from these host copies of the observation lists
    CsrLists csr;
    std::vector<int32_t> h_pt_kf, h_ls_kf;
    DevBuf schur;
    HostBuf schur_pin;
    char* schur_pin_dev = nullptr;   // the device address of schur_pin (mapped page-locked memory), or nullptr: kernels write S, b in place
    bool schur_ready = false, schur_done = false;
    int schur_parity = 0;              // which of the two counters of singular landmarks the next plslam_lba_plan_schur counts in
    int32_t nblk = 0, schur_chunks = 0;
    size_t oSpair = 0, oSblk = 0, oVp = 0, oVl = 0, oTp = 0, oTl = 0, oSpart = 0, oBpart = 0, oS = 0, oDp = 0, oDx = 0, oDxPart = 0, oSing = 0;
};

extern "C" int plslam_lba_plan_create(plslam_ctx* ctx, const plslam_cam* K, double homog_th, int32_t n_pose_slots,
                                      int32_t nkf, int32_t npt, int32_t nls, const int32_t* pt_lm_loc,
                                      const int32_t* pt_pose_slot, const int32_t* pt_kf_loc, const double* pt_obs_uv,
                                      int32_t n_pt_obs, const int32_t* ls_lm_loc, const int32_t* ls_pose_slot,
                                      const int32_t* ls_kf_loc, const double* ls_l_obs, int32_t n_ls_obs,
                                      plslam_lba_plan** out)
{
    PLSLAM_REQUIRE(ctx && K && out && n_pose_slots >= 0 && nkf >= 0 && npt >= 0 && nls >= 0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(n_pt_obs >= 0 && n_ls_obs >= 0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(n_pt_obs == 0 || (pt_lm_loc && pt_pose_slot && pt_kf_loc && pt_obs_uv), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(n_ls_obs == 0 || (ls_lm_loc && ls_pose_slot && ls_kf_loc && ls_l_obs), PLSLAM_EINVAL);
    *out = nullptr;
    for (int32_t o = 0; o < n_pt_obs; ++o)
        PLSLAM_REQUIRE(pt_lm_loc[o] >= 0 && pt_lm_loc[o] < npt && pt_kf_loc[o] >= -1 && pt_kf_loc[o] < nkf &&
                       pt_pose_slot[o] >= 0 && pt_pose_slot[o] < n_pose_slots, PLSLAM_EINVAL);
    for (int32_t o = 0; o < n_ls_obs; ++o)
        PLSLAM_REQUIRE(ls_lm_loc[o] >= 0 && ls_lm_loc[o] < nls && ls_kf_loc[o] >= -1 && ls_kf_loc[o] < nkf &&
                       ls_pose_slot[o] >= 0 && ls_pose_slot[o] < n_pose_slots, PLSLAM_EINVAL);
    plslam_lba_plan* P = new (std::nothrow) plslam_lba_plan();
    PLSLAM_REQUIRE(P != nullptr, PLSLAM_ENOMEM);
    P->ctx = ctx; P->cam = *K; P->th = homog_th; P->n_slots = n_pose_slots; P->nkf = nkf; P->npt = npt; P->nls = nls;
    P->np = n_pt_obs; P->nl = n_ls_obs;
    CsrLists& c = P->csr;
    build_csr(pt_lm_loc, pt_kf_loc, n_pt_obs, ls_lm_loc, ls_kf_loc, n_ls_obs, nkf, npt, nls, c);
    if (n_pt_obs) P->h_pt_kf.assign(pt_kf_loc, pt_kf_loc + n_pt_obs);
    if (n_ls_obs) P->h_ls_kf.assign(ls_kf_loc, ls_kf_loc + n_ls_obs);
    const size_t np = (size_t)n_pt_obs, nl = (size_t)n_ls_obs;
    Carve cs;
    P->oPlm = cs.take(np * 4); P->oPslot = cs.take(np * 4); P->oPkf = cs.take(np * 4); P->oPuv = cs.take(np * 16);
    P->oLlm = cs.take(nl * 4); P->oLslot = cs.take(nl * 4); P->oLkf = cs.take(nl * 4); P->oLobs = cs.take(nl * 24);
    P->oPtp = cs.take(c.ptp.size() * 4); P->oPti = cs.take(c.pti.size() * 4 + 4); P->oLsp = cs.take(c.lsp.size() * 4);
    P->oLsi = cs.take(c.lsi.size() * 4 + 4); P->oKfp = cs.take(c.kfp.size() * 4); P->oKfi = cs.take(c.kfi.size() * 4 + 4);
    Carve cd;
    P->oT = cd.take((size_t)n_pose_slots * 128 + 8); P->oX = cd.take((size_t)npt * 24 + 8); P->oL = cd.take((size_t)nls * 48 + 8);
    Carve cr;
    P->oPJp = cr.take(np * 48 + 8); P->oPJl = cr.take(np * 24 + 8); P->oPr = cr.take(np * 8 + 8); P->oPw = cr.take(np * 8 + 8);
    P->oLJp = cr.take(nl * 48 + 8); P->oLJl = cr.take(nl * 48 + 8); P->oLr = cr.take(nl * 8 + 8); P->oLw = cr.take(nl * 8 + 8);
    const size_t N = 6 * (size_t)nkf + 3 * (size_t)npt + 6 * (size_t)nls;
    Carve co;
    P->oG = co.take(N * 8 + 8); P->oHp = co.take((size_t)nkf * 288 + 8); P->oHpt = co.take((size_t)npt * 72 + 8);
    P->oHls = co.take((size_t)nls * 288 + 8); P->oWp = co.take(np * 144 + 8); P->oWl = co.take(nl * 288 + 8);
    P->oErr = P->oG + N * 8;                      // err: the double right behind g (one copy brings both back)
    P->oErrPart = co.take(8 * ((np + 255) / 256 + (nl + 255) / 256) + 8);      // the row workgroups' partial sums of err
    P->max_chunks = pose_max_chunks(c.kfp);
    P->oPart = co.take((size_t)nkf * (size_t)P->max_chunks * 42 * 8 + 8);
    int rc;
    P->dyn_bytes = cd.off;
    if ((rc = P->stat.reserve(cs.off + 256)) || (rc = P->dyn.reserve(cd.off + 256)) ||
        (rc = P->rows.reserve(cr.off + 256)) || (rc = P->out.reserve(co.off + 256)) ||
        (rc = P->pin_in.reserve(cd.off + 256)) || (rc = P->pin_out.reserve(N * 8 + 256))) {
        P->stat.release(); P->dyn.release(); P->rows.release(); P->out.release(); P->pin_in.release(); P->pin_out.release();
        delete P;
        return rc;
    }
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);    // every entry point runs on the context's device, whatever the calling thread's current one
    hipStream_t s = ctx->stream;
    char* d = P->stat.as<char>();
    auto up = [&](size_t off, const void* src, size_t bytes) -> int {
        if (bytes) PLSLAM_HIP_CHECK(hipMemcpyAsync(d + off, src, bytes, hipMemcpyHostToDevice, s));
        return PLSLAM_OK;
    };
    if ((rc = up(P->oPlm, pt_lm_loc, np * 4)) || (rc = up(P->oPslot, pt_pose_slot, np * 4)) ||
        (rc = up(P->oPkf, pt_kf_loc, np * 4)) || (rc = up(P->oPuv, pt_obs_uv, np * 16)) ||
        (rc = up(P->oLlm, ls_lm_loc, nl * 4)) || (rc = up(P->oLslot, ls_pose_slot, nl * 4)) ||
        (rc = up(P->oLkf, ls_kf_loc, nl * 4)) || (rc = up(P->oLobs, ls_l_obs, nl * 24)) ||
        (rc = up(P->oPtp, c.ptp.data(), c.ptp.size() * 4)) || (rc = up(P->oPti, c.pti.data(), c.pti.size() * 4)) ||
        (rc = up(P->oLsp, c.lsp.data(), c.lsp.size() * 4)) || (rc = up(P->oLsi, c.lsi.data(), c.lsi.size() * 4)) ||
        (rc = up(P->oKfp, c.kfp.data(), c.kfp.size() * 4)) || (rc = up(P->oKfi, c.kfi.data(), c.kfi.size() * 4)) ||
        (hipStreamSynchronize(s) != hipSuccess && (rc = PLSLAM_EHIP))) {   // the staging vectors die here
        (void)hipStreamSynchronize(s);
        P->stat.release(); P->dyn.release(); P->rows.release(); P->out.release(); P->pin_in.release(); P->pin_out.release();
        delete P;
        return rc;
    }
    *out = P;
    return PLSLAM_OK;
}

static LbaBlockArgs lba_block_args(plslam_lba_plan* P)
{
    char *ds = P->stat.as<char>(), *dr = P->rows.as<char>(), *dout = P->out.as<char>();
    const size_t N6 = 6 * (size_t)P->nkf;
    LbaBlockArgs B{};
    B.pt_ptr = (int32_t*)(ds + P->oPtp); B.pt_ids = (int32_t*)(ds + P->oPti); B.ls_ptr = (int32_t*)(ds + P->oLsp);
    B.ls_ids = (int32_t*)(ds + P->oLsi); B.kf_ptr = (int32_t*)(ds + P->oKfp); B.kf_ids = (int32_t*)(ds + P->oKfi);
    B.pJp = (double*)(dr + P->oPJp); B.pJl = (double*)(dr + P->oPJl); B.pr = (double*)(dr + P->oPr); B.pw = (double*)(dr + P->oPw);
    B.lJp = (double*)(dr + P->oLJp); B.lJl = (double*)(dr + P->oLJl); B.lr = (double*)(dr + P->oLr); B.lw = (double*)(dr + P->oLw);
    double* g = (double*)(dout + P->oG);
    B.H_pt = (double*)(dout + P->oHpt); B.g_pt = g + N6; B.H_ls = (double*)(dout + P->oHls); B.g_ls = g + N6 + 3 * (size_t)P->npt;
    B.pose_part = (double*)(dout + P->oPart); B.H_pose = (double*)(dout + P->oHp); B.g_pose = g;
    B.err = (double*)(dout + P->oErr); B.err_part = (double*)(dout + P->oErrPart);
    B.npt = P->npt; B.nls = P->nls; B.nkf = P->nkf; B.np = P->np;
    B.nb3 = (P->npt + 63) / 64; B.nb6 = (P->nls + 63) / 64; B.max_chunks = P->max_chunks;
    B.nerr = (P->np + 255) / 256 + (P->nl + 255) / 256;
    return B;
}

// upload X (one copy), rows + cross blocks + err partials (F1), landmark blocks + keyframe chunk partials (F2), keyframe blocks +
// err (F3): enqueued on the context's stream, nothing downloaded.  Caller holds ctx->mu.
// upload = false: the poses and landmarks already on the device are used (plslam_lba_plan_iterate_resident: a device-side
// solver has updated them in place)
// fused_lambda >= 0 (plslam_lba_plan_iterate_schur; the Schur step's buffers exist): the landmark inverses for that damping are
// written by the blocks' launch, and the last stage (K10) is NOT launched here -- it rides in the Schur partials' launch
// (lba_schur_enqueue(fused)), which the caller enqueues next
static int lba_plan_enqueue(plslam_lba_plan* P, const double* T_kf_w, const double* Xw, const double* Lw, int compat_flags,
                            bool upload = true, double fused_lambda = -1.0)
{
    plslam_ctx* ctx = P->ctx;
    hipStream_t s = ctx->stream;
    char *ds = P->stat.as<char>(), *dd = P->dyn.as<char>(), *dr = P->rows.as<char>(), *dout = P->out.as<char>();
    if (upload) {
        // (the page-locked image is rewritten per call: the previous call's copy has completed -- every caller synchronises
        // the stream before it returns)
        char* hi = P->pin_in.as<char>();
        // (a caller that keeps its state IN the image -- plslam_lba_plan_host_state -- passes the image's own pointers: no copy)
        if (P->n_slots && (const char*)T_kf_w != hi + P->oT) memcpy(hi + P->oT, T_kf_w, (size_t)P->n_slots * 128);
        if (P->npt && (const char*)Xw != hi + P->oX) memcpy(hi + P->oX, Xw, (size_t)P->npt * 24);
        if (P->nls && (const char*)Lw != hi + P->oL) memcpy(hi + P->oL, Lw, (size_t)P->nls * 48);
        if (P->dyn_bytes) PLSLAM_HIP_CHECK(hipMemcpyAsync(dd, hi, P->dyn_bytes, hipMemcpyHostToDevice, s));
        P->state_valid = true;
    }
    const int32_t nbp = (P->np + 255) / 256, nbl = (P->nl + 255) / 256;
    const size_t N6 = 6 * (size_t)P->nkf;
    LbaIterArgs A{};
    A.K = CamD{P->cam.fx, P->cam.fy, P->cam.cx, P->cam.cy, (double)P->cam.width, (double)P->cam.height};
    A.th = P->th;
    A.compat_iter = (compat_flags & PLSLAM_LBA_COMPAT_ITER_PASS) ? 1 : 0;
    A.transpose_ls_cross = (compat_flags & PLSLAM_LBA_COMPAT_GBA) ? 1 : 0;
    A.T = (double*)(dd + P->oT); A.Xw = (double*)(dd + P->oX); A.Lw = (double*)(dd + P->oL);
    A.uv = (double*)(ds + P->oPuv); A.lobs = (double*)(ds + P->oLobs);
    A.pt_lm = (int32_t*)(ds + P->oPlm); A.pt_slot = (int32_t*)(ds + P->oPslot); A.pt_kf_loc = (int32_t*)(ds + P->oPkf);
    A.ls_lm = (int32_t*)(ds + P->oLlm); A.ls_slot = (int32_t*)(ds + P->oLslot); A.ls_kf_loc = (int32_t*)(ds + P->oLkf);
    A.np = P->np; A.nl = P->nl; A.nbp = nbp; A.nbl = nbl; A.n_slots = P->n_slots;
    A.pJp = (double*)(dr + P->oPJp); A.pJl = (double*)(dr + P->oPJl); A.pr = (double*)(dr + P->oPr); A.pw = (double*)(dr + P->oPw);
    A.lJp = (double*)(dr + P->oLJp); A.lJl = (double*)(dr + P->oLJl); A.lr = (double*)(dr + P->oLr); A.lw = (double*)(dr + P->oLw);
    A.Wp = (double*)(dout + P->oWp); A.Wl = (double*)(dout + P->oWl); A.err_part = (double*)(dout + P->oErrPart);
    if (nbp + nbl > 0) hipLaunchKernelGGL(k_lba_rows_cross, dim3(nbp + nbl), dim3(256), 0, s, A);
    LbaBlockArgs B = lba_block_args(P);
    const bool fused = fused_lambda >= 0.0;
    if (fused) {
        char* d = P->schur.as<char>();
        B.lambda = fused_lambda;
        B.Vp = (double*)(d + P->oVp); B.Vl = (double*)(d + P->oVl); B.tp = (double*)(d + P->oTp); B.tl = (double*)(d + P->oTl);
        B.nsing = (int32_t*)((double*)(d + P->oS) + N6 * N6 + N6 + 1) + P->schur_parity;
    }
    const int32_t nchunk_wgs = (P->nkf * P->max_chunks + 3) / 4;
    if (B.nb3 + B.nb6 + nchunk_wgs > 0)
        hipLaunchKernelGGL(k_lba_blocks, dim3(B.nb3 + B.nb6 + nchunk_wgs), dim3(256), 0, s, B);
    if (!fused) hipLaunchKernelGGL(k_lba_finish, dim3(P->nkf + 1), dim3(256), 0, s, B);
    PLSLAM_HIP_CHECK(hipGetLastError());
    P->blocks_valid = true;
    P->blocks_gba = (compat_flags & PLSLAM_LBA_COMPAT_GBA) != 0;
    P->schur_done = false;             // (the blocks have changed: the landmark inverses of an earlier Schur step are stale)
    return PLSLAM_OK;
}

// g (+ err) down by a KERNEL that writes the page-locked block where the device can address it: a copy command of this size runs on
// the copy engine, which starts ~13 us after the kernel in front of it (profiles/r6_r_call_timeline_map2kf_points_fast.txt: the
// same pattern); a kernel starts 2-3 us after, and 0.34 MB of posted writes cross PCIe in ~7 us either way
__global__ void __launch_bounds__(256)
k_copy_out(const double* __restrict__ src, double* __restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// device -> host copies of the blocks of the last iteration (NULL pointers are skipped), then one synchronise
static int lba_plan_download(plslam_lba_plan* P, double* g, double* H_pose, double* H_pt, double* H_ls, double* W_pt,
                             double* W_ls, double* err)
{
    hipStream_t s = P->ctx->stream;
    char* dout = P->out.as<char>();
    const size_t N = 6 * (size_t)P->nkf + 3 * (size_t)P->npt + 6 * (size_t)P->nls;
    auto down = [&](void* dst, size_t off, size_t bytes) -> int {
        if (dst && bytes) PLSLAM_HIP_CHECK(hipMemcpyAsync(dst, dout + off, bytes, hipMemcpyDeviceToHost, s));
        return PLSLAM_OK;
    };
    int rc;
    if (!H_pose && !H_pt && !H_ls && !W_pt && !W_ls) {
        // g and err (or err alone): one copy into page-locked memory, then out of it
        char* ho = P->pin_out.as<char>();
        // err alone lands in the slot it has in the g + err copy, BEHIND the image's g (plslam_lba_plan_host_state hands
        // out ho as g: an err-only iteration -- a rejected LM trial step -- must leave the gradient there untouched)
        const size_t off = g ? P->oG : P->oErr, bytes = g ? N * 8 + 8 : 8;
        if (g && P->pin_out.dev) {
            const size_t nd = N + 1;
            hipLaunchKernelGGL(k_copy_out, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, s, (const double*)(dout + off),
                               static_cast<double*>(P->pin_out.dev), nd);
            PLSLAM_HIP_CHECK(hipGetLastError());
        } else {
            PLSLAM_HIP_CHECK(hipMemcpyAsync(ho + (g ? 0 : N * 8), dout + off, bytes, hipMemcpyDeviceToHost, s));
        }
        PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
        if (g && (char*)g != ho) memcpy(g, ho, N * 8);      // (g = the page-locked image itself: plslam_lba_plan_host_state)
        if (err) memcpy(err, ho + N * 8, 8);
        return PLSLAM_OK;
    }
    if ((rc = down(g, P->oG, N * 8)) || (rc = down(H_pose, P->oHp, (size_t)P->nkf * 288)) ||
        (rc = down(H_pt, P->oHpt, (size_t)P->npt * 72)) || (rc = down(H_ls, P->oHls, (size_t)P->nls * 288)) ||
        (rc = down(W_pt, P->oWp, (size_t)P->np * 144)) || (rc = down(W_ls, P->oWl, (size_t)P->nl * 288)) ||
        (rc = down(err, P->oErr, 8))) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return PLSLAM_OK;
}

extern "C" int plslam_lba_plan_iterate(plslam_lba_plan* P, const double* T_kf_w, const double* Xw, const double* Lw,
                                       int compat_flags, double* g, double* H_pose, double* H_pt, double* H_ls,
                                       double* W_pt, double* W_ls, double* err)
{
    PLSLAM_REQUIRE(P && g && err, PLSLAM_EINVAL);
    PLSLAM_REQUIRE((P->n_slots == 0 || T_kf_w) && (P->npt == 0 || (Xw && H_pt)) && (P->nls == 0 || (Lw && H_ls)), PLSLAM_EINVAL);
    PLSLAM_REQUIRE((P->nkf == 0 || H_pose) && (P->np == 0 || W_pt) && (P->nl == 0 || W_ls), PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);    // every entry point runs on the context's device, whatever the calling thread's current one
    int rc = lba_plan_enqueue(P, T_kf_w, Xw, Lw, compat_flags);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    return lba_plan_download(P, g, H_pose, H_pt, H_ls, W_pt, W_ls, err);
}

// The same iteration with the blocks LEFT ON THE DEVICE: what crosses PCIe is X up (0.34 MB at C3) and err (+ g when
// asked for) down, not the 11.5 MB of blocks.  plslam_lba_plan_device_blocks names them for a device-side solver;
// plslam_lba_plan_blocks fetches any of them later.
extern "C" int plslam_lba_plan_iterate_dev(plslam_lba_plan* P, const double* T_kf_w, const double* Xw, const double* Lw,
                                           int compat_flags, double* g, double* err)
{
    PLSLAM_REQUIRE(P && err, PLSLAM_EINVAL);
    PLSLAM_REQUIRE((P->n_slots == 0 || T_kf_w) && (P->npt == 0 || Xw) && (P->nls == 0 || Lw), PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);
    int rc = lba_plan_enqueue(P, T_kf_w, Xw, Lw, compat_flags);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    return lba_plan_download(P, g, nullptr, nullptr, nullptr, nullptr, nullptr, err);
}

// The iteration on the state that already lives on the device: nothing goes up, err comes down.  For a device-side solver
// (plslam_lba_plan_device_state names T / Xw / Lw: it updates them in place on the context's stream between iterations).
extern "C" int plslam_lba_plan_iterate_resident(plslam_lba_plan* P, int compat_flags, double* err)
{
    PLSLAM_REQUIRE(P && err, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(P->state_valid, PLSLAM_EINVAL);         // one plslam_lba_plan_iterate(_dev) first: it uploads the state
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);
    int rc = lba_plan_enqueue(P, nullptr, nullptr, nullptr, compat_flags, false);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    return lba_plan_download(P, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, err);
}

extern "C" int plslam_lba_plan_device_state(plslam_lba_plan* P, plslam_lba_state* out)
{
    PLSLAM_REQUIRE(P && out, PLSLAM_EINVAL);
    char* dd = P->dyn.as<char>();
    out->T_kf_w = (double*)(dd + P->oT); out->Xw = (double*)(dd + P->oX); out->Lw = (double*)(dd + P->oL);
    out->n_pose_slots = P->n_slots; out->npt = P->npt; out->nls = P->nls;
    out->stream = P->ctx->stream;
    return PLSLAM_OK;
}

// The plan's page-locked images by name: a host solver that keeps T / Xw / Lw there and reads g from there hands these very
// pointers to plslam_lba_plan_iterate / _iterate_dev, which then skip their staging copies (0.34 MB in, 0.34 MB out at C3).
extern "C" int plslam_lba_plan_host_state(plslam_lba_plan* P, plslam_lba_host_state* out)
{
    PLSLAM_REQUIRE(P && out, PLSLAM_EINVAL);
    char *hi = P->pin_in.as<char>(), *ho = P->pin_out.as<char>();      // (fixed at plan creation: nothing to lock)
    out->T_kf_w = (double*)(hi + P->oT); out->Xw = (double*)(hi + P->oX); out->Lw = (double*)(hi + P->oL);
    out->g = (double*)ho;
    out->n_pose_slots = P->n_slots; out->npt = P->npt; out->nls = P->nls;
    out->n = (int64_t)(6 * (size_t)P->nkf + 3 * (size_t)P->npt + 6 * (size_t)P->nls);
    return PLSLAM_OK;
}

extern "C" int plslam_lba_plan_device_blocks(plslam_lba_plan* P, plslam_lba_blocks* out)
{
    PLSLAM_REQUIRE(P && out, PLSLAM_EINVAL);
    char* dout = P->out.as<char>();
    out->g = (const double*)(dout + P->oG); out->H_pose = (const double*)(dout + P->oHp);
    out->H_pt = (const double*)(dout + P->oHpt); out->H_ls = (const double*)(dout + P->oHls);
    out->W_pt = (const double*)(dout + P->oWp); out->W_ls = (const double*)(dout + P->oWl);
    out->err = (const double*)(dout + P->oErr);
    out->stream = P->ctx->stream;
    return PLSLAM_OK;
}

extern "C" int plslam_lba_plan_blocks(plslam_lba_plan* P, double* g, double* H_pose, double* H_pt, double* H_ls,
                                      double* W_pt, double* W_ls, double* err)
{
    PLSLAM_REQUIRE(P != nullptr, PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);
    return lba_plan_download(P, g, H_pose, H_pt, H_ls, W_pt, W_ls, err);
}

// the rows of the last iterate() (device -> host), e.g. for the outlier logic of :1831-1846
extern "C" int plslam_lba_plan_rows(plslam_lba_plan* P, double* pt_J_pose, double* pt_J_lm, double* pt_r, double* pt_w,
                                    double* ls_J_pose, double* ls_J_lm, double* ls_r, double* ls_w)
{
    PLSLAM_REQUIRE(P != nullptr, PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);    // every entry point runs on the context's device, whatever the calling thread's current one
    hipStream_t s = ctx->stream;
    char* dr = P->rows.as<char>();
    const size_t np = (size_t)P->np, nl = (size_t)P->nl;
    auto down = [&](void* dst, size_t off, size_t bytes) -> int {
        if (dst && bytes) PLSLAM_HIP_CHECK(hipMemcpyAsync(dst, dr + off, bytes, hipMemcpyDeviceToHost, s));
        return PLSLAM_OK;
    };
    int rc;
    if ((rc = down(pt_J_pose, P->oPJp, np * 48)) || (rc = down(pt_J_lm, P->oPJl, np * 24)) || (rc = down(pt_r, P->oPr, np * 8)) ||
        (rc = down(pt_w, P->oPw, np * 8)) || (rc = down(ls_J_pose, P->oLJp, nl * 48)) || (rc = down(ls_J_lm, P->oLJl, nl * 48)) ||
        (rc = down(ls_r, P->oLr, nl * 8)) || (rc = down(ls_w, P->oLw, nl * 8)))
        return rc;
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return PLSLAM_OK;
}

extern "C" void plslam_lba_plan_destroy(plslam_lba_plan* P)
{
    if (!P) return;
    (void)hipStreamSynchronize(P->ctx->stream);
    P->stat.release(); P->dyn.release(); P->rows.release(); P->out.release(); P->pin_in.release(); P->pin_out.release();
    P->schur.release(); P->schur_pin.release();
    delete P;
}

extern "C" int plslam_lba_assemble(plslam_ctx* ctx, int32_t nkf, int32_t npt, int32_t nls,
                                   const int32_t* pt_lm_loc, const int32_t* pt_kf_loc, int32_t n_pt_obs,
                                   const double* pt_J_pose, const double* pt_J_lm, const double* pt_r,
                                   const double* pt_w, const int32_t* ls_lm_loc, const int32_t* ls_kf_loc,
                                   int32_t n_ls_obs, const double* ls_J_pose, const double* ls_J_lm,
                                   const double* ls_r, const double* ls_w, double* g, double* H_pose,
                                   double* H_pt, double* H_ls, double* W_pt, double* W_ls, double* err)
{
    PLSLAM_REQUIRE(ctx && nkf >= 0 && npt >= 0 && nls >= 0 && n_pt_obs >= 0 && n_ls_obs >= 0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(n_pt_obs == 0 || (pt_lm_loc && pt_kf_loc && pt_J_pose && pt_J_lm && pt_r && pt_w && W_pt), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(n_ls_obs == 0 || (ls_lm_loc && ls_kf_loc && ls_J_pose && ls_J_lm && ls_r && ls_w && W_ls), PLSLAM_EINVAL);
    PLSLAM_REQUIRE(g && err && (nkf == 0 || H_pose) && (npt == 0 || H_pt) && (nls == 0 || H_ls), PLSLAM_EINVAL);
    for (int32_t o = 0; o < n_pt_obs; ++o)
        PLSLAM_REQUIRE(pt_lm_loc[o] >= 0 && pt_lm_loc[o] < npt && pt_kf_loc[o] >= -1 && pt_kf_loc[o] < nkf, PLSLAM_EINVAL);
    for (int32_t o = 0; o < n_ls_obs; ++o)
        PLSLAM_REQUIRE(ls_lm_loc[o] >= 0 && ls_lm_loc[o] < nls && ls_kf_loc[o] >= -1 && ls_kf_loc[o] < nkf, PLSLAM_EINVAL);

    CsrLists L;
    build_csr(pt_lm_loc, pt_kf_loc, n_pt_obs, ls_lm_loc, ls_kf_loc, n_ls_obs, nkf, npt, nls, L);
    std::vector<int32_t>&ptp = L.ptp, &pti = L.pti, &lsp = L.lsp, &lsi = L.lsi, &kfp = L.kfp, &kfi = L.kfi;

    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);    // every entry point runs on the context's device, whatever the calling thread's current one
    hipStream_t s = ctx->stream;
    Carve c;
    const size_t np = (size_t)n_pt_obs, nl = (size_t)n_ls_obs;
    const size_t oPJp = c.take(np * 48), oPJl = c.take(np * 24), oPr = c.take(np * 8), oPw = c.take(np * 8),
                 oPk = c.take(np * 4), oLJp = c.take(nl * 48), oLJl = c.take(nl * 48), oLr = c.take(nl * 8),
                 oLw = c.take(nl * 8), oLk = c.take(nl * 4), oPtp = c.take(ptp.size() * 4), oPti = c.take(pti.size() * 4),
                 oLsp = c.take(lsp.size() * 4), oLsi = c.take(lsi.size() * 4), oKfp = c.take(kfp.size() * 4),
                 oKfi = c.take(kfi.size() * 4 + 4);
    const size_t N = 6 * (size_t)nkf + 3 * (size_t)npt + 6 * (size_t)nls;
    Carve co;
    const size_t oG = co.take(N * 8 + 8), oHp = co.take((size_t)nkf * 288 + 8), oHpt = co.take((size_t)npt * 72 + 8),
                 oHls = co.take((size_t)nls * 288 + 8), oWp = co.take(np * 144 + 8), oWl = co.take(nl * 288 + 8),
                 oErr = co.take(8 * (1 + ERR_BLOCKS));
    const int32_t max_chunks = pose_max_chunks(kfp);
    const size_t oPart = co.take((size_t)nkf * (size_t)max_chunks * 42 * 8 + 8);
    int rc;
    if ((rc = ctx->in_a.reserve(c.off + 256))) return rc;
    if ((rc = ctx->out_a.reserve(co.off + 256))) return rc;
    char* di = ctx->in_a.as<char>();
    char* dout = ctx->out_a.as<char>();
    auto up = [&](size_t off, const void* src, size_t bytes) -> int {
        if (bytes) PLSLAM_HIP_CHECK(hipMemcpyAsync(di + off, src, bytes, hipMemcpyHostToDevice, s));
        return PLSLAM_OK;
    };
    if ((rc = up(oPJp, pt_J_pose, np * 48)) || (rc = up(oPJl, pt_J_lm, np * 24)) || (rc = up(oPr, pt_r, np * 8)) ||
        (rc = up(oPw, pt_w, np * 8)) || (rc = up(oPk, pt_kf_loc, np * 4)) || (rc = up(oLJp, ls_J_pose, nl * 48)) ||
        (rc = up(oLJl, ls_J_lm, nl * 48)) || (rc = up(oLr, ls_r, nl * 8)) || (rc = up(oLw, ls_w, nl * 8)) ||
        (rc = up(oLk, ls_kf_loc, nl * 4)) || (rc = up(oPtp, ptp.data(), ptp.size() * 4)) ||
        (rc = up(oPti, pti.data(), pti.size() * 4)) || (rc = up(oLsp, lsp.data(), lsp.size() * 4)) ||
        (rc = up(oLsi, lsi.data(), lsi.size() * 4)) || (rc = up(oKfp, kfp.data(), kfp.size() * 4)) ||
        (rc = up(oKfi, kfi.data(), kfi.size() * 4)))
        return rc;
    AssembleDev a{(int32_t*)(di + oPk), (int32_t*)(di + oLk), (int32_t*)(di + oPtp), (int32_t*)(di + oPti),
                  (int32_t*)(di + oLsp), (int32_t*)(di + oLsi), (int32_t*)(di + oKfp), (int32_t*)(di + oKfi),
                  (double*)(di + oPJp), (double*)(di + oPJl), (double*)(di + oPr), (double*)(di + oPw),
                  (double*)(di + oLJp), (double*)(di + oLJl), (double*)(di + oLr), (double*)(di + oLw),
                  (double*)(dout + oG), (double*)(dout + oHp), (double*)(dout + oHpt), (double*)(dout + oHls),
                  (double*)(dout + oWp), (double*)(dout + oWl), (double*)(dout + oErr), (double*)(dout + oPart),
                  max_chunks};
    if ((rc = assemble_on_device(a, nkf, npt, nls, n_pt_obs, n_ls_obs, s))) return rc;
    auto down = [&](void* dst, size_t off, size_t bytes) -> int {
        if (bytes) PLSLAM_HIP_CHECK(hipMemcpyAsync(dst, dout + off, bytes, hipMemcpyDeviceToHost, s));
        return PLSLAM_OK;
    };
    if ((rc = down(g, oG, N * 8)) || (rc = down(H_pose, oHp, (size_t)nkf * 288)) || (rc = down(H_pt, oHpt, (size_t)npt * 72)) ||
        (rc = down(H_ls, oHls, (size_t)nls * 288)) || (rc = down(W_pt, oWp, np * 144)) || (rc = down(W_ls, oWl, nl * 288)) ||
        (rc = down(err, oErr, 8)))
        return rc;
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    return PLSLAM_OK;
}


// =============================================================================================================================
// The Schur step on the resident blocks (round 5) -- the consumer of plslam_lba_plan_iterate_dev / _resident.
//
// The reference solves the damped normal equations whole: H(i,i) += lambda * H(i,i), H.sparseView(), SimplicialLDLT
// (src/mapHandler.cpp:1552-1556; :1779-1783 in the loop) -- N = 6 Nkf + 3 Npt + 6 Nls unknowns, 42 054 at C3.  H has the block
// structure this file already produces, and the landmark blocks are independent of each other, so the same system is
//     S dp = b,   S = Hpp' - sum_j Wj^T Vj'^-1 Wj,   b = gp - sum_j Wj^T Vj'^-1 gj,   dxj = Vj'^-1 (gj - Wj dp)
// with Hpp' / Vj' the damped pose / landmark blocks and Wj the cross blocks of landmark j's observations: 6 Nkf unknowns (60 at
// C3) for the host's dense LDLT, everything else stays on the device.  What crosses PCIe per iteration is S and b down (29 kB at
// C3) and dp up (480 B) instead of 11.5 MB of blocks.  Algebraically the reference's solve; numerically equal to rounding
// (tests/test_gpu_lba.py compares with a dense solve of the assembled system).
//
// No atomics, nothing scheduling-dependent (as everywhere in this file): the pairs (o1, o2) of observations of one landmark that
// fall into block (k1 <= k2) of S are listed ONCE per plan on the host, sorted by block; a block's sum is a fixed-shape two-level
// sum -- chunks of SCH_CHUNK pairs sequentially in list order, then the chunk partials in chunk order -- and the lower triangle
// is the mirror of the upper one.
//   K19 k_schur_landmarks<DL>   a lane per landmark: Vj' inverse (closed form 3x3 / Gauss-Jordan 6x6), tj = Vj'^-1 gj
//   K20 k_schur_partials        a lane per (block, chunk, entry of the 6x6 block): sum over the chunk's pairs of W1^T Vinv W2
//   K21 (in K20's launch)       a lane per (keyframe, chunk, entry of b): sum over the chunk's observations of W^T tj
//   K22 k_schur_finish          a lane per (block, entry) / (keyframe, entry): chunk partials -> S (both triangles), b
//   K23 k_schur_backsub         a lane per (landmark, row): dxj = tj - Vinv_j sum_o W_o dp[kf(o)]  (+ the update of Xw / Lw in place)
//   K24 k_lba_diag_max          max |H(i,i)| over all diagonal entries (the reference's lambda *= Hmax, :1544-1550)
// =============================================================================================================================
namespace plslam {

constexpr int SCH_CHUNK = 64;
#ifndef PLSLAM_SCHUR_X
#define PLSLAM_SCHUR_X 0            // timing experiments only (tools/r6_schur_knockouts.sh): parts of the partials' launch left out
#endif
struct SchurPair { int32_t o1, o2, lm, line; };     // observation ids within their own list (points / lines), the landmark

template <int DL>
__device__ __forceinline__ void schur_landmark(const double* __restrict__ H, const double* __restrict__ g, int j, double lambda,
                                               double* __restrict__ Vinv, double* __restrict__ t, int32_t* __restrict__ nsing)
{
    double A[DL][DL], I[DL][DL];
#pragma unroll
    for (int a = 0; a < DL; ++a)
#pragma unroll
        for (int b = 0; b < DL; ++b) {
            const double h = H[(size_t)j * DL * DL + a * DL + b];
            A[a][b] = a == b ? h + lambda * h : h;
            I[a][b] = a == b ? 1.0 : 0.0;
        }
    // Gauss-Jordan without pivoting (the damped block is symmetric positive definite whenever its diagonal is positive);
    // a pivot that is not positive marks the landmark as singular: no step, no contribution (Vinv = 0, t = 0)
    bool ok = true;
#pragma unroll
    for (int c = 0; c < DL; ++c) {
        const double piv = A[c][c];
        ok = ok && piv > 0.0;
        const double ip = 1.0 / (piv > 0.0 ? piv : 1.0);
#pragma unroll
        for (int b = 0; b < DL; ++b) { A[c][b] *= ip; I[c][b] *= ip; }
#pragma unroll
        for (int a = 0; a < DL; ++a) {
            if (a == c) continue;
            const double f = A[a][c];
#pragma unroll
            for (int b = 0; b < DL; ++b) { A[a][b] -= f * A[c][b]; I[a][b] -= f * I[c][b]; }
        }
    }
    double gj[DL];
#pragma unroll
    for (int a = 0; a < DL; ++a) gj[a] = g[(size_t)j * DL + a];
#pragma unroll
    for (int a = 0; a < DL; ++a) {
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < DL; ++b) {
            const double v = ok ? I[a][b] : 0.0;
            Vinv[(size_t)j * DL * DL + a * DL + b] = v;
            acc += v * gj[b];
        }
        t[(size_t)j * DL + a] = acc;
    }
    if (!ok && nsing) atomicAdd(nsing, 1);     // (a count only: no sum depends on it)
}

// one launch for both kinds: workgroups [0, nb3) take the points, the rest the lines
__global__ void __launch_bounds__(256)
k_schur_landmarks(const double* __restrict__ H_pt, const double* __restrict__ g_pt, int32_t npt, const double* __restrict__ H_ls,
                  const double* __restrict__ g_ls, int32_t nls, int32_t nb3, double lambda, double* __restrict__ Vp,
                  double* __restrict__ tp, double* __restrict__ Vl, double* __restrict__ tl, int32_t* __restrict__ nsing)
{
    if ((int)blockIdx.x < nb3) {
        const int j = blockIdx.x * 256 + threadIdx.x;
        if (j < npt) schur_landmark<3>(H_pt, g_pt, j, lambda, Vp, tp, nsing);
    } else {
        const int j = ((int)blockIdx.x - nb3) * 256 + threadIdx.x;
        if (j < nls) schur_landmark<6>(H_ls, g_ls, j, lambda, Vl, tl, nsing);
    }
}

// W_pt[o]: 3 x 6 (landmark row x, pose column a), W_ls[o]: 6 x 6.  Entry (a, b) of W1^T Vinv W2 = sum_x sum_y W1[x][a] Vinv[x][y] W2[y][b].
// A workgroup = one chunk of SCH_CHUNK pairs: lane i computes the whole 6 x 6 contribution of pair i (its loads are one round
// trip: a first form with a lane per entry walking the chunk was a chain of 64 dependent round trips, 177 us per call at C3),
// parks it in LDS, and lane e < 36 adds the 64 values of entry e SEQUENTIALLY in pair order -- the same sum, term for term.
template <int DL>
__device__ __forceinline__ void schur_pair_product(const double (&w1)[DL * 6], const double (&w2)[DL * 6], const double (&v)[DL * DL],
                                                   double* __restrict__ out /* [36], stride 1 */)
{
#pragma unroll
    for (int b = 0; b < 6; ++b) {
        double u[DL];
#pragma unroll
        for (int x = 0; x < DL; ++x) {
            double t = 0.0;
#pragma unroll
            for (int y = 0; y < DL; ++y) t += v[x * DL + y] * w2[y * 6 + b];
            u[x] = t;
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double sacc = 0.0;
#pragma unroll
            for (int x = 0; x < DL; ++x) sacc += w1[x * 6 + a] * u[x];
            out[a * 6 + b] = sacc;
        }
    }
}
template <int DL>
__device__ __forceinline__ void schur_pair_block(const double* __restrict__ W1, const double* __restrict__ W2,
                                                 const double* __restrict__ V, double* __restrict__ out /* [36], stride 1 */)
{
    double w1[DL * 6], w2[DL * 6], v[DL * DL];
#pragma unroll
    for (int i = 0; i < DL * 6; ++i) { w1[i] = W1[i]; w2[i] = W2[i]; }
#pragma unroll
    for (int i = 0; i < DL * DL; ++i) v[i] = V[i];
    schur_pair_product<DL>(w1, w2, v, out);
}

// 64 rows of NE doubles, row r at base + NE * id(r) where lane r holds id(r): fetched by the WAVE in runs (consecutive lanes read
// consecutive words of a row: 4-8 cache lines per instruction; a lane reading its own row word by word touches 64), parked in
// the tile, and every lane takes its own row out of it.  One wave per workgroup: the barriers are the wave's own.
// One chunk per workgroup of one wave (PLSLAM_SCH_WAVES = 1).  Measured at C3 (tools/r6_schur_knockouts.sh): two or four chunks
// per workgroup, a wave each with its own tile and wave-level synchronisation, are SLOWER (24.0 / 24.9 us against 19.8: a
// workgroup's LDS and registers stay taken until its slowest wave is through), and so is the wave-level form at one wave (22.3: the
// scheduler takes the weaker barrier as leave to spread the gathers' loads over 285 registers).
#ifndef PLSLAM_SCH_OFF64
#define PLSLAM_SCH_OFF64 0
#endif
#ifndef PLSLAM_SCH_STATIC_LDS
#define PLSLAM_SCH_STATIC_LDS (PLSLAM_SCH_WAVES == 1)   // (a tile of known size: 20.8 us against 26.8 with the same tile as dynamic LDS --
#endif                                                  //  the register allocator sees what the workgroup's LDS allows and keeps two waves)
#ifndef PLSLAM_SCH_FINISH_LAST
#define PLSLAM_SCH_FINISH_LAST 0
#endif
__device__ __forceinline__ void wave_sync()
{
#if PLSLAM_SCH_WAVES == 1
    __syncthreads();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}
// (fetch: the loads, into registers -- several gathers' loads can be in flight together; park: through the tile, one at a time)
template <int NE>
__device__ __forceinline__ void gather_fetch(const double* __restrict__ base, int id, double (&flat)[NE])
{
    const int lane = threadIdx.x & 63;
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    if constexpr (NE % 2 == 0) {
        constexpr int H = NE / 2;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int idx = j * 64 + lane, r = idx / H, c2 = idx - r * H;
            const int rid = __shfl(id, r);
            // (a 32-bit byte offset from the uniform base: the load takes it beside the base in scalar registers -- a 64-bit
            // address per load was two more registers for each of the 18 loads in flight; lba_schur_prepare bounds the tables)
#if PLSLAM_SCH_OFF64
            const f64x2 v = *reinterpret_cast<const f64x2*>(base + (size_t)rid * NE + 2 * c2);
#else
            const uint32_t boff = ((uint32_t)rid * NE + 2 * c2) * 8u;
            const f64x2 v = *reinterpret_cast<const f64x2*>(reinterpret_cast<const char*>(base) + boff);
#endif
            flat[2 * j] = v.x;
            flat[2 * j + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int idx = j * 64 + lane, r = idx / NE, c = idx - r * NE;
            const int rid = __shfl(id, r);
#if PLSLAM_SCH_OFF64
            flat[j] = base[(size_t)rid * NE + c];
#else
            const uint32_t boff = ((uint32_t)rid * NE + c) * 8u;
            flat[j] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + boff);
#endif
        }
    }
}
// (park_only: the rows stay in the tile -- every lane reads its own row from there as it goes; the caller synchronises before the
// tile's next use)
template <int NE>
__device__ __forceinline__ void gather_park_only(double (*tile)[37], const double (&flat)[NE])
{
    static_assert(NE <= 36 && NE % 2 == 0, "the tile is 37 doubles wide");
    const int lane = threadIdx.x & 63;
    constexpr int H = NE / 2;
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const int idx = j * 64 + lane, r = idx / H, c2 = idx - r * H;
        tile[r][2 * c2] = flat[2 * j];
        tile[r][2 * c2 + 1] = flat[2 * j + 1];
    }
    wave_sync();
}
template <int NE>
__device__ __forceinline__ void gather_park(double (*tile)[37], const double (&flat)[NE], double (&row_out)[NE])
{
    static_assert(NE <= 36, "the tile is 37 doubles wide");
    const int lane = threadIdx.x & 63;
    if constexpr (NE % 2 == 0) {
        constexpr int H = NE / 2;
#pragma unroll
        for (int j = 0; j < H; ++j) {
            const int idx = j * 64 + lane, r = idx / H, c2 = idx - r * H;
            tile[r][2 * c2] = flat[2 * j];
            tile[r][2 * c2 + 1] = flat[2 * j + 1];
        }
    } else {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int idx = j * 64 + lane, r = idx / NE, c = idx - r * NE;
            tile[r][c] = flat[j];
        }
    }
    wave_sync();
#pragma unroll
    for (int e = 0; e < NE; ++e) row_out[e] = tile[lane][e];
    wave_sync();
}
template <int NE>
__device__ __forceinline__ void gather_rows(double (*tile)[37], const double* __restrict__ base, int id, double (&row_out)[NE])
{
    double flat[NE];
    gather_fetch<NE>(base, id, flat);
    gather_park<NE>(tile, flat, row_out);
}

__device__ __forceinline__ void
schur_partials_wg(double (*tile)[37], int B, int c, const SchurPair* __restrict__ pairs, const int32_t* __restrict__ blk_ptr,
                  const double* __restrict__ W_pt, const double* __restrict__ W_ls, const double* __restrict__ Vp,
                  const double* __restrict__ Vl, int32_t max_chunks, double* __restrict__ part /* [nblk][max_chunks][36] */)
{
    const int i = threadIdx.x & 63;
    const int beg = blk_ptr[B] + c * SCH_CHUNK;
    const int end = beg + SCH_CHUNK < blk_ptr[B + 1] ? beg + SCH_CHUNK : blk_ptr[B + 1];
    const int n = end - beg;                       // <= 0: the block has fewer chunks than the grid is wide
    if (n <= 0) return;
    SchurPair q = pairs[beg + (i < n ? i : n - 1)];           // (lanes past the chunk's end replay its last pair: their block is dropped)
#if PLSLAM_SCHUR_X & 1
    if (q.o1 != -12345) return;
#endif
    // the chunk's kind (lba_schur_prepare: one kind per chunk, null pairs -- line = 2 -- behind a block's last point pairs); a null
    // pair fetches what the chunk's first pair fetches and contributes zeros
    const bool null_pair = q.line == 2;
    if (null_pair) { q.o1 = __shfl(q.o1, 0); q.o2 = __shfl(q.o2, 0); q.lm = __shfl(q.lm, 0); }
    const bool line_chunk = __shfl(q.line, 0) == 1;
    // The wave fetches its 64 pairs' rows together (gather_*), each lane takes its own out of the tile, and the pair's 6 x 6 block
    // goes straight back into the tile (the sums are schur_pair_product's, term for term).  Registers: what a lane holds at once
    // decides how many waves a SIMD keeps (a first form with all operands and the block in registers: 256 and one wave; this one: two).
#if PLSLAM_SCHUR_X & 16
    if (line_chunk) return;
#endif
    if (!line_chunk) {
        double w1[18], w2[18], v[9];
        {
            double f1[18], f2[18], fv[9];
            gather_fetch<18>(W_pt, q.o1, f1);      // (the three gathers' loads: one round trip)
            gather_fetch<18>(W_pt, q.o2, f2);
            gather_fetch<9>(Vp, q.lm, fv);
            gather_park<18>(tile, f1, w1);
            gather_park<18>(tile, f2, w2);
            gather_park<9>(tile, fv, v);
        }
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            double u[3];
#pragma unroll
            for (int x = 0; x < 3; ++x) {
                double t = 0.0;
#pragma unroll
                for (int y = 0; y < 3; ++y) t += v[x * 3 + y] * w2[y * 6 + b];
                u[x] = t;
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double sacc = 0.0;
#pragma unroll
                for (int x = 0; x < 3; ++x) sacc += w1[x * 6 + a] * u[x];
                tile[i][a * 6 + b] = null_pair ? 0.0 : sacc;
            }
        }
    } else {
        // W2, then Y = Vinv W2 (column b of a lane's W2 row is dead once Y's column b is known: Y overwrites W2 in place), then W1,
        // and entry (a, b) = sum_x W1[x][a] Y[x][b] lands where W1[b][a] was (column a of W1 is dead by then): the block sits
        // TRANSPOSED in the tile, and the lanes that add the chunk's blocks read it so
        {
            double v[36];
            {
                double fv[36], f2[36];
                gather_fetch<36>(Vl, q.lm, fv);
                gather_fetch<36>(W_ls, q.o2, f2);
                gather_park<36>(tile, fv, v);
                gather_park_only<36>(tile, f2);
            }
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                double w2c[6], yc[6];
#pragma unroll
                for (int y = 0; y < 6; ++y) w2c[y] = tile[i][y * 6 + b];
#pragma unroll
                for (int x = 0; x < 6; ++x) {
                    double t = 0.0;
#pragma unroll
                    for (int y = 0; y < 6; ++y) t += v[x * 6 + y] * w2c[y];
                    yc[x] = t;                     // u[x] of column b
                }
#pragma unroll
                for (int x = 0; x < 6; ++x) tile[i][x * 6 + b] = yc[x];
            }
        }
        double Y[36];
#pragma unroll
        for (int e = 0; e < 36; ++e) Y[e] = tile[i][e];
        {
            double f1[36];
            gather_fetch<36>(W_ls, q.o1, f1);
            wave_sync();                       // (every lane has taken its Y)
            gather_park_only<36>(tile, f1);
        }
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            double w1c[6], oc[6];
#pragma unroll
            for (int x = 0; x < 6; ++x) w1c[x] = tile[i][x * 6 + a];
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                double sacc = 0.0;
#pragma unroll
                for (int x = 0; x < 6; ++x) sacc += w1c[x] * Y[x * 6 + b];
                oc[b] = sacc;
            }
#pragma unroll
            for (int b = 0; b < 6; ++b) tile[i][b * 6 + a] = null_pair ? 0.0 : oc[b];
        }
    }
    wave_sync();
#if PLSLAM_SCHUR_X & 8
    if (n != -12345) return;
#endif
    if (i < 36) {                                  // (eight LDS reads in flight, added in pair order)
        const int slot = line_chunk ? (i % 6) * 6 + i / 6 : i;
        double acc = 0.0;
        for (int k0 = 0; k0 < n; k0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = tile[k0 + u < n ? k0 + u : n - 1][slot];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (k0 + u < n) acc += v[u];
        }
        part[((size_t)B * max_chunks + c) * 36 + i] = acc;
    }
}

// b's share of keyframe k: sum over its observations (the keyframe lists of the assembly: points first, then lines) of W_o^T t_lm
// -- a lane per observation of the chunk, then six lanes add the chunk's terms sequentially in list order
__device__ __forceinline__ void
schur_b_partials_wg(double (*tile)[37], int k, int c, const int32_t* __restrict__ kf_ptr, const int32_t* __restrict__ kf_obs,
                    int32_t n_pt_obs, const int32_t* __restrict__ pt_lm, const int32_t* __restrict__ ls_lm,
                    const double* __restrict__ W_pt, const double* __restrict__ W_ls, const double* __restrict__ tp,
                    const double* __restrict__ tl, int32_t max_chunks, double* __restrict__ part /* [nkf][max_chunks][6] */)
{
    const int i = threadIdx.x & 63;
    const int beg = kf_ptr[k] + c * POSE_CHUNK;
    const int end = beg + POSE_CHUNK < kf_ptr[k + 1] ? beg + POSE_CHUNK : kf_ptr[k + 1];
    const int n = end - beg;
    if (n <= 0) return;
    if (i < n) {
        const int o = kf_obs[beg + i];
        if (o < n_pt_obs) {
            const double* W = W_pt + (size_t)o * 18;
            const double* t = tp + (size_t)pt_lm[o] * 3;
            const double t0 = t[0], t1 = t[1], t2 = t[2];
#pragma unroll
            for (int a = 0; a < 6; ++a) tile[i][a] = W[0 * 6 + a] * t0 + W[1 * 6 + a] * t1 + W[2 * 6 + a] * t2;
        } else {
            const int ol = o - n_pt_obs;
            const double* W = W_ls + (size_t)ol * 36;
            const double* t = tl + (size_t)ls_lm[ol] * 6;
            double tt[6];
#pragma unroll
            for (int x = 0; x < 6; ++x) tt[x] = t[x];
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                double sacc = 0.0;
#pragma unroll
                for (int x = 0; x < 6; ++x) sacc += W[x * 6 + a] * tt[x];
                tile[i][a] = sacc;
            }
        }
    }
    wave_sync();
    if (i < 6) {
        double acc = 0.0;
        for (int q0 = 0; q0 < n; q0 += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = tile[q0 + u < n ? q0 + u : n - 1][i];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (q0 + u < n) acc += v[u];
        }
        part[((size_t)k * max_chunks + c) * 6 + i] = acc;
    }
}

// both kinds of chunk partials in ONE launch (they need the landmark inverses and nothing of each other): workgroups
// [0, nblk * schur_chunks) take the (block, chunk) pairs of S, the rest the (keyframe, chunk) pairs of b
struct SchurPartArgs {
    const SchurPair* pairs; const int32_t* blk_ptr; const int32_t* kf_ptr; const int32_t* kf_obs; const int32_t* pt_lm; const int32_t* ls_lm;
    const double *W_pt, *W_ls, *Vp, *Vl, *tp, *tl;
    double *spart, *bpart;
    int32_t nblk, schur_chunks, nkf, pose_chunks, n_pt_obs;
};
static_assert(SCH_CHUNK == POSE_CHUNK, "one workgroup shape for both kinds of chunk");
// (PLSLAM_SCH_WAVES chunks per workgroup, a wave and a tile each: see wave_sync; the tiles are dynamic LDS -- four of them are past
// the 64 kB a kernel may use without asking; PLSLAM_SCH_VGPRS: a register budget for the experiments, 0 = the compiler's own)
constexpr int SCH_WAVES = PLSLAM_SCH_WAVES;
#ifndef PLSLAM_SCH_VGPRS
#define PLSLAM_SCH_VGPRS 0
#endif
#if PLSLAM_SCH_VGPRS
#define SCH_VGPR_ATTR __attribute__((amdgpu_num_vgpr(PLSLAM_SCH_VGPRS / 2)))
#else
#define SCH_VGPR_ATTR
#endif
constexpr size_t SCH_TILE_DOUBLES = (size_t)SCH_CHUNK * 37;
constexpr size_t SCH_LDS_BYTES = SCH_WAVES * SCH_TILE_DOUBLES * 8;
__device__ __forceinline__ void schur_partials_item(double (*tile)[37], const SchurPartArgs& A, int w, int npart)
{
    const int nS = A.nblk * A.schur_chunks;
    if (w < nS)
        schur_partials_wg(tile, w / A.schur_chunks, w % A.schur_chunks, A.pairs, A.blk_ptr, A.W_pt, A.W_ls, A.Vp, A.Vl, A.schur_chunks, A.spart);
    else if (w < npart)
        schur_b_partials_wg(tile, (w - nS) / A.pose_chunks, (w - nS) % A.pose_chunks, A.kf_ptr, A.kf_obs, A.n_pt_obs, A.pt_lm, A.ls_lm,
                            A.W_pt, A.W_ls, A.tp, A.tl, A.pose_chunks, A.bpart);
}
__global__ void __launch_bounds__(SCH_CHUNK * SCH_WAVES) SCH_VGPR_ATTR
k_schur_partials(SchurPartArgs A, int32_t npart)
{
#if PLSLAM_SCH_STATIC_LDS
    __shared__ __attribute__((aligned(16))) double sch_lds[SCH_WAVES * SCH_TILE_DOUBLES];
#else
    extern __shared__ __attribute__((aligned(16))) double sch_lds[];
#endif
    const int wv = (int)threadIdx.x >> 6;
    schur_partials_item(reinterpret_cast<double (*)[37]>(sch_lds + wv * SCH_TILE_DOUBLES), A, (int)blockIdx.x * SCH_WAVES + wv, npart);
}

// plslam_lba_plan_iterate_schur: the Schur partials and, in the SAME launch, the iteration's last stage (K10: keyframe blocks and err
// from their partials) -- the two are independent (the partials read W and the landmark inverses, K22 behind them reads H_pose /
// g_pose), so the iteration's third launch rides in the Schur step's second.  The last stage's workgroups come FIRST (a wave each;
// chains of round trips: at the grid's end they would be the launch's tail).
__global__ void __launch_bounds__(SCH_CHUNK * SCH_WAVES) SCH_VGPR_ATTR
k_schur_partials_lba_finish(SchurPartArgs A, const LbaBlockArgs B, int32_t npart)
{
#if PLSLAM_SCH_STATIC_LDS
    __shared__ __attribute__((aligned(16))) double sch_lds[SCH_WAVES * SCH_TILE_DOUBLES];
#else
    extern __shared__ __attribute__((aligned(16))) double sch_lds[];
#endif
    const int wv = (int)threadIdx.x >> 6;
    const int nF = B.nkf + 1;
#if PLSLAM_SCH_FINISH_LAST
    const int ngrid = (npart + SCH_WAVES - 1) / SCH_WAVES;
    if ((int)blockIdx.x >= ngrid) {
        if (wv == 0) lba_finish_wg<SCH_CHUNK>(B, (int)blockIdx.x - ngrid, sch_lds);
        return;
    }
    schur_partials_item(reinterpret_cast<double (*)[37]>(sch_lds + wv * SCH_TILE_DOUBLES), A, (int)blockIdx.x * SCH_WAVES + wv, npart);
#else
    if ((int)blockIdx.x < nF) {
        if (wv == 0) lba_finish_wg<SCH_CHUNK>(B, (int)blockIdx.x, sch_lds);
        return;
    }
    schur_partials_item(reinterpret_cast<double (*)[37]>(sch_lds + wv * SCH_TILE_DOUBLES), A, ((int)blockIdx.x - nF) * SCH_WAVES + wv, npart);
#endif
}

// block B = (k1 <= k2) in row-major upper-triangle order; S is (6 nkf) x (6 nkf) row-major, b follows it
__global__ void __launch_bounds__(64)
k_schur_finish(const int32_t* __restrict__ blk_ptr, const int32_t* __restrict__ kf_ptr, const double* __restrict__ spart,
               const double* __restrict__ bpart, const double* __restrict__ H_pose, const double* __restrict__ g_pose,
               int32_t nkf, int32_t nblk, int32_t schur_chunks, int32_t pose_chunks, double lambda, double* __restrict__ S,
               double* __restrict__ bvec, int32_t* __restrict__ next_sing, const double* __restrict__ err_src, double* __restrict__ err_dst,
               const int32_t* __restrict__ sing_cur, int32_t* __restrict__ sing_out)
{
    const int B = blockIdx.x, e = threadIdx.x;
    const int n6 = 6 * nkf;
    if (B == 0 && e == 0) {
        *next_sing = 0;                            // the NEXT call's counter of singular landmarks (this call's: the other word)
        *err_dst = *err_src;                       // the iteration's err beside S and b: plslam_lba_plan_iterate_schur's ONE copy
        if (sing_out) *sing_out = *sing_cur;       // (S, b written in place in the host's image: this call's counter beside them)
    }
    if (B < nblk) {
        if (e >= 36) return;
        int k1 = 0, rem = B;                       // B = offset(k1) + (k2 - k1), offset(k1) = sum_{i < k1} (nkf - i)
        while (rem >= nkf - k1) { rem -= nkf - k1; ++k1; }
        const int k2 = k1 + rem, a = e / 6, b = e % 6;
        const int nch = (blk_ptr[B + 1] - blk_ptr[B] + SCH_CHUNK - 1) / SCH_CHUNK;
        // (the chunk partials are added in chunk order; their loads go out eight at a time -- one by one they are a chain of
        // round trips)
        double acc = 0.0;
        for (int c0 = 0; c0 < nch; c0 += 32) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = c0 + u < nch ? spart[((size_t)B * schur_chunks + c0 + u) * 36 + e] : 0.0;
#pragma unroll
            for (int u = 0; u < 32; ++u) if (c0 + u < nch) acc += v[u];
        }
        double h = 0.0;
        if (k1 == k2) {
            h = H_pose[(size_t)k1 * 36 + e];
            if (a == b) h += lambda * h;
        }
        const double v = h - acc;
        S[(size_t)(6 * k1 + a) * n6 + 6 * k2 + b] = v;
        if (k1 != k2) S[(size_t)(6 * k2 + b) * n6 + 6 * k1 + a] = v;
    } else {
        const int k = B - nblk;
        if (e >= 6) return;
        const int nch = (kf_ptr[k + 1] - kf_ptr[k] + POSE_CHUNK - 1) / POSE_CHUNK;
        double acc = 0.0;
        for (int c0 = 0; c0 < nch; c0 += 32) {
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = c0 + u < nch ? bpart[((size_t)k * pose_chunks + c0 + u) * 6 + e] : 0.0;
#pragma unroll
            for (int u = 0; u < 32; ++u) if (c0 + u < nch) acc += v[u];
        }
        bvec[6 * k + e] = g_pose[6 * k + e] - acc;
    }
}

// A lane per (landmark, row x of its step): the row's share of sum_o W_o dp[kf(o)] in list order, the DL shares of a landmark
// exchanged through LDS, then row x of Vinv times them.  192 lanes = 64 points or 32 lines per workgroup; points' workgroups
// first, lines' behind them: ONE launch (a lane per landmark was 8 workgroups for C3's 2 000 lines: 15 + 9 us in two launches).
constexpr int BACK_WG = 192;
template <int DL>
__device__ __forceinline__ void
schur_backsub_wg(double* __restrict__ sh, int wg, const int32_t* __restrict__ lm_ptr, const int32_t* __restrict__ lm_obs,
                 const int32_t* __restrict__ kf_loc, int32_t n, const double* __restrict__ W, const double* __restrict__ Vinv,
                 const double* __restrict__ t, const double* __restrict__ dp, double* __restrict__ dx, double* __restrict__ X,
                 double* __restrict__ part)
{
    constexpr int PER_WG = BACK_WG / DL;
    const int q = (int)threadIdx.x / DL, x = (int)threadIdx.x % DL;
    const int j = wg * PER_WG + q;
    double acc = 0.0;
    if (j < n) {
        for (int i = lm_ptr[j]; i < lm_ptr[j + 1]; ++i) {
            const int o = lm_obs[i], k = kf_loc[o];
            if (k < 0) continue;                       // a fixed keyframe: no cross block, no step
            const double* Wo = W + (size_t)o * DL * 6 + x * 6;
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) s += Wo[a] * dp[6 * k + a];
            acc += s;
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    double d = 0.0;
    if (j < n) {
        double s = 0.0;
#pragma unroll
        for (int y = 0; y < DL; ++y) s += Vinv[(size_t)j * DL * DL + x * DL + y] * sh[q * DL + y];
        d = t[(size_t)j * DL + x] - s;
        dx[(size_t)j * DL + x] = d;
        if (X) X[(size_t)j * DL + x] += d;             // :1570-1575 "update point / line LMs": X(i) += DX(i)
    }
    // the workgroup's share of sum DX^2 (the loop's ||DX|| test, :1808), summed in a fixed tree: the same bits on every run
    __syncthreads();
    sh[threadIdx.x] = d * d;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w && (int)threadIdx.x + w < BACK_WG) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *part = sh[0];
}

struct SchurBackArgs {
    const int32_t *pt_ptr, *pt_obs, *pt_kf, *ls_ptr, *ls_obs, *ls_kf;
    const double *W_pt, *W_ls, *Vp, *Vl, *tp, *tl, *dp;
    double *dx_pt, *dx_ls, *X, *L, *part;              // X / L = nullptr: do not apply; part: a sum of squares per workgroup
    int32_t npt, nls, nwg_pt;
};
__global__ void __launch_bounds__(BACK_WG)
k_schur_backsub(SchurBackArgs A)
{
    __shared__ double sh[BACK_WG];
    if ((int)blockIdx.x < A.nwg_pt)
        schur_backsub_wg<3>(sh, (int)blockIdx.x, A.pt_ptr, A.pt_obs, A.pt_kf, A.npt, A.W_pt, A.Vp, A.tp, A.dp, A.dx_pt, A.X, A.part + blockIdx.x);
    else
        schur_backsub_wg<6>(sh, (int)blockIdx.x - A.nwg_pt, A.ls_ptr, A.ls_obs, A.ls_kf, A.nls, A.W_ls, A.Vl, A.tl, A.dp, A.dx_ls, A.L, A.part + blockIdx.x);
}

// max over all diagonal entries of |H(i,i)| (a maximum: exact whatever the order)
__global__ void __launch_bounds__(256)
k_lba_diag_max(const double* __restrict__ H_pose, int32_t nkf, const double* __restrict__ H_pt, int32_t npt,
               const double* __restrict__ H_ls, int32_t nls, double* __restrict__ out)
{
    __shared__ double red[256];
    double m = 0.0;
    const int total = 6 * nkf + 3 * npt + 6 * nls;
    for (int i = threadIdx.x; i < total; i += 256) {
        double v;
        if (i < 6 * nkf) v = H_pose[(size_t)(i / 6) * 36 + (i % 6) * 7];
        else if (i < 6 * nkf + 3 * npt) { const int q = i - 6 * nkf; v = H_pt[(size_t)(q / 3) * 9 + (q % 3) * 4]; }
        else { const int q = i - 6 * nkf - 3 * npt; v = H_ls[(size_t)(q / 6) * 36 + (q % 6) * 7]; }
        v = fabs(v);
        m = v > m ? v : m;
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = red[threadIdx.x] > red[threadIdx.x + s] ? red[threadIdx.x] : red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

}  // namespace plslam

// the pair lists of the plan (host, once): every ordered pair (o1, o2) of observations of ONE landmark by OPTIMISED keyframes
// with kf(o1) <= kf(o2), sorted by block (k1, k2) -- stable: landmark order, then list order
static int lba_schur_prepare(plslam_lba_plan* P)
{
    if (P->schur_ready) return PLSLAM_OK;
    // (the partials address the cross blocks and the landmark inverses by 32-bit byte offsets: 288 bytes per line row)
    PLSLAM_REQUIRE((size_t)P->np * 144 < (size_t(1) << 32) && (size_t)P->nl * 288 < (size_t(1) << 32) &&
                   (size_t)P->npt * 72 < (size_t(1) << 32) && (size_t)P->nls * 288 < (size_t(1) << 32), PLSLAM_EINVAL);
    const int32_t nkf = P->nkf;
    P->nblk = nkf * (nkf + 1) / 2;
    auto blk_of = [nkf](int32_t k1, int32_t k2) { return k1 * nkf - k1 * (k1 - 1) / 2 + (k2 - k1); };
    std::vector<int32_t> cnt((size_t)P->nblk + 1, 0);
    auto each_pair = [&](auto&& fn) {
        for (int line = 0; line < 2; ++line) {
            const std::vector<int32_t>& ptr = line ? P->csr.lsp : P->csr.ptp;
            const std::vector<int32_t>& ids = line ? P->csr.lsi : P->csr.pti;
            const std::vector<int32_t>& kf = line ? P->h_ls_kf : P->h_pt_kf;
            const int32_t nlm = line ? P->nls : P->npt;
            for (int32_t j = 0; j < nlm; ++j)
                for (int32_t i1 = ptr[j]; i1 < ptr[j + 1]; ++i1) {
                    const int32_t o1 = ids[i1], k1 = kf[o1];
                    if (k1 < 0) continue;
                    for (int32_t i2 = ptr[j]; i2 < ptr[j + 1]; ++i2) {
                        const int32_t o2 = ids[i2], k2 = kf[o2];
                        if (k2 < k1) continue;             // (k2 < 0 included)
                        fn(blk_of(k1, k2), SchurPair{o1, o2, j, line});
                    }
                }
        }
    };
    // a block's point pairs, then its line pairs starting at a CHUNK boundary (null pairs -- line = 2: no contribution -- fill the last
    // point chunk of a block that has both kinds): every chunk is of one kind, and the wave fetches its rows together
    std::vector<int32_t> npt_pairs((size_t)P->nblk, 0), nls_pairs((size_t)P->nblk, 0);
    each_pair([&](int32_t B, const SchurPair& q) { ++(q.line ? nls_pairs : npt_pairs)[(size_t)B]; });
    auto pt_room = [&](int32_t B) {
        const int32_t np_ = npt_pairs[(size_t)B];
        return nls_pairs[(size_t)B] ? (np_ + SCH_CHUNK - 1) / SCH_CHUNK * SCH_CHUNK : np_;
    };
    for (int32_t B = 0; B < P->nblk; ++B) cnt[(size_t)B + 1] = cnt[B] + pt_room(B) + nls_pairs[(size_t)B];
    std::vector<SchurPair> pairs((size_t)cnt[P->nblk], SchurPair{0, 0, 0, 2});
    std::vector<int32_t> pos_pt(cnt.begin(), cnt.end() - 1), pos_ls((size_t)P->nblk);
    for (int32_t B = 0; B < P->nblk; ++B) pos_ls[(size_t)B] = cnt[B] + pt_room(B);
    each_pair([&](int32_t B, const SchurPair& q) { pairs[(size_t)(q.line ? pos_ls : pos_pt)[(size_t)B]++] = q; });
    int32_t mc = 0;
    for (int32_t B = 0; B < P->nblk; ++B) mc = std::max(mc, (cnt[(size_t)B + 1] - cnt[B] + SCH_CHUNK - 1) / SCH_CHUNK);
    P->schur_chunks = mc;
    const size_t n6 = 6 * (size_t)nkf;
    Carve c;
    P->oSpair = c.take(pairs.size() * sizeof(SchurPair) + 16); P->oSblk = c.take(cnt.size() * 4);
    P->oVp = c.take((size_t)P->npt * 72 + 8); P->oVl = c.take((size_t)P->nls * 288 + 8);
    P->oTp = c.take((size_t)P->npt * 24 + 8); P->oTl = c.take((size_t)P->nls * 48 + 8);
    P->oSpart = c.take((size_t)P->nblk * (size_t)mc * 36 * 8 + 8);
    P->oBpart = c.take((size_t)nkf * (size_t)P->max_chunks * 6 * 8 + 8);
    P->oS = c.take((n6 * n6 + n6 + 3) * 8 + 16);       // S, then b, then the diagonal maximum, the singular-block counters, err
    P->oDp = c.take(n6 * 8 + 8);
    P->oDx = c.take((3 * (size_t)P->npt + 6 * (size_t)P->nls) * 8 + 8);
    // (then a partial sum of squares per workgroup of the back-substitution, and their sum)
    P->oDxPart = c.take(((size_t)(P->npt + BACK_WG / 3 - 1) / (BACK_WG / 3) + (size_t)(P->nls + BACK_WG / 6 - 1) / (BACK_WG / 6) + 2) * 8);
    P->oSing = c.take(8);
    int rc;
    if ((rc = P->schur.reserve(c.off + 256)) || (rc = P->schur_pin.reserve(std::max((n6 * n6 + n6 + 3) * 8, (3 * (size_t)P->npt + 6 * (size_t)P->nls) * 8) + n6 * 8 +
                                                                              ((size_t)P->npt / (BACK_WG / 3) + (size_t)P->nls / (BACK_WG / 6) + 2) * 8 + 256)))
        return rc;
    P->schur_pin_dev = static_cast<char*>(P->schur_pin.dev);
#if !PLSLAM_SCH_STATIC_LDS
    // (the experiments' four tiles are past the 64 kB a kernel may use without asking)
    PLSLAM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_partials), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SCH_LDS_BYTES));
    PLSLAM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_schur_partials_lba_finish), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)SCH_LDS_BYTES));
#endif
    hipStream_t s = P->ctx->stream;
    char* d = P->schur.as<char>();
    if (!pairs.empty()) PLSLAM_HIP_CHECK(hipMemcpyAsync(d + P->oSpair, pairs.data(), pairs.size() * sizeof(SchurPair), hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemcpyAsync(d + P->oSblk, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice, s));
    PLSLAM_HIP_CHECK(hipMemsetAsync(d + P->oS + (n6 * n6 + n6 + 1) * 8, 0, 8, s));      // both counters of singular landmarks
    P->schur_parity = 0;
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));          // (the staging vectors die here)
    P->schur_ready = true;
    return PLSLAM_OK;
}

extern "C" int plslam_lba_plan_diag_max(plslam_lba_plan* P, double* hmax)
{
    PLSLAM_REQUIRE(P && hmax, PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);               // (the plan's state flags are read under the lock: ADVICE r5)
    DeviceGuard dg_(ctx->device);
    PLSLAM_REQUIRE(P->blocks_valid, PLSLAM_EINVAL);        // one plslam_lba_plan_iterate* first
    int rc = lba_schur_prepare(P);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    char *dout = P->out.as<char>(), *d = P->schur.as<char>();
    const size_t n6 = 6 * (size_t)P->nkf;
    double* dmax = (double*)(d + P->oS) + n6 * n6 + n6;
    hipLaunchKernelGGL(k_lba_diag_max, dim3(1), dim3(256), 0, s, (const double*)(dout + P->oHp), P->nkf, (const double*)(dout + P->oHpt),
                       P->npt, (const double*)(dout + P->oHls), P->nls, dmax);
    PLSLAM_HIP_CHECK(hipGetLastError());
    double* ho = P->schur_pin.as<double>();
    PLSLAM_HIP_CHECK(hipMemcpyAsync(ho, dmax, 8, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    *hmax = ho[0];
    return PLSLAM_OK;
}

// the Schur step's three launches on the blocks of the last iteration (no copy, no synchronisation); *par = the counter word of
// singular landmarks this call counts in
static int lba_schur_enqueue(plslam_lba_plan* P, double lambda, int* par_out, bool fused = false)
{
    int rc = lba_schur_prepare(P);
    if (rc) return rc;
    hipStream_t s = P->ctx->stream;
    char *ds = P->stat.as<char>(), *dout = P->out.as<char>(), *d = P->schur.as<char>();
    const size_t n6 = 6 * (size_t)P->nkf;
    const double* g = (const double*)(dout + P->oG);
    double *Vp = (double*)(d + P->oVp), *Vl = (double*)(d + P->oVl), *tp = (double*)(d + P->oTp), *tl = (double*)(d + P->oTl);
    double* dS = (double*)(d + P->oS);
    // two counters of singular landmarks behind S, b and the diagonal maximum (ONE copy brings S, b and them back): a call counts
    // in one of them and its last kernel clears the other for the next call (both cleared at plan creation) -- no memset launch
    int32_t* sing2 = (int32_t*)(dS + n6 * n6 + n6 + 1);
    const int par = P->schur_parity;
    const int32_t nb3 = (P->npt + 255) / 256, nb6 = (P->nls + 255) / 256;
    if (nb3 + nb6 > 0 && !fused)                          // (fused: the iteration's blocks launch has written them)
        hipLaunchKernelGGL(k_schur_landmarks, dim3(nb3 + nb6), dim3(256), 0, s, (const double*)(dout + P->oHpt), g + n6, P->npt,
                           (const double*)(dout + P->oHls), g + n6 + 3 * (size_t)P->npt, P->nls, nb3, lambda, Vp, tp, Vl, tl, sing2 + par);
    SchurPartArgs A{};
    A.pairs = (const SchurPair*)(d + P->oSpair); A.blk_ptr = (const int32_t*)(d + P->oSblk);
    A.kf_ptr = (const int32_t*)(ds + P->oKfp); A.kf_obs = (const int32_t*)(ds + P->oKfi);
    A.pt_lm = (const int32_t*)(ds + P->oPlm); A.ls_lm = (const int32_t*)(ds + P->oLlm);
    A.W_pt = (const double*)(dout + P->oWp); A.W_ls = (const double*)(dout + P->oWl); A.Vp = Vp; A.Vl = Vl; A.tp = tp; A.tl = tl;
    A.spart = (double*)(d + P->oSpart); A.bpart = (double*)(d + P->oBpart);
    A.nblk = P->nblk; A.schur_chunks = P->schur_chunks; A.nkf = P->nkf; A.pose_chunks = P->max_chunks; A.n_pt_obs = P->np;
    const int32_t npart_wgs = P->nblk * P->schur_chunks + P->nkf * P->max_chunks;
    const int32_t npart_grid = (npart_wgs + SCH_WAVES - 1) / SCH_WAVES;
    if (fused)
        hipLaunchKernelGGL(k_schur_partials_lba_finish, dim3(npart_grid + P->nkf + 1), dim3(SCH_CHUNK * SCH_WAVES), PLSLAM_SCH_STATIC_LDS ? 0 : SCH_LDS_BYTES, s, A,
                           lba_block_args(P), npart_wgs);
    else if (npart_grid > 0)
        hipLaunchKernelGGL(k_schur_partials, dim3(npart_grid), dim3(SCH_CHUNK * SCH_WAVES), PLSLAM_SCH_STATIC_LDS ? 0 : SCH_LDS_BYTES, s, A, npart_wgs);
    // S, b, err and this call's counter: written where the host reads them (the page-locked image, mapped) when it can be -- no
    // copy behind the kernel; otherwise beside the partials on the device, and lba_schur_fetch copies
    const bool in_place = P->schur_pin_dev != nullptr;
    double* oS = in_place ? (double*)P->schur_pin_dev : dS;
    hipLaunchKernelGGL(k_schur_finish, dim3(P->nblk + P->nkf), dim3(64), 0, s, (const int32_t*)(d + P->oSblk), (const int32_t*)(ds + P->oKfp),
                       (const double*)(d + P->oSpart), (const double*)(d + P->oBpart), (const double*)(dout + P->oHp), g, P->nkf, P->nblk,
                       P->schur_chunks, P->max_chunks, lambda, oS, oS + n6 * n6, sing2 + (par ^ 1), (const double*)(dout + P->oErr),
                       oS + n6 * n6 + n6 + 2, sing2 + par, in_place ? (int32_t*)(oS + n6 * n6 + n6 + 1) + par : nullptr);
    PLSLAM_HIP_CHECK(hipGetLastError());
    P->schur_parity = par ^ 1;        // (only now: the kernel that clears the other counter is in the stream)
    *par_out = par;
    return PLSLAM_OK;
}

// S, b and the counter behind them: one copy, one synchronisation
static int lba_schur_fetch(plslam_lba_plan* P, int par, double* S, double* b, int32_t* n_singular, double* err = nullptr)
{
    hipStream_t s = P->ctx->stream;
    const size_t n6 = 6 * (size_t)P->nkf;
    double* dS = (double*)(P->schur.as<char>() + P->oS);
    char* ho = P->schur_pin.as<char>();
    if (!P->schur_pin_dev) PLSLAM_HIP_CHECK(hipMemcpyAsync(ho, dS, (n6 * n6 + n6 + 3) * 8, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    memcpy(S, ho, n6 * n6 * 8);
    memcpy(b, ho + n6 * n6 * 8, n6 * 8);
    if (n_singular) memcpy(n_singular, ho + (n6 * n6 + n6 + 1) * 8 + 4 * par, 4);
    if (err) memcpy(err, ho + (n6 * n6 + n6 + 2) * 8, 8);
    P->schur_done = true;
    return PLSLAM_OK;
}

extern "C" int plslam_lba_plan_schur(plslam_lba_plan* P, double lambda, double* S, double* b, int32_t* n_singular)
{
    PLSLAM_REQUIRE(P && S && b && lambda >= 0.0, PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);
    PLSLAM_REQUIRE(P->blocks_valid && P->nkf > 0, PLSLAM_EINVAL);
    // the cross blocks must be the local BA's (landmark rows x pose columns): an iteration run with PLSLAM_LBA_COMPAT_GBA wrote the
    // pose x line blocks transposed, as the reference's GBA does (:2341-2352) -- a defect this step does not reproduce
    PLSLAM_REQUIRE(!P->blocks_gba, PLSLAM_EINVAL);
    int par = 0;
    int rc = lba_schur_enqueue(P, lambda, &par);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    return lba_schur_fetch(P, par, S, b, n_singular);
}

// One LM iteration's device half in ONE call and ONE synchronisation (round 6): H, g, err on the RESIDENT state
// (plslam_lba_plan_iterate_resident) and, straight behind it, the Schur step for `lambda` (plslam_lba_plan_schur) -- the
// reference's :1587-1783 for a lambda the caller already knows (every iteration but the first, whose lambda needs Hmax).
extern "C" int plslam_lba_plan_iterate_schur(plslam_lba_plan* P, int compat_flags, double lambda, double* err, double* S, double* b,
                                             int32_t* n_singular)
{
    PLSLAM_REQUIRE(P && err && S && b && lambda >= 0.0, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(!(compat_flags & PLSLAM_LBA_COMPAT_GBA), PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);
    PLSLAM_REQUIRE(P->state_valid && P->nkf > 0, PLSLAM_EINVAL);
    hipStream_t s = ctx->stream;
    // four launches for the iteration and its Schur step: rows + cross blocks | landmark blocks AND their damped inverses, keyframe
    // chunk partials | Schur partials AND the keyframe blocks + err | S, b (+ err beside them); the separate calls make seven
    int rc = lba_schur_prepare(P);
    if (!rc) rc = lba_plan_enqueue(P, nullptr, nullptr, nullptr, compat_flags, false, lambda);
    int par = 0;
    if (!rc) rc = lba_schur_enqueue(P, lambda, &par, true);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    return lba_schur_fetch(P, par, S, b, n_singular, err);    // (err rides behind S and b: k_schur_finish put it there)
}

// dp up, the back-substitution (and the landmark update when `apply`) into the stream; no synchronisation
static int lba_backsub_enqueue(plslam_lba_plan* P, const double* dpose, int apply, double* part_out = nullptr)
{
    hipStream_t s = P->ctx->stream;
    char *ds = P->stat.as<char>(), *dd = P->dyn.as<char>(), *dout = P->out.as<char>(), *d = P->schur.as<char>();
    const size_t n6 = 6 * (size_t)P->nkf;
    char* hp = P->schur_pin.as<char>();
    memcpy(hp, dpose, n6 * 8);
    double* ddp = (double*)(d + P->oDp);
    PLSLAM_HIP_CHECK(hipMemcpyAsync(ddp, hp, n6 * 8, hipMemcpyHostToDevice, s));
    double* dx = (double*)(d + P->oDx);
    SchurBackArgs A{};
    A.pt_ptr = (const int32_t*)(ds + P->oPtp); A.pt_obs = (const int32_t*)(ds + P->oPti); A.pt_kf = (const int32_t*)(ds + P->oPkf);
    A.ls_ptr = (const int32_t*)(ds + P->oLsp); A.ls_obs = (const int32_t*)(ds + P->oLsi); A.ls_kf = (const int32_t*)(ds + P->oLkf);
    A.W_pt = (const double*)(dout + P->oWp); A.W_ls = (const double*)(dout + P->oWl);
    A.Vp = (const double*)(d + P->oVp); A.Vl = (const double*)(d + P->oVl); A.tp = (const double*)(d + P->oTp); A.tl = (const double*)(d + P->oTl);
    A.dp = ddp; A.dx_pt = dx; A.dx_ls = dx + 3 * (size_t)P->npt; A.part = part_out ? part_out : (double*)(d + P->oDxPart);
    A.X = apply ? (double*)(dd + P->oX) : nullptr; A.L = apply ? (double*)(dd + P->oL) : nullptr;
    A.npt = P->npt; A.nls = P->nls; A.nwg_pt = (P->npt + BACK_WG / 3 - 1) / (BACK_WG / 3);
    const int32_t nwg = A.nwg_pt + (P->nls + BACK_WG / 6 - 1) / (BACK_WG / 6);
    if (nwg > 0) hipLaunchKernelGGL(k_schur_backsub, dim3(nwg), dim3(BACK_WG), 0, s, A);
    PLSLAM_HIP_CHECK(hipGetLastError());
    if (apply) P->schur_done = false;                      // (a second application of the same step would be a bug of the caller)
    return PLSLAM_OK;
}

extern "C" int plslam_lba_plan_backsub(plslam_lba_plan* P, const double* dpose, int apply, double* dX_pt, double* dX_ls)
{
    PLSLAM_REQUIRE(P && dpose, PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);
    PLSLAM_REQUIRE(P->schur_done, PLSLAM_EINVAL);          // plslam_lba_plan_schur on the blocks of the last iteration first
    PLSLAM_REQUIRE(!apply || P->state_valid, PLSLAM_EINVAL);
    hipStream_t s = ctx->stream;
    int rc = lba_backsub_enqueue(P, dpose, apply);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    char* hp = P->schur_pin.as<char>();
    const double* dx = (const double*)(P->schur.as<char>() + P->oDx);
    if (dX_pt || dX_ls) {
        const size_t bytes = (3 * (size_t)P->npt + 6 * (size_t)P->nls) * 8;
        if (bytes) PLSLAM_HIP_CHECK(hipMemcpyAsync(hp, dx, bytes, hipMemcpyDeviceToHost, s));
        PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
        if (dX_pt && P->npt) memcpy(dX_pt, hp, (size_t)P->npt * 24);
        if (dX_ls && P->nls) memcpy(dX_ls, hp + (size_t)P->npt * 24, (size_t)P->nls * 48);
    } else {
        PLSLAM_HIP_CHECK(hipStreamSynchronize(s));         // (the page-locked image of dp is the caller's to rewrite next)
    }
    return PLSLAM_OK;
}

// The rest of an LM iteration in ONE call and ONE synchronisation (round 6): the back-substitution of `dpose` (and X(i) += DX(i)
// when `apply`, :1570-1575 / :1801-1806), the pose slots for the next iteration (T_kf_w: n_slots x 16 or NULL to leave them --
// a rejected step), and sum_i DX_landmark(i)^2 for the loop's ||DX|| test (:1808) -- 8 bytes come back instead of the steps.
extern "C" int plslam_lba_plan_apply_step(plslam_lba_plan* P, const double* dpose, const double* T_kf_w, int apply, double* dx_sumsq)
{
    PLSLAM_REQUIRE(P && dpose, PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);
    PLSLAM_REQUIRE(P->schur_done, PLSLAM_EINVAL);
    PLSLAM_REQUIRE(P->state_valid || (!apply && !T_kf_w), PLSLAM_EINVAL);
    hipStream_t s = ctx->stream;
    // (the per-workgroup sums of squares: written in place behind dp's image when the image is mapped)
    const size_t n6 = 6 * (size_t)P->nkf;
    const bool in_place = dx_sumsq && P->schur_pin_dev;
    int rc = lba_backsub_enqueue(P, dpose, apply, in_place ? (double*)(P->schur_pin_dev + n6 * 8) : nullptr);
    if (rc) { (void)hipStreamSynchronize(s); return rc; }
    if (T_kf_w && P->n_slots) {
        char* hi = P->pin_in.as<char>();
        memcpy(hi + P->oT, T_kf_w, (size_t)P->n_slots * 128);
        PLSLAM_HIP_CHECK(hipMemcpyAsync(P->dyn.as<char>() + P->oT, hi + P->oT, (size_t)P->n_slots * 128, hipMemcpyHostToDevice, s));
    }
    char* hp = P->schur_pin.as<char>();
    // the back-substitution left one sum of squares per workgroup (220 at C3): they come down and are added here in workgroup
    // order -- a fixed order, and no launch for it
    const size_t nwg = (size_t)(P->npt + BACK_WG / 3 - 1) / (BACK_WG / 3) + (size_t)(P->nls + BACK_WG / 6 - 1) / (BACK_WG / 6);
    if (dx_sumsq && nwg && !in_place)
        PLSLAM_HIP_CHECK(hipMemcpyAsync(hp + n6 * 8, P->schur.as<char>() + P->oDxPart, nwg * 8, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    if (dx_sumsq) {
        const double* part = (const double*)(hp + n6 * 8);
        double acc = 0.0;
        for (size_t i = 0; i < nwg; ++i) acc += part[i];
        *dx_sumsq = acc;
    }
    return PLSLAM_OK;
}

// The resident landmarks, device -> host (round 6): what plslam_lba_plan_backsub(apply) has updated in place comes back -- into
// the plan's page-locked images first (plslam_lba_plan_host_state: the image no longer holds the landmarks of BEFORE the step, so
// a later plslam_lba_plan_iterate / _iterate_dev with the image's own pointers does not revert it -- ADVICE r5), then to the
// caller's arrays (either may be NULL).  The reference's write-back reads them there (src/mapHandler.cpp:1822-1852).
extern "C" int plslam_lba_plan_get_landmarks(plslam_lba_plan* P, double* Xw, double* Lw)
{
    PLSLAM_REQUIRE(P != nullptr, PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);
    PLSLAM_REQUIRE(P->state_valid, PLSLAM_EINVAL);         // one plslam_lba_plan_iterate(_dev) first: it uploads the state
    hipStream_t s = ctx->stream;
    char *hi = P->pin_in.as<char>(), *dd = P->dyn.as<char>();
    const size_t bx = (size_t)P->npt * 24, bl = (size_t)P->nls * 48;
    if (bx) PLSLAM_HIP_CHECK(hipMemcpyAsync(hi + P->oX, dd + P->oX, bx, hipMemcpyDeviceToHost, s));
    if (bl) PLSLAM_HIP_CHECK(hipMemcpyAsync(hi + P->oL, dd + P->oL, bl, hipMemcpyDeviceToHost, s));
    PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    if (Xw && bx && (char*)Xw != hi + P->oX) memcpy(Xw, hi + P->oX, bx);
    if (Lw && bl && (char*)Lw != hi + P->oL) memcpy(Lw, hi + P->oL, bl);
    return PLSLAM_OK;
}

// the optimised poses alone (the host has applied dp to them: expmap / logmap of SE(3) stay with the caller, :1560-1566)
extern "C" int plslam_lba_plan_set_poses(plslam_lba_plan* P, const double* T_kf_w)
{
    PLSLAM_REQUIRE(P && (P->n_slots == 0 || T_kf_w), PLSLAM_EINVAL);
    plslam_ctx* ctx = P->ctx;
    std::lock_guard<std::mutex> lk(ctx->mu);
    DeviceGuard dg_(ctx->device);
    PLSLAM_REQUIRE(P->state_valid, PLSLAM_EINVAL);
    hipStream_t s = ctx->stream;
    if (P->n_slots) {
        char* hi = P->pin_in.as<char>();
        memcpy(hi + P->oT, T_kf_w, (size_t)P->n_slots * 128);
        PLSLAM_HIP_CHECK(hipMemcpyAsync(P->dyn.as<char>() + P->oT, hi + P->oT, (size_t)P->n_slots * 128, hipMemcpyHostToDevice, s));
        PLSLAM_HIP_CHECK(hipStreamSynchronize(s));
    }
    return PLSLAM_OK;
}
