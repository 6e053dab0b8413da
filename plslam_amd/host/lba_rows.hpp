// lba_rows.hpp -- C++ host side of the local-BA row build, mirroring the data flow of
// PLSLAM::MapHandler::localBundleAdjustment / levMarquardtOptimizationLBA
// (src/mapHandler.cpp:1220-1330 gather, :1332-1540 first pass, :1587-1772 iteration pass).
//
// The reference walks `pt_obs_list` / `ls_obs_list` (one Vector6i per observation:
// [lm_idx, lm_loc, obs_idx, kf_idx, kf_loc, inlier], :1257-1264) and, per observation, computes the
// residual, the two Jacobians and the Cauchy weight, then adds 6x6 / 3x6 / 3x3 (6x6 for lines) blocks
// into a dense H and g.  Here the per-observation arithmetic runs on the MI355X
// (plslam_lba_point_rows / plslam_lba_line_rows); this class owns the SoA staging around it and the
// host-side accumulation with the reference's exact block placement (:1410-1429, :1519-1538).
// The LM control flow and the sparse LDLT solve stay with the caller, as in the reference.
#pragma once

#include <stdint.h>

#include <array>
#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "plslam_hip.h"

namespace PLSLAM {

typedef std::array<int, 6> Vector6i;  // include/mapHandler.h:63

struct LbaProblem {
    // local keyframes to optimise are kf_loc = 0..Nkf-1 (kf_idx 0 is never optimised, :1231);
    // poses_T_kf_w holds one row-major 4x4 per distinct KF referenced by any observation
    // ("pose slot"); kf_slot_of_obs maps an observation to its slot.
    std::vector<double> poses_T_kf_w;   // n_slots * 16
    std::vector<double> points;         // Npt * 3   (X block :1251-1253, or X.block(6Nkf+3*loc) :1597)
    std::vector<double> lines;          // Nls * 6
    std::vector<Vector6i> pt_obs_list, ls_obs_list;
    std::vector<int32_t> pt_pose_slot, ls_pose_slot;
    std::vector<double> pt_obs;         // Npt_obs * 2   map_points[..]->obs_list[..]   (:1375)
    std::vector<double> ls_obs;         // Nls_obs * 3   map_lines[..]->obs_list[..]    (:1456)
    int Nkf = 0;
};

struct LbaRows {
    std::vector<double> J_pose, J_lm, r, w;
};

class LbaRowBuilder {
public:
    LbaRowBuilder(plslam_ctx* ctx, const plslam_cam& cam, double homog_th) : ctx_(ctx), cam_(cam), th_(homog_th) {}

    // rows of every point observation (:1358-1407 / :1587-1642)
    void pointRows(const LbaProblem& p, LbaRows& out) const
    {
        const int32_t n = (int32_t)p.pt_obs_list.size();
        std::vector<int32_t> lm(n);
        for (int32_t o = 0; o < n; ++o) lm[o] = p.pt_obs_list[o][1];
        out.J_pose.resize((size_t)n * 6); out.J_lm.resize((size_t)n * 3); out.r.resize(n); out.w.resize(n);
        check(plslam_lba_point_rows(ctx_, &cam_, th_, p.poses_T_kf_w.data(), (int32_t)(p.poses_T_kf_w.size() / 16),
                                    p.points.data(), (int32_t)(p.points.size() / 3), p.pt_obs.data(), lm.data(),
                                    p.pt_pose_slot.data(), n, out.J_pose.data(), out.J_lm.data(), out.r.data(),
                                    out.w.data()), "plslam_lba_point_rows");
    }

    // rows of every line observation (:1436-1516; compat_iter_pass = the quirks of :1668-1748)
    void lineRows(const LbaProblem& p, bool compat_iter_pass, LbaRows& out) const
    {
        const int32_t n = (int32_t)p.ls_obs_list.size();
        std::vector<int32_t> lm(n);
        for (int32_t o = 0; o < n; ++o) lm[o] = p.ls_obs_list[o][1];
        out.J_pose.resize((size_t)n * 6); out.J_lm.resize((size_t)n * 6); out.r.resize(n); out.w.resize(n);
        check(plslam_lba_line_rows(ctx_, &cam_, th_, compat_iter_pass ? 1 : 0, p.poses_T_kf_w.data(),
                                   (int32_t)(p.poses_T_kf_w.size() / 16), p.lines.data(), (int32_t)p.lines.size(),
                                   p.ls_obs.data(), lm.data(), p.ls_pose_slot.data(), n, out.J_pose.data(),
                                   out.J_lm.data(), out.r.data(), out.w.data()), "plslam_lba_line_rows");
    }

    // dense accumulation exactly as :1410-1429 (dl = 3) and :1519-1538 (dl = 6); H is N x N row-major
    static void accumulate(const std::vector<Vector6i>& obs, const LbaRows& rows, int dl, int lm_base, int N,
                           std::vector<double>& H, std::vector<double>& g, double& err)
    {
        for (size_t o = 0; o < obs.size(); ++o) {
            const double* Jp = &rows.J_pose[o * 6];
            const double* Jl = &rows.J_lm[o * (size_t)dl];
            const int kf_loc = obs[o][4];
            const int idx = 6 * kf_loc, jdx = lm_base + dl * obs[o][1];
            const double r = rows.r[o], w = rows.w[o];
            for (int a = 0; a < dl; ++a) g[jdx + a] += Jl[a] * r * w;
            err += r * r * w;
            for (int a = 0; a < dl; ++a)
                for (int b = 0; b < dl; ++b) H[(size_t)(jdx + a) * N + jdx + b] += Jl[a] * Jl[b] * w;
            if (kf_loc == -1) continue;
            for (int a = 0; a < 6; ++a) g[idx + a] += Jp[a] * r * w;
            for (int a = 0; a < 6; ++a)
                for (int b = 0; b < 6; ++b) H[(size_t)(idx + a) * N + idx + b] += Jp[a] * Jp[b] * w;
            for (int a = 0; a < dl; ++a)
                for (int b = 0; b < 6; ++b) {
                    const double h = Jl[a] * Jp[b] * w;
                    H[(size_t)(jdx + a) * N + idx + b] += h;
                    H[(size_t)(idx + b) * N + jdx + a] += h;
                }
        }
    }

    // one H, g, err build = the body of the first pass (:1358-1540) or of an LM iteration (:1587-1772)
    void buildNormalEquations(const LbaProblem& p, bool iteration_pass, std::vector<double>& H,
                              std::vector<double>& g, double& err) const
    {
        const int Npt = (int)(p.points.size() / 3), Nls = (int)(p.lines.size() / 6);
        const int N = 6 * p.Nkf + 3 * Npt + 6 * Nls;
        H.assign((size_t)N * N, 0.0);
        g.assign((size_t)N, 0.0);
        err = 0.0;
        LbaRows rows;
        pointRows(p, rows);
        accumulate(p.pt_obs_list, rows, 3, 6 * p.Nkf, N, H, g, err);
        lineRows(p, iteration_pass, rows);
        accumulate(p.ls_obs_list, rows, 6, 6 * p.Nkf + 3 * Npt, N, H, g, err);
    }

    // Block (Schur-ready) form of the same normal equations, assembled on the device
    // (plslam_lba_assemble): H.block(idx,idx,6,6) per keyframe, H.block(jdx,jdx,.,.) per landmark,
    // one Haux (:1424, :1533) per observation.  Entries are bit-identical to the dense accumulation.
    struct BlockNormalEquations {
        std::vector<double> g, H_pose, H_pt, H_ls, W_pt, W_ls;
        double err = 0.0;
        int Nkf = 0, Npt = 0, Nls = 0;
    };
    void buildBlockNormalEquations(const LbaProblem& p, bool iteration_pass, BlockNormalEquations& out) const
    {
        out.Nkf = p.Nkf; out.Npt = (int)(p.points.size() / 3); out.Nls = (int)(p.lines.size() / 6);
        LbaRows rp, rl;
        pointRows(p, rp);
        lineRows(p, iteration_pass, rl);
        const int32_t np = (int32_t)p.pt_obs_list.size(), nl = (int32_t)p.ls_obs_list.size();
        std::vector<int32_t> plm(np), pkf(np), llm(nl), lkf(nl);
        for (int32_t o = 0; o < np; ++o) { plm[o] = p.pt_obs_list[o][1]; pkf[o] = p.pt_obs_list[o][4]; }
        for (int32_t o = 0; o < nl; ++o) { llm[o] = p.ls_obs_list[o][1]; lkf[o] = p.ls_obs_list[o][4]; }
        out.g.assign((size_t)(6 * out.Nkf + 3 * out.Npt + 6 * out.Nls), 0.0);
        out.H_pose.assign((size_t)out.Nkf * 36, 0.0); out.H_pt.assign((size_t)out.Npt * 9, 0.0);
        out.H_ls.assign((size_t)out.Nls * 36, 0.0); out.W_pt.assign((size_t)np * 18, 0.0);
        out.W_ls.assign((size_t)nl * 36, 0.0);
        check(plslam_lba_assemble(ctx_, out.Nkf, out.Npt, out.Nls, plm.data(), pkf.data(), np, rp.J_pose.data(),
                                  rp.J_lm.data(), rp.r.data(), rp.w.data(), llm.data(), lkf.data(), nl,
                                  rl.J_pose.data(), rl.J_lm.data(), rl.r.data(), rl.w.data(), out.g.data(),
                                  out.H_pose.data(), out.H_pt.data(), out.H_ls.data(), out.W_pt.data(),
                                  out.W_ls.data(), &out.err), "plslam_lba_assemble");
    }

private:
    static void check(int rc, const char* fn)
    {
        if (rc != PLSLAM_OK)
            throw std::runtime_error(std::string("[") + fn + "] " + plslam_strerror(rc) + ": " + plslam_last_error());
    }
    plslam_ctx* ctx_;
    plslam_cam cam_;
    double th_;
};

// The LM loop of levMarquardtOptimizationLBA on a device-resident plan (round 5): the observation lists go up once, an iteration
// leaves its blocks on the device, and the step of :1552-1575 / :1777-1807 -- damp H(i,i) += lambda H(i,i), solve, update -- is the
// Schur step of the C ABI: only the 6 Nkf x 6 Nkf reduced camera system comes down, a dense LDL^T of it runs here (the
// reference: SimplicialLDLT over all N unknowns, :1555-1556), the landmark steps and their update stay on the device.  The
// SE(3) maps of the pose update (:1560-1566: expmap_se3 / inverse_se3, stvo-pl) stay with the caller: setPoses() takes the
// updated matrices.
class LbaPlanSolver {
public:
    LbaPlanSolver(plslam_ctx* ctx, const plslam_cam& cam, double homog_th, const LbaProblem& p) : nkf_(p.Nkf)
    {
        npt_ = (int32_t)(p.points.size() / 3); nls_ = (int32_t)(p.lines.size() / 6);
        nslots_ = (int32_t)(p.poses_T_kf_w.size() / 16);
        const int32_t np = (int32_t)p.pt_obs_list.size(), nl = (int32_t)p.ls_obs_list.size();
        std::vector<int32_t> plm(np), pkf(np), llm(nl), lkf(nl);
        for (int32_t o = 0; o < np; ++o) { plm[o] = p.pt_obs_list[o][1]; pkf[o] = p.pt_obs_list[o][4]; }
        for (int32_t o = 0; o < nl; ++o) { llm[o] = p.ls_obs_list[o][1]; lkf[o] = p.ls_obs_list[o][4]; }
        check(plslam_lba_plan_create(ctx, &cam, homog_th, nslots_, nkf_, npt_, nls_, plm.data(), p.pt_pose_slot.data(), pkf.data(),
                                     p.pt_obs.data(), np, llm.data(), p.ls_pose_slot.data(), lkf.data(), p.ls_obs.data(), nl, &plan_),
              "plslam_lba_plan_create");
    }
    ~LbaPlanSolver() { plslam_lba_plan_destroy(plan_); }
    LbaPlanSolver(const LbaPlanSolver&) = delete;
    LbaPlanSolver& operator=(const LbaPlanSolver&) = delete;

    // H, g, err of the state in `p` (uploaded); iteration_pass: the quirks of :1668-1748.  The blocks stay on the device.
    double iterate(const LbaProblem& p, bool iteration_pass)
    {
        double err = 0;
        check(plslam_lba_plan_iterate_dev(plan_, p.poses_T_kf_w.data(), p.points.data(), p.lines.data(),
                                          iteration_pass ? PLSLAM_LBA_COMPAT_ITER_PASS : 0, nullptr, &err), "plslam_lba_plan_iterate_dev");
        return err;
    }
    // The plan's page-locked images of the state and of g: a solver whose X_aux (:1231-1330) lives THERE hands them to
    // iterateInPlace() and no staging copy is made on the host (plslam_lba_plan_host_state)
    plslam_lba_host_state hostState()
    {
        plslam_lba_host_state h{};
        check(plslam_lba_plan_host_state(plan_, &h), "plslam_lba_plan_host_state");
        return h;
    }
    // H, g, err of the state written into hostState()'s T_kf_w / Xw / Lw; the gradient lands in hostState().g
    double iterateInPlace(bool iteration_pass, bool want_g = true)
    {
        const plslam_lba_host_state h = hostState();
        double err = 0;
        check(plslam_lba_plan_iterate_dev(plan_, h.T_kf_w, h.Xw, h.Lw, iteration_pass ? PLSLAM_LBA_COMPAT_ITER_PASS : 0,
                                          want_g ? h.g : nullptr, &err), "plslam_lba_plan_iterate_dev");
        return err;
    }
    // the same on the state the device already holds (landmarks updated by solveStep(apply), poses by setPoses)
    double iterateResident(bool iteration_pass)
    {
        double err = 0;
        check(plslam_lba_plan_iterate_resident(plan_, iteration_pass ? PLSLAM_LBA_COMPAT_ITER_PASS : 0, &err), "plslam_lba_plan_iterate_resident");
        return err;
    }
    double diagMax()                                        // :1544-1550 "Hmax"
    {
        double h = 0;
        check(plslam_lba_plan_diag_max(plan_, &h), "plslam_lba_plan_diag_max");
        return h;
    }
    // damp, reduce, solve: dp (6 Nkf); the landmark steps where asked for; apply = true: X(i) += DX(i) on the device (:1570-1575)
    void solveStep(double lambda, std::vector<double>& dp, bool apply, std::vector<double>* dX_pt = nullptr,
                   std::vector<double>* dX_ls = nullptr, int32_t* n_singular = nullptr)
    {
        const size_t n = 6 * (size_t)nkf_;
        S_.resize(n * n); dp.resize(n);
        check(plslam_lba_plan_schur(plan_, lambda, S_.data(), dp.data(), n_singular), "plslam_lba_plan_schur");
        ldlt_solve(S_, dp, (int)n);
        if (dX_pt) dX_pt->resize((size_t)npt_ * 3);
        if (dX_ls) dX_ls->resize((size_t)nls_ * 6);
        check(plslam_lba_plan_backsub(plan_, dp.data(), apply ? 1 : 0, dX_pt ? dX_pt->data() : nullptr, dX_ls ? dX_ls->data() : nullptr),
              "plslam_lba_plan_backsub");
    }
    // An LM iteration in two calls and two synchronisations (round 6; what optimize() runs):
    //   iterateSchur: H, g, err on the RESIDENT state, the Schur step for `lambda` behind it, the dense LDL^T here -> dp (6 Nkf);
    //   applyStep: the back-substitution of dp (+ X(i) += DX(i) when apply), the pose slots for the next iteration (T_kf_w, or
    //   nullptr to leave them: a rejected step) -> sum of squares of the landmark steps (for ||DX||, :1808)
    double iterateSchur(double lambda, bool iteration_pass, std::vector<double>& dp, int32_t* n_singular = nullptr)
    {
        const size_t n = 6 * (size_t)nkf_;
        S_.resize(n * n); dp.resize(n);
        double err = 0;
        check(plslam_lba_plan_iterate_schur(plan_, iteration_pass ? PLSLAM_LBA_COMPAT_ITER_PASS : 0, lambda, &err, S_.data(), dp.data(),
                                            n_singular), "plslam_lba_plan_iterate_schur");
        ldlt_solve(S_, dp, (int)n);
        return err;
    }
    double applyStep(const std::vector<double>& dp, const std::vector<double>* T_kf_w, bool apply)
    {
        if (dp.size() != 6 * (size_t)nkf_ || (T_kf_w && T_kf_w->size() != (size_t)nslots_ * 16))
            throw std::runtime_error("[LbaPlanSolver::applyStep] dp: 6 doubles per optimised key frame; T_kf_w: one 4 x 4 per pose slot");
        double s2 = 0;
        check(plslam_lba_plan_apply_step(plan_, dp.data(), T_kf_w ? T_kf_w->data() : nullptr, apply ? 1 : 0, &s2), "plslam_lba_plan_apply_step");
        return s2;
    }
    void setPoses(const std::vector<double>& T_kf_w)        // n_slots x 16, after T <- T inv(exp(dp)) on the host
    {
        if (T_kf_w.size() != (size_t)nslots_ * 16) throw std::runtime_error("[LbaPlanSolver::setPoses] one 4 x 4 per pose slot");
        check(plslam_lba_plan_set_poses(plan_, T_kf_w.data()), "plslam_lba_plan_set_poses");
    }

    // ---- the LM loop as the reference runs it (round 6) ----------------------------------------------------------------------
    // levMarquardtOptimizationLBA's control flow (src/mapHandler.cpp:1332-1812) on the resident plan:
    //   first pass (:1358-1540): H, g, err at the stored state; err /= (Npt_obs + Nls_obs) (:1541) -- two counters the text
    //   declares (:1355, :1433) and never increments: the first err is +inf (NaN for a zero sum), so the first iteration's step is
    //   always accepted (inf > inf is false) and its |err - err_prev| test never fires.  Kept: it decides the control flow.
    //   lambda = lambdaLbaLM * max_i |H(i,i)| (:1544-1550); damped solve and the step applied unconditionally (:1552-1575);
    //   err_prev = err (:1578); then up to max_iters - 1 iterations (:1583-1812):
    //     H, g, err with the iteration pass's quirks; err /= (Npt + Nls) -- LANDMARKS, not observations (:1773);
    //     stop if |err - err_prev| < minErrorChange or err < minError (:1775);
    //     damped solve (:1778-1783); err > err_prev: lambda /= lambda_k and the step is NOT applied; otherwise lambda *= lambda_k
    //     and the step is applied (:1786-1806: yes, lambda GROWS on success -- the reference's schedule, kept);
    //     stop if ||DX|| < minErrorChange (:1808); err_prev = err (:1811) -- also after a rejected step, so the iteration after
    //     a rejection recomputes the same err and stops at the first test.
    // The pose update X_k <- logmap(expmap(X_k) * inverse(expmap(DX_k))) (:1560-1566, :1793-1799) uses stvo-pl's SE(3) maps, which
    // stay with the caller: `maps` holds them (in PL-SLAM: expmap_se3 / logmap_se3 / inverse_se3 of stvo-pl's auxiliar.h).
    // Pose slots (plslam_lba_plan_create): slot `first_estimate_slot + k` holds expmap(X_k) of local key frame k -- what the point
    // rows of the iteration pass read (:1600-1601); the other slots keep the stored T_kf_w (line rows, :1680, and key frames that
    // are not optimised).  x_kf (6 Nkf) = the key frames' x_kf_w (:1233), updated in place; p.points / p.lines receive the final
    // landmarks.  What the reference solves with SimplicialLDLT over all N unknowns is solved here by the Schur step of the C ABI
    // (landmark blocks on the device, a dense LDL^T of the 6 Nkf x 6 Nkf reduced system on the host): equal to rounding.
    struct LmParams {                    // SlamConfig::lambdaLbaLM(), lambdaLbaK(), maxItersLba(); Config::minErrorChange(), minError()
        double lambda_lba_lm = 0.00001, lambda_lba_k = 10.0;
        int max_iters_lba = 15;
        double min_error_change = 1e-7, min_error = 1e-7;
    };
    struct Se3Maps {
        void (*expmap)(const double x[6], double T[16]);
        void (*logmap)(const double T[16], double x[6]);
        void (*inverse)(const double T[16], double Ti[16]);
    };
    struct LmTrace {                     // one entry per H / g build: [0] = the first pass
        std::vector<double> err, lambda;       // err as the reference normalises it; lambda used by that build's solve (0: none)
        std::vector<int> applied;              // 1: the step was applied, 0: rejected (err > err_prev), -1: stopped before the solve
        int iters = 0;                         // the reference's `iters` when its loop ends
        int stop = 0;                          // 0 max_iters reached, 1 |err - err_prev| / err test, 2 ||DX|| test
        int n_singular = 0;                    // landmarks whose damped block was not positive definite in any solve
    };
    void optimize(LbaProblem& p, std::vector<double>& x_kf, int32_t first_estimate_slot, const LmParams& prm, const Se3Maps& maps,
                  LmTrace* trace = nullptr)
    {
        if (x_kf.size() != 6 * (size_t)nkf_ || first_estimate_slot < 0 || first_estimate_slot + nkf_ > nslots_)
            throw std::runtime_error("[LbaPlanSolver::optimize] x_kf must hold 6 doubles per optimised key frame, and their estimate slots must exist");
        const double n_obs_counted = 0.0;                             // Npt_obs + Nls_obs as the reference leaves them (see above)
        const double n_lm = (double)(npt_ + nls_);
        LmTrace local;
        LmTrace& tr = trace ? *trace : local;
        tr = LmTrace();
        std::vector<double> dp;
        auto set_estimates = [&]() {                                  // slot first_estimate_slot + k <- expmap(X_k)
            for (int32_t k = 0; k < nkf_; ++k) maps.expmap(&x_kf[6 * (size_t)k], &p.poses_T_kf_w[16 * (size_t)(first_estimate_slot + k)]);
        };
        auto update_poses = [&]() {                                   // :1560-1566 (the matrices go up with the step)
            double Tprev[16], Tinc[16], Tinv[16], Tcur[16];
            for (int32_t k = 0; k < nkf_; ++k) {
                maps.expmap(&x_kf[6 * (size_t)k], Tprev);
                maps.expmap(&dp[6 * (size_t)k], Tinc);
                maps.inverse(Tinc, Tinv);
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < 4; ++j) {
                        double a = 0.0;
                        for (int q = 0; q < 4; ++q) a += Tprev[4 * i + q] * Tinv[4 * q + j];
                        Tcur[4 * i + j] = a;
                    }
                maps.logmap(Tcur, &x_kf[6 * (size_t)k]);
            }
            set_estimates();
        };
        // ---- first pass ----
        set_estimates();
        double err = iterate(p, false) / n_obs_counted;
        double lambda = prm.lambda_lba_lm * diagMax();
        int32_t nsing = 0;
        const size_t n6 = 6 * (size_t)nkf_;
        S_.resize(n6 * n6); dp.resize(n6);
        check(plslam_lba_plan_schur(plan_, lambda, S_.data(), dp.data(), &nsing), "plslam_lba_plan_schur");
        ldlt_solve(S_, dp, (int)n6);
        tr.n_singular += nsing;
        update_poses();
        check(plslam_lba_plan_apply_step(plan_, dp.data(), p.poses_T_kf_w.data(), 1, nullptr), "plslam_lba_plan_apply_step");
        tr.err.push_back(err); tr.lambda.push_back(lambda); tr.applied.push_back(1);
        double err_prev = err;
        // ---- LM iterations ----
        int iters;
        // an iteration = two calls, two synchronisations: [H, g, err + Schur step] -> dense LDL^T here -> [back-substitution,
        // landmark update, pose slots, sum DX^2].  (The Schur step of the build that stops at the first test is computed and unused.)
        for (iters = 1; iters < prm.max_iters_lba; ++iters) {
            check(plslam_lba_plan_iterate_schur(plan_, PLSLAM_LBA_COMPAT_ITER_PASS, lambda, &err, S_.data(), dp.data(), &nsing),
                  "plslam_lba_plan_iterate_schur");     // (iterateSchur() without the solve: the first stop test comes before it)
            err /= n_lm;
            tr.err.push_back(err);
            if (std::fabs(err - err_prev) < prm.min_error_change || err < prm.min_error) {
                tr.lambda.push_back(0.0); tr.applied.push_back(-1); tr.stop = 1;
                break;
            }
            const bool reject = err > err_prev;
            ldlt_solve(S_, dp, (int)n6);
            tr.n_singular += nsing;
            tr.lambda.push_back(lambda); tr.applied.push_back(reject ? 0 : 1);
            if (reject) lambda /= prm.lambda_lba_k;
            else { lambda *= prm.lambda_lba_k; update_poses(); }
            double s2 = 0.0, s2_lm = 0.0;
            for (double v : dp) s2 += v * v;
            check(plslam_lba_plan_apply_step(plan_, dp.data(), reject ? nullptr : p.poses_T_kf_w.data(), reject ? 0 : 1, &s2_lm),
                  "plslam_lba_plan_apply_step");
            if (std::sqrt(s2 + s2_lm) < prm.min_error_change) { tr.stop = 2; break; }      // ||DX|| over all N unknowns (:1808)
            err_prev = err;
        }
        tr.iters = iters;
        landmarks(p.points, p.lines);
    }
    // the reference's write-back test (:1822-1846): landmark i is flagged when its estimate moved by more than `th` (0.01)
    static void movedLandmarks(const std::vector<double>& before, const std::vector<double>& after, int dim, double th, std::vector<uint8_t>& moved)
    {
        const size_t n = before.size() / (size_t)dim;
        moved.assign(n, 0);
        for (size_t i = 0; i < n; ++i) {
            double s2 = 0.0;
            for (int a = 0; a < dim; ++a) { const double d = after[i * dim + a] - before[i * dim + a]; s2 += d * d; }
            moved[i] = std::sqrt(s2) > th ? 1 : 0;
        }
    }
    // the resident landmarks (after solveStep(apply) / optimize())
    void landmarks(std::vector<double>& Xw, std::vector<double>& Lw)
    {
        Xw.resize((size_t)npt_ * 3); Lw.resize((size_t)nls_ * 6);
        check(plslam_lba_plan_get_landmarks(plan_, Xw.data(), Lw.data()), "plslam_lba_plan_get_landmarks");
    }

    // x <- A^-1 x for a symmetric positive definite A (n x n, row-major, overwritten): LDL^T without pivoting
    static void ldlt_solve(std::vector<double>& A, std::vector<double>& x, int n)
    {
        for (int j = 0; j < n; ++j) {
            double d = A[(size_t)j * n + j];
            for (int k = 0; k < j; ++k) d -= A[(size_t)j * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
            if (!(d > 0.0)) throw std::runtime_error("[LbaPlanSolver] the reduced camera system is not positive definite");
            A[(size_t)j * n + j] = d;
            for (int i = j + 1; i < n; ++i) {
                double l = A[(size_t)i * n + j];
                for (int k = 0; k < j; ++k) l -= A[(size_t)i * n + k] * A[(size_t)j * n + k] * A[(size_t)k * n + k];
                A[(size_t)i * n + j] = l / d;
            }
        }
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < i; ++k) x[i] -= A[(size_t)i * n + k] * x[k];
        for (int i = 0; i < n; ++i) x[i] /= A[(size_t)i * n + i];
        for (int i = n - 1; i >= 0; --i)
            for (int k = i + 1; k < n; ++k) x[i] -= A[(size_t)k * n + i] * x[k];
    }

private:
    static void check(int rc, const char* fn)
    {
        if (rc != PLSLAM_OK)
            throw std::runtime_error(std::string("[") + fn + "] " + plslam_strerror(rc) + ": " + plslam_last_error());
    }
    plslam_lba_plan* plan_ = nullptr;
    int32_t nkf_, npt_ = 0, nls_ = 0, nslots_ = 0;
    std::vector<double> S_;
};

}  // namespace PLSLAM
