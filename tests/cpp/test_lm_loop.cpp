// Runs PLSLAM::LbaPlanSolver::optimize (plslam_amd/host/lba_rows.hpp) -- the LM loop of levMarquardtOptimizationLBA,
// src/mapHandler.cpp:1334-1812, on the device-resident plan -- on a problem file written by tests/test_gpu_lba_lm.py from the
// fixture tests/golden/lba_lm_golden.npz (the REFERENCE'S OWN text run on the same inputs), and writes its trace and final state.
// stvo-pl's SE(3) maps are not in the reference tree: the checker's restatements (oracle/plslam_oracle.c) stand in for them here,
// exactly as they do inside the fixture's generator.
// usage: test_lm_loop <problem.bin> <result.bin>
//        test_lm_loop <problem.bin> --time <reps>      one LM iteration with state and blocks resident, host to host, as optimize()
//                                                      runs it (iterateSchur + applyStep without the update): microseconds, printed
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../oracle/plslam_oracle.h"
#include "../../plslam_amd/host/lba_rows.hpp"

template <class T>
static void rd(FILE* f, std::vector<T>& v, size_t n)
{
    v.resize(n);
    if (n && fread(v.data(), sizeof(T), n, f) != n) { std::fprintf(stderr, "short problem file\n"); std::exit(2); }
}
template <class T>
static void wr(FILE* f, const std::vector<T>& v) { if (!v.empty()) fwrite(v.data(), sizeof(T), v.size(), f); }

int main(int argc, char** argv)
{
    if (argc != 3 && !(argc == 4 && std::string(argv[2]) == "--time")) {
        std::fprintf(stderr, "usage: %s <problem.bin> <result.bin> | <problem.bin> --time <reps>\n", argv[0]);
        return 2;
    }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { std::perror(argv[1]); return 2; }
    std::vector<int32_t> hdr;
    rd(f, hdr, 6);
    const int nkf = hdr[0], n_kf_map = hdr[1], npt = hdr[2], nls = hdr[3], npo = hdr[4], nlo = hdr[5];
    std::vector<double> cfg, cam4, T_map, x_kf, Xw, Lw, pt_uv, ls_l;
    std::vector<int32_t> pt_lm, pt_kf_map, pt_kf_loc, ls_lm, ls_kf_map, ls_kf_loc;
    rd(f, cfg, 6); rd(f, cam4, 4);
    rd(f, T_map, (size_t)n_kf_map * 16); rd(f, x_kf, (size_t)nkf * 6); rd(f, Xw, (size_t)npt * 3); rd(f, Lw, (size_t)nls * 6);
    rd(f, pt_lm, npo); rd(f, pt_kf_map, npo); rd(f, pt_kf_loc, npo); rd(f, pt_uv, (size_t)npo * 2);
    rd(f, ls_lm, nlo); rd(f, ls_kf_map, nlo); rd(f, ls_kf_loc, nlo); rd(f, ls_l, (size_t)nlo * 3);
    fclose(f);

    // pose slots: [0, n_kf_map) the stored T_kf_w of every key frame (line rows, :1680; key frames that are not optimised);
    // [n_kf_map, n_kf_map + nkf) the current estimates expmap(X_k) of the optimised ones (point rows, :1600-1601)
    PLSLAM::LbaProblem p;
    p.Nkf = nkf;
    p.poses_T_kf_w.assign(T_map.begin(), T_map.end());
    p.poses_T_kf_w.resize((size_t)(n_kf_map + nkf) * 16, 0.0);
    p.points = Xw; p.lines = Lw; p.pt_obs = pt_uv; p.ls_obs = ls_l;
    std::vector<int> seen_p(npt, 0), seen_l(nls, 0);
    for (int o = 0; o < npo; ++o) {
        p.pt_obs_list.push_back({pt_lm[o], pt_lm[o], seen_p[pt_lm[o]]++, pt_kf_map[o], pt_kf_loc[o], 1});
        p.pt_pose_slot.push_back(pt_kf_loc[o] >= 0 ? n_kf_map + pt_kf_loc[o] : pt_kf_map[o]);
    }
    for (int o = 0; o < nlo; ++o) {
        p.ls_obs_list.push_back({ls_lm[o], ls_lm[o], seen_l[ls_lm[o]]++, ls_kf_map[o], ls_kf_loc[o], 1});
        p.ls_pose_slot.push_back(ls_kf_map[o]);
    }
    plslam_ctx* ctx = nullptr;
    if (plslam_ctx_create(0, &ctx) != PLSLAM_OK) { std::fprintf(stderr, "plslam_ctx_create: %s\n", plslam_last_error()); return 3; }
    plslam_cam cam{};
    cam.fx = cam4[0]; cam.fy = cam4[1]; cam.cx = cam4[2]; cam.cy = cam4[3];
    int rc = 0;
    bool timed = false;
    try {
        PLSLAM::LbaPlanSolver::LmParams prm;
        prm.lambda_lba_lm = cfg[1]; prm.lambda_lba_k = cfg[2]; prm.max_iters_lba = (int)cfg[3];
        prm.min_error_change = cfg[4]; prm.min_error = cfg[5];
        const PLSLAM::LbaPlanSolver::Se3Maps maps = {plo_expmap_se3, plo_logmap_se3, plo_inverse_se3};
        const std::vector<double> Xw0 = p.points, Lw0 = p.lines;
        if (argc == 4) {
            PLSLAM::LbaPlanSolver solver(ctx, cam, cfg[0], p);
            const int reps = std::atoi(argv[3]);
            for (int k = 0; k < nkf; ++k) plo_expmap_se3(&x_kf[6 * (size_t)k], &p.poses_T_kf_w[16 * (size_t)(n_kf_map + k)]);
            solver.iterate(p, false);
            const double lambda = 1e-5 * solver.diagMax();
            std::vector<double> dp, us;
            for (int it = 0; it < reps + 10; ++it) {
                const auto t0 = std::chrono::steady_clock::now();
                solver.iterateSchur(lambda, true, dp);
                solver.applyStep(dp, nullptr, false);
                const auto t1 = std::chrono::steady_clock::now();
                if (it >= 10) us.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
            }
            std::sort(us.begin(), us.end());
            std::printf("LM iteration resident: median %.1f us, p10 %.1f, p90 %.1f over %d\n", us[us.size() / 2], us[us.size() / 10],
                        us[us.size() * 9 / 10], reps);
            timed = true;
        }
        if (timed) { plslam_ctx_destroy(ctx); return 0; }     // (the solver has died first: its plan belongs to the context)
        PLSLAM::LbaPlanSolver solver(ctx, cam, cfg[0], p);
        PLSLAM::LbaPlanSolver::LmTrace tr;
        solver.optimize(p, x_kf, n_kf_map, prm, maps, &tr);
        std::vector<uint8_t> moved_p, moved_l;
        PLSLAM::LbaPlanSolver::movedLandmarks(Xw0, p.points, 3, 0.01, moved_p);      // :1825-1827
        PLSLAM::LbaPlanSolver::movedLandmarks(Lw0, p.lines, 6, 0.01, moved_l);       // :1840-1842
        FILE* o = fopen(argv[2], "wb");
        if (!o) { std::perror(argv[2]); return 2; }
        const std::vector<int32_t> oh = {(int32_t)tr.err.size(), tr.iters, tr.stop, tr.n_singular};
        wr(o, oh); wr(o, tr.err); wr(o, tr.lambda);
        std::vector<int32_t> ap(tr.applied.begin(), tr.applied.end());
        wr(o, ap); wr(o, x_kf); wr(o, p.points); wr(o, p.lines); wr(o, moved_p); wr(o, moved_l);
        fclose(o);
        std::printf("LM loop: %d builds, iters %d, stop %d, final err %.9g\n", (int)tr.err.size(), tr.iters, tr.stop, tr.err.back());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "%s\n", e.what());
        rc = 4;
    }
    plslam_ctx_destroy(ctx);
    return rc;
}
