// Exercises the C++ host shim (plslam_amd/host) the way the reference's callers use the matcher
// (src/mapHandler.cpp:277-283) and the LBA row build (:1358-1540), and checks every result against
// the CPU oracle.  Built and run by tests/test_gpu_host_shim.py on a GPU box.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../oracle/plslam_oracle.h"
#include "../../plslam_amd/host/lba_rows.hpp"
#include "../../plslam_amd/host/map_features.hpp"
#include "../../plslam_amd/host/stvo_match.hpp"

static int g_fail = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++g_fail; } } while (0)

static std::vector<uint8_t> rand_desc(std::mt19937& g, int n)
{
    std::vector<uint8_t> d((size_t)n * 32);
    for (auto& b : d) b = (uint8_t)(g() & 0xFF);
    return d;
}

static void noisy(std::mt19937& g, const std::vector<uint8_t>& src, std::vector<uint8_t>& dst)
{
    dst = src;
    std::bernoulli_distribution flip(0.06);
    for (auto& b : dst)
        for (int k = 0; k < 8; ++k)
            if (flip(g)) b ^= (uint8_t)(1u << k);
}

static void match_case(int n1, int n2, float nnr, bool mutual, unsigned seed)
{
    std::mt19937 g(seed);
    std::vector<uint8_t> a = rand_desc(g, n1), b;
    std::vector<uint8_t> base = rand_desc(g, n2);
    b = base;
    if (n1 && n2) {
        std::vector<uint8_t> na;
        noisy(g, a, na);
        for (int i = 0; i < std::min(n1, n2) / 2; ++i) std::copy(&na[(size_t)i * 32], &na[(size_t)i * 32 + 32], &b[(size_t)i * 32]);
    }
    StVO::DescMat d1(a.data(), n1), d2(b.data(), n2);
    StVO::bestLRMatches() = mutual;
    std::vector<int> m12;
    const int n = StVO::match(d1, d2, nnr, m12);
    std::vector<int32_t> ref((size_t)n1);
    const int nref = plo_match(a.data(), n1, b.data(), n2, nnr, mutual ? 1 : 0, ref.data());
    EXPECT(n == nref);
    EXPECT((int)m12.size() == n1);
    for (int i = 0; i < n1; ++i) EXPECT(m12[i] == ref[i]);
    // the caller-side contract of src/mapHandler.cpp:280-283
    int cnt = 0;
    for (int i1 = 0; i1 < (int)m12.size(); ++i1) {
        const int i2 = m12[i1];
        if (i2 < 0) continue;
        EXPECT(i2 < n2);
        ++cnt;
    }
    EXPECT(cnt == n);
    // the SAME vector handed to match() again, holding an earlier table (what mapHandler.cpp:271+277 does with
    // matchGrid's result): rows the ratio test rejects keep their entry, stvo-pl's resize() semantics
    if (n1 > 0 && n2 > 0) {
        std::vector<int> again((size_t)n1);
        std::vector<int32_t> ref2((size_t)n1);
        for (int i = 0; i < n1; ++i) ref2[(size_t)i] = again[(size_t)i] = (i % 3 == 0) ? (int)(g() % (unsigned)n2) : -1;
        const int n_again = StVO::match(d1, d2, nnr, again);
        const int nref2 = plo_match_prior(a.data(), n1, b.data(), n2, nnr, mutual ? 1 : 0, ref2.data());
        EXPECT(n_again == nref2);
        for (int i = 0; i < n1; ++i) EXPECT(again[(size_t)i] == ref2[(size_t)i]);
    }
}

// StVO::matchGrid used exactly as MapHandler::matchKF2KFPoints / matchKF2KFLines do (src/mapHandler.cpp:252-271,
// :381-418): project, fill the GridStructure, window of matching_f2f_ws cells, match; against the oracle's literal
// sequential restatement fed with the same grid in CSR form.
static void grid_case(bool lines, int n1, int n2, bool mutual, unsigned seed)
{
    std::mt19937 g(seed);
    std::uniform_real_distribution<double> ux(-20.0, 772.0), uy(-20.0, 500.0), ul(0.0, 180.0), ua(0.0, 3.14159);
    const double inv_width = GRID_COLS / 752.0, inv_height = GRID_ROWS / 480.0;
    std::vector<uint8_t> b = rand_desc(g, n2), a;
    noisy(g, b, a);
    a.resize((size_t)n1 * 32);
    for (size_t k = (size_t)std::min(n1, n2) * 32; k < a.size(); ++k) a[k] = (uint8_t)(g() & 0xFF);
    StVO::DescMat d1(a.data(), n1), d2(b.data(), n2);
    StVO::bestLRMatches() = mutual;
    StVO::GridStructure grid(GRID_ROWS, GRID_COLS);
    StVO::GridWindow w;
    w.width = std::make_pair(3, 3);
    w.height = std::make_pair(3, 3);
    std::vector<int> m12;
    int n = 0;
    std::vector<int32_t> centres;
    std::vector<double> dir1, dir2;
    if (!lines) {
        std::vector<StVO::point_2d> pj_points;
        std::vector<std::pair<double, double>> px2((size_t)n2);
        for (int i = 0; i < n2; ++i) {
            px2[i] = std::make_pair(ux(g), uy(g));
            grid.at(px2[i].first * inv_width, px2[i].second * inv_height).push_back(i);
        }
        for (int i = 0; i < n1; ++i) {
            const double x = i < n2 ? px2[i].first + 5.0 : ux(g), y = i < n2 ? px2[i].second - 4.0 : uy(g);
            pj_points.push_back(std::make_pair(x * inv_width, y * inv_height));
            centres.push_back(pj_points.back().first);
            centres.push_back(pj_points.back().second);
        }
        n = StVO::matchGrid(pj_points, d1, grid, d2, w, m12, 0.8f);
    } else {
        std::vector<StVO::line_2d> pj_lines;
        std::vector<std::pair<double, double>> directions((size_t)n2);
        std::vector<double> seg((size_t)n2 * 4);
        std::list<StVO::point_2d> line_coords;
        for (int i = 0; i < n2; ++i) {
            const double x = ux(g), y = uy(g), l = ul(g), t = ua(g);
            seg[4 * i] = x; seg[4 * i + 1] = y; seg[4 * i + 2] = x + l * std::cos(t); seg[4 * i + 3] = y + l * std::sin(t);
            std::pair<double, double>& v = directions[i];
            v = std::make_pair((seg[4 * i + 2] - x) * inv_width, (seg[4 * i + 3] - y) * inv_height);
            StVO::normalize(v);
            StVO::getLineCoords(x * inv_width, y * inv_height, seg[4 * i + 2] * inv_width, seg[4 * i + 3] * inv_height, line_coords);
            for (const StVO::point_2d& p : line_coords) grid.at(p.first, p.second).push_back(i);
            dir2.push_back(v.first);
            dir2.push_back(v.second);
        }
        for (int i = 0; i < n1; ++i) {
            double s[4];
            for (int k = 0; k < 4; ++k) s[k] = i < n2 ? seg[4 * i + k] + 3.0 : (k & 1 ? uy(g) : ux(g));
            pj_lines.push_back(std::make_pair(std::make_pair(s[0] * inv_width, s[1] * inv_height),
                                              std::make_pair(s[2] * inv_width, s[3] * inv_height)));
            const StVO::line_2d& L = pj_lines.back();
            centres.push_back(L.first.first); centres.push_back(L.first.second);
            centres.push_back(L.second.first); centres.push_back(L.second.second);
            double v[2] = {(double)(L.second.first - L.first.first), (double)(L.second.second - L.first.second)};
            plo_normalize2(v);
            dir1.push_back(v[0]);
            dir1.push_back(v[1]);
        }
        n = StVO::matchGrid(pj_lines, d1, grid, d2, directions, w, m12, 0.8f);
    }
    std::vector<int32_t> cs, items, ref((size_t)n1);
    grid.toCSR(cs, items);
    const int32_t win[4] = {3, 3, 3, 3};
    const int nref = plo_match_grid(centres.data(), lines ? 2 : 1, a.data(), n1, cs.data(), items.data(), GRID_COLS, GRID_ROWS,
                                    b.data(), n2, lines ? dir1.data() : nullptr, lines ? dir2.data() : nullptr,
                                    StVO::lineSimTh(), win, (double)0.8f, mutual ? 1 : 0, ref.data());
    EXPECT(n == nref);
    EXPECT(nref > 0);
    EXPECT((int)m12.size() == n1);
    for (int i = 0; i < n1; ++i) EXPECT(m12[i] == ref[i]);
}

int main()
{
    // --- StVO::matchGrid drop-in ------------------------------------------------------------
    grid_case(false, 1500, 1400, true, 11);
    grid_case(false, 800, 900, false, 12);
    grid_case(true, 200, 220, true, 13);
    grid_case(true, 150, 100, false, 14);
    {   // getLineCoords against the oracle's restatement
        std::mt19937 g(5);
        std::uniform_real_distribution<double> u(-3.0, 70.0);
        for (int k = 0; k < 200; ++k) {
            const double x1 = u(g), y1 = u(g), x2 = u(g), y2 = u(g);
            std::list<StVO::point_2d> lc;
            StVO::getLineCoords(x1, y1, x2, y2, lc);
            std::vector<int32_t> out(2 * 256);
            const int n = plo_get_line_coords(x1, y1, x2, y2, out.data(), 256);
            EXPECT(n == (int)lc.size());
            int j = 0;
            for (const auto& p : lc) { EXPECT(p.first == out[2 * j] && p.second == out[2 * j + 1]); ++j; }
        }
    }
    // --- StVO::match drop-in -------------------------------------------------------------
    match_case(1500, 1500, 0.75f, true, 1);
    match_case(200, 200, 0.9f, true, 2);
    match_case(777, 301, 0.75f, false, 3);
    match_case(5, 1, 0.9f, true, 4);      // n2 < 2: defined as "no match"
    match_case(0, 10, 0.9f, true, 5);
    match_case(10, 0, 0.9f, true, 6);

    // error behaviour: exceptions, like the reference (std::runtime_error)
    {
        StVO::DescMat bad(reinterpret_cast<const uint8_t*>("x"), 3);
        bad.cols = 16;
        std::vector<int> m;
        bool threw = false;
        try { StVO::match(bad, bad, 0.9f, m); } catch (const std::runtime_error&) { threw = true; }
        EXPECT(threw);
    }

    // concurrent callers (VO thread, local mapping, loop closure): one context per thread
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 3; ++t) th.emplace_back([t] { for (int k = 0; k < 4; ++k) match_case(400 + 37 * t, 390, 0.75f, true, 100 + 10 * t + k); });
        for (auto& x : th) x.join();
        // the finished threads handed their contexts back: a second wave re-uses them, shutdown() destroys all of them
        // and the next call starts over
        std::vector<std::thread> th2;
        for (int t = 0; t < 3; ++t) th2.emplace_back([t] { match_case(300 + t, 310, 0.9f, true, 500 + t); });
        for (auto& x : th2) x.join();
        StVO::shutdown();
        match_case(257, 300, 0.75f, true, 601);
    }

    // batch of jobs
    {
        std::mt19937 g(77);
        std::vector<std::vector<uint8_t>> store;
        std::vector<StVO::MatchJob> jobs;
        const int sizes[][2] = {{300, 280}, {0, 5}, {64, 64}, {1, 2}, {129, 1000}};
        for (auto& s : sizes) {
            store.push_back(rand_desc(g, s[0]));
            store.push_back(rand_desc(g, s[1]));
        }
        for (size_t b = 0; b < 5; ++b)
            jobs.push_back({StVO::DescMat(store[2 * b].data(), sizes[b][0]), StVO::DescMat(store[2 * b + 1].data(), sizes[b][1])});
        StVO::bestLRMatches() = true;
        std::vector<std::vector<int>> out;
        std::vector<int> counts = StVO::matchBatch(jobs, 0.9f, out);
        for (size_t b = 0; b < 5; ++b) {
            std::vector<int32_t> ref((size_t)sizes[b][0]);
            const int nref = plo_match(store[2 * b].data(), sizes[b][0], store[2 * b + 1].data(), sizes[b][1], 0.9f, 1, ref.data());
            EXPECT(counts[b] == nref);
            for (int i = 0; i < sizes[b][0]; ++i) EXPECT(out[b][i] == ref[i]);
        }
    }

    // --- representative descriptors of a local map + LBD binary rows (map_features.hpp) -----
    {
        std::mt19937 g(91);
        const int n_lm = 257;
        std::vector<std::vector<uint8_t>> obs;          // every observation row
        std::vector<std::vector<const uint8_t*>> ptrs(n_lm);
        std::vector<int> lens(n_lm);
        for (int l = 0; l < n_lm; ++l) lens[l] = (int)(g() % 9);   // 0..8 observations (0: never in the reference)
        for (int l = 0; l < n_lm; ++l) {
            std::vector<uint8_t> base = rand_desc(g, 1);
            for (int k = 0; k < lens[l]; ++k) {
                std::vector<uint8_t> d;
                noisy(g, base, d);
                if (l % 5 == 0 && k) d = obs.back();                // exact duplicates: ties, first row must win
                obs.push_back(d);
            }
        }
        size_t o = 0;
        for (int l = 0; l < n_lm; ++l)
            for (int k = 0; k < lens[l]; ++k) ptrs[l].push_back(obs[o++].data());
        std::vector<PLSLAM::DescList> lms(n_lm);
        for (int l = 0; l < n_lm; ++l) { lms[l].rows = ptrs[l].data(); lms[l].n = lens[l]; }
        std::vector<int> med_idx;
        std::vector<uint8_t> med_desc;
        PLSLAM::updateAverageDescriptors(lms, med_idx, &med_desc);
        EXPECT((int)med_idx.size() == n_lm);
        for (int l = 0; l < n_lm; ++l) {
            std::vector<uint8_t> flat;
            for (int k = 0; k < lens[l]; ++k) flat.insert(flat.end(), ptrs[l][k], ptrs[l][k] + 32);
            const int ref = lens[l] ? plo_median_desc(flat.data(), lens[l]) : -1;
            EXPECT(med_idx[l] == ref);
            for (int b = 0; b < 32; ++b) EXPECT(med_desc[(size_t)l * 32 + b] == (lens[l] ? ptrs[l][ref][b] : 0));
        }
        // LBD rows
        const int nl = 203;
        std::vector<float> lbd((size_t)nl * 72);
        std::uniform_int_distribution<int> q(0, 7);
        for (auto& f : lbd) f = 0.05f * (float)q(g);               // coarse levels: many exact ties
        std::vector<uint8_t> rows((size_t)nl * 32), ref((size_t)nl * 32);
        PLSLAM::binaryDescriptorRows(lbd.data(), nl, rows.data());
        plo_lbd_binarise(lbd.data(), nl, ref.data());
        for (size_t i = 0; i < rows.size(); ++i) EXPECT(rows[i] == ref[i]);
    }

    // --- LBA normal equations through the row builder -------------------------------------
    {
        plslam_ctx* ctx = nullptr;
        EXPECT(plslam_ctx_create(0, &ctx) == PLSLAM_OK);
        plslam_cam cam{458.654, 457.296, 367.215, 248.375, 0.11, 752, 480};
        plo_cam ocam{458.654, 457.296, 367.215, 248.375, 0.11, 752, 480};
        std::mt19937 g(5);
        std::uniform_real_distribution<double> U(-1.0, 1.0);
        PLSLAM::LbaProblem p;
        const int nslots = 4, Npt = 60, Nls = 20;
        p.Nkf = 3;  // slot 0 = KF 0 (never optimised, kf_loc = -1), slots 1..3 -> kf_loc 0..2
        for (int k = 0; k < nslots; ++k) {
            double x[6] = {0.02 * U(g), 0.02 * U(g), 0.25 * k, 0.01 * U(g), 0.01 * U(g), 0.01 * U(g)}, T[16];
            plo_expmap_se3(x, T);
            p.poses_T_kf_w.insert(p.poses_T_kf_w.end(), T, T + 16);
        }
        for (int i = 0; i < Npt; ++i) { p.points.push_back(3 * U(g)); p.points.push_back(2 * U(g)); p.points.push_back(8 + 4 * U(g)); }
        for (int i = 0; i < Nls; ++i) for (int e = 0; e < 2; ++e) { p.lines.push_back(3 * U(g)); p.lines.push_back(2 * U(g)); p.lines.push_back(8 + 4 * U(g)); }
        for (int i = 0; i < Npt; ++i)
            for (int s = 0; s < 3; ++s) {
                const int slot = (i + s) % nslots;
                p.pt_obs_list.push_back({i, i, s, slot, slot - 1, 1});
                p.pt_pose_slot.push_back(slot);
                p.pt_obs.push_back(367 + 150 * U(g)); p.pt_obs.push_back(248 + 100 * U(g));
            }
        for (int i = 0; i < Nls; ++i)
            for (int s = 0; s < 2; ++s) {
                const int slot = (i + 2 * s) % nslots;
                p.ls_obs_list.push_back({i, i, s, slot, slot - 1, 1});
                p.ls_pose_slot.push_back(slot);
                const double a = U(g), b = U(g), nn = std::sqrt(a * a + b * b) + 1e-3;
                p.ls_obs.push_back(a / nn); p.ls_obs.push_back(b / nn); p.ls_obs.push_back(-300 * U(g));
            }
        PLSLAM::LbaRowBuilder rb(ctx, cam, 1e-7);
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<double> H, gv;
            double err = 0;
            rb.buildNormalEquations(p, pass == 1, H, gv, err);
            // oracle: same rows + same accumulation
            const int N = 6 * p.Nkf + 3 * Npt + 6 * Nls;
            std::vector<double> Ho((size_t)N * N, 0.0), go((size_t)N, 0.0);
            double erro = 0;
            const int np = (int)p.pt_obs_list.size(), nl = (int)p.ls_obs_list.size();
            std::vector<int32_t> lm(np), kfl(np), lml(nl), kfll(nl);
            for (int o = 0; o < np; ++o) { lm[o] = p.pt_obs_list[o][1]; kfl[o] = p.pt_obs_list[o][4]; }
            for (int o = 0; o < nl; ++o) { lml[o] = p.ls_obs_list[o][1]; kfll[o] = p.ls_obs_list[o][4]; }
            std::vector<double> Jp((size_t)np * 6), Jl((size_t)np * 3), r(np), w(np);
            plo_lba_point_rows(&ocam, 1e-7, p.poses_T_kf_w.data(), p.points.data(), p.pt_obs.data(), lm.data(), p.pt_pose_slot.data(), np, Jp.data(), Jl.data(), r.data(), w.data());
            plo_lba_accumulate_points(p.Nkf, Npt, Nls, lm.data(), kfl.data(), np, Jp.data(), Jl.data(), r.data(), w.data(), Ho.data(), go.data(), &erro);
            std::vector<double> Jp2((size_t)nl * 6), Jl2((size_t)nl * 6), r2(nl), w2(nl);
            plo_lba_line_rows(&ocam, 1e-7, pass, p.poses_T_kf_w.data(), p.lines.data(), p.ls_obs.data(), lml.data(), p.ls_pose_slot.data(), nl, Jp2.data(), Jl2.data(), r2.data(), w2.data());
            plo_lba_accumulate_lines(p.Nkf, Npt, Nls, lml.data(), kfll.data(), nl, Jp2.data(), Jl2.data(), r2.data(), w2.data(), Ho.data(), go.data(), &erro);
            double worst = 0;
            for (size_t i = 0; i < H.size(); ++i) worst = std::fmax(worst, std::fabs(H[i] - Ho[i]) / std::fmax(1e-300, std::fabs(Ho[i])));
            for (size_t i = 0; i < gv.size(); ++i) worst = std::fmax(worst, std::fabs(gv[i] - go[i]) / std::fmax(1e-300, std::fabs(go[i])));
            EXPECT(worst <= 1e-6);            // the contract (north_star)
            EXPECT(std::fabs(err - erro) <= 1e-6 * std::fabs(erro));
            std::printf("LBA normal equations pass %d: N=%d max rel diff vs oracle %.3g\n", pass, N, worst);
            // block form (device assembly) == the dense accumulation, entry for entry
            PLSLAM::LbaRowBuilder::BlockNormalEquations B;
            rb.buildBlockNormalEquations(p, pass == 1, B);
            int bad = 0;
            double pose_scale = 0, pose_diff = 0;   // keyframe blocks: two-level sum, equal to rounding
            for (int i = 6 * p.Nkf; i < N; ++i) bad += B.g[i] != gv[i];
            for (int i = 0; i < 6 * p.Nkf; ++i) { pose_scale = std::fmax(pose_scale, std::fabs(gv[i])); pose_diff = std::fmax(pose_diff, std::fabs(B.g[i] - gv[i])); }
            EXPECT(pose_diff <= 1e-13 * pose_scale);
            pose_scale = pose_diff = 0;
            for (int k = 0; k < p.Nkf; ++k)
                for (int a = 0; a < 6; ++a)
                    for (int b = 0; b < 6; ++b) {
                        const double ref = H[(size_t)(6 * k + a) * N + 6 * k + b];
                        pose_scale = std::fmax(pose_scale, std::fabs(ref));
                        pose_diff = std::fmax(pose_diff, std::fabs(B.H_pose[(size_t)k * 36 + a * 6 + b] - ref));
                    }
            EXPECT(pose_diff <= 1e-13 * pose_scale);
            for (int l = 0; l < Npt; ++l)
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        const int j = 6 * p.Nkf + 3 * l;
                        bad += B.H_pt[(size_t)l * 9 + a * 3 + b] != H[(size_t)(j + a) * N + j + b];
                    }
            for (int o = 0; o < np; ++o) {
                const int kf = p.pt_obs_list[o][4];
                if (kf < 0) continue;
                const int j = 6 * p.Nkf + 3 * p.pt_obs_list[o][1], i0 = 6 * kf;
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 6; ++b) bad += B.W_pt[(size_t)o * 18 + a * 6 + b] != H[(size_t)(j + a) * N + i0 + b];
            }
            EXPECT(bad == 0);
            EXPECT(std::fabs(B.err - err) <= 1e-12 * std::fabs(err));
            if (pass == 0) {
                // --- the LM step on a device-resident plan (PLSLAM::LbaPlanSolver: Schur step of the C ABI + a dense LDL^T of the
                // reduced camera system here) against a dense solve of the damped system accumulated above (:1552-1556)
                PLSLAM::LbaPlanSolver solver(ctx, cam, 1e-7, p);
                const double e_plan = solver.iterate(p, false);
                EXPECT(std::fabs(e_plan - err) <= 1e-12 * std::fabs(err));
                const double hmax = solver.diagMax();
                double hmax_ref = 0;
                for (int i = 0; i < N; ++i) hmax_ref = std::fmax(hmax_ref, std::fabs(H[(size_t)i * N + i]));
                EXPECT(std::fabs(hmax - hmax_ref) <= 1e-12 * hmax_ref);
                const double lambda = 1e-3;
                std::vector<double> dp, dxp, dxl;
                int32_t nsing = -1;
                solver.solveStep(lambda, dp, /*apply=*/false, &dxp, &dxl, &nsing);
                EXPECT(nsing == 0);
                std::vector<double> Hd(H), x(gv);
                for (int i = 0; i < N; ++i) Hd[(size_t)i * N + i] += lambda * Hd[(size_t)i * N + i];
                // Gaussian elimination with partial pivoting on the dense damped system
                for (int c = 0; c < N; ++c) {
                    int piv = c;
                    for (int r2 = c + 1; r2 < N; ++r2) if (std::fabs(Hd[(size_t)r2 * N + c]) > std::fabs(Hd[(size_t)piv * N + c])) piv = r2;
                    if (piv != c) { for (int k = 0; k < N; ++k) std::swap(Hd[(size_t)c * N + k], Hd[(size_t)piv * N + k]); std::swap(x[c], x[piv]); }
                    for (int r2 = c + 1; r2 < N; ++r2) {
                        const double f = Hd[(size_t)r2 * N + c] / Hd[(size_t)c * N + c];
                        if (f == 0.0) continue;
                        for (int k = c; k < N; ++k) Hd[(size_t)r2 * N + k] -= f * Hd[(size_t)c * N + k];
                        x[r2] -= f * x[c];
                    }
                }
                for (int r2 = N - 1; r2 >= 0; --r2) {
                    for (int k = r2 + 1; k < N; ++k) x[r2] -= Hd[(size_t)r2 * N + k] * x[k];
                    x[r2] /= Hd[(size_t)r2 * N + r2];
                }
                double scale = 0, diff = 0;
                for (int i = 0; i < N; ++i) scale = std::fmax(scale, std::fabs(x[i]));
                for (int i = 0; i < 6 * p.Nkf; ++i) diff = std::fmax(diff, std::fabs(dp[i] - x[i]));
                for (int i = 0; i < 3 * Npt; ++i) diff = std::fmax(diff, std::fabs(dxp[i] - x[6 * p.Nkf + i]));
                for (int i = 0; i < 6 * Nls; ++i) diff = std::fmax(diff, std::fabs(dxl[i] - x[6 * p.Nkf + 3 * Npt + i]));
                EXPECT(diff <= 1e-7 * scale);
                std::printf("LBA Schur step: reduced system %d x %d, max |step - dense solve| = %.3g of max |step|\n", 6 * p.Nkf, 6 * p.Nkf, diff / scale);
                // the update in place + the poses: the resident iteration equals the uploaded one on the same state
                solver.solveStep(lambda, dp, /*apply=*/true);
                PLSLAM::LbaProblem p2 = p;
                for (size_t i = 0; i < p2.points.size(); ++i) p2.points[i] += dxp[i];
                for (size_t i = 0; i < p2.lines.size(); ++i) p2.lines[i] += dxl[i];
                solver.setPoses(p2.poses_T_kf_w);                       // (unchanged poses: the call itself is what is exercised)
                const double e_res = solver.iterateResident(false);
                PLSLAM::LbaPlanSolver fresh(ctx, cam, 1e-7, p2);
                EXPECT(e_res == fresh.iterate(p2, false));
                // the state kept in the plan's page-locked images: the same error, the gradient where the caller reads it
                const plslam_lba_host_state hs = fresh.hostState();
                EXPECT(hs.n == (int64_t)N && hs.npt == Npt && hs.nls == Nls);
                std::copy(p.poses_T_kf_w.begin(), p.poses_T_kf_w.end(), hs.T_kf_w);
                std::copy(p.points.begin(), p.points.end(), hs.Xw);
                std::copy(p.lines.begin(), p.lines.end(), hs.Lw);
                EXPECT(fresh.iterateInPlace(false) == e_plan);
                double gscale = 0, gdiff = 0;        // (bit-identity with the staged call: tests/test_gpu_lba.py)
                for (int i = 0; i < N; ++i) { gscale = std::fmax(gscale, std::fabs(B.g[i])); gdiff = std::fmax(gdiff, std::fabs(hs.g[i] - B.g[i])); }
                EXPECT(gdiff <= 1e-12 * gscale);
            }
        }
        plslam_ctx_destroy(ctx);
    }
    {   // --- the newer C entry points through a C++ translation unit: struct layouts + results vs the oracle ---
        plslam_ctx* ctx = nullptr;
        EXPECT(plslam_ctx_create(0, &ctx) == PLSLAM_OK);
        std::mt19937 g(99);
        std::uniform_real_distribution<double> U(0.0, 1.0);
        // stereo point gate
        const int n = 500;
        std::vector<float> kl(2 * n), kr(2 * n);
        std::vector<int32_t> m12(n), s12(n), rs12(n);
        std::vector<double> disp(n), rdisp(n);
        for (int i = 0; i < n; ++i) {
            kl[2 * i] = (float)(752 * U(g)); kl[2 * i + 1] = (float)(480 * U(g));
            kr[2 * i] = kl[2 * i] - (float)(40 * U(g) - 2); kr[2 * i + 1] = kl[2 * i + 1] + (float)(3 * U(g) - 1.5);
            m12[i] = U(g) < 0.8 ? i : -1;
        }
        int32_t ns = 0;
        EXPECT(plslam_stereo_point_gate(ctx, m12.data(), n, kl.data(), kr.data(), n, 1.0, 1.0, s12.data(), disp.data(), &ns) == PLSLAM_OK);
        EXPECT(ns == plo_stereo_point_gate(m12.data(), n, kl.data(), kr.data(), n, 1.0, 1.0, rs12.data(), rdisp.data()));
        for (int i = 0; i < n; ++i) EXPECT(s12[i] == rs12[i] && disp[i] == rdisp[i]);
        // LBD float descriptor: plslam_lbd_line and plo_lbd_line share their layout
        static_assert(sizeof(plslam_lbd_line) == sizeof(plo_lbd_line), "line record layout");
        const int W = 160, H = 120, nl = 20;
        std::vector<int16_t> gx((size_t)W * H), gy((size_t)W * H);
        for (auto& v : gx) v = (int16_t)(int)(600 * U(g) - 300);
        for (auto& v : gy) v = (int16_t)(int)(600 * U(g) - 300);
        std::vector<plslam_lbd_line> lines(nl);
        for (auto& L : lines) {
            L.sx = (float)(20 + 100 * U(g)); L.sy = (float)(20 + 70 * U(g));
            const double a = 6.28 * U(g), len = 10 + 50 * U(g);
            L.ex = L.sx + (float)(len * std::cos(a)); L.ey = L.sy + (float)(len * std::sin(a));
            L.num_pixels = (int32_t)len;
            L.direction = (float)std::atan2(L.ey - L.sy, L.ex - L.sx);
        }
        std::vector<float> lbd((size_t)nl * 72), rlbd((size_t)nl * 72);
        EXPECT(plslam_lbd_compute(ctx, gx.data(), gy.data(), W, H, lines.data(), nl, 7, lbd.data()) == PLSLAM_OK);
        plo_lbd_compute(gx.data(), gy.data(), W, H, reinterpret_cast<const plo_lbd_line*>(lines.data()), nl, 7, rlbd.data());
        EXPECT(std::memcmp(lbd.data(), rlbd.data(), lbd.size() * 4) == 0);
        // pose-only GN system
        plslam_cam K{458.654, 457.296, 367.215, 248.375, 0.11, 752, 480};
        plo_cam oK{458.654, 457.296, 367.215, 248.375, 0.11, 752, 480};
        const int np = 120, nls = 30;
        std::vector<double> P(3 * np), po(2 * np), S(6 * nls), lo(3 * nls);
        std::vector<uint8_t> pi(np), li(nls);
        for (int i = 0; i < np; ++i) {
            P[3 * i] = 4 * U(g) - 2; P[3 * i + 1] = 3 * U(g) - 1.5; P[3 * i + 2] = 3 + 10 * U(g);
            po[2 * i] = K.cx + K.fx * P[3 * i] / P[3 * i + 2] + 2 * U(g); po[2 * i + 1] = K.cy + K.fy * P[3 * i + 1] / P[3 * i + 2] - 2 * U(g);
            pi[i] = U(g) < 0.9;
        }
        for (int i = 0; i < nls; ++i) {
            for (int k = 0; k < 2; ++k) { S[6 * i + 3 * k] = 4 * U(g) - 2; S[6 * i + 3 * k + 1] = 3 * U(g) - 1.5; S[6 * i + 3 * k + 2] = 3 + 10 * U(g); }
            lo[3 * i] = 0.6; lo[3 * i + 1] = 0.8; lo[3 * i + 2] = -400 + 20 * U(g);
            li[i] = U(g) < 0.9;
        }
        const double T[16] = {1, 0, 0, 0.01, 0, 1, 0, -0.02, 0, 0, 1, 0.03, 0, 0, 0, 1};
        double Hm[36], gv[6], e, rH[36], rg[6], re;
        int32_t cnt[2], rcnt[2];
        EXPECT(plslam_pose_gn_accumulate(ctx, &K, 1e-7, T, P.data(), po.data(), pi.data(), np, S.data(), lo.data(), li.data(), nls, Hm, gv, &e, cnt) == PLSLAM_OK);
        plo_pose_gn_accumulate(&oK, 1e-7, T, P.data(), po.data(), pi.data(), np, S.data(), lo.data(), li.data(), nls, rH, rg, &re, rcnt);
        EXPECT(cnt[0] == rcnt[0] && cnt[1] == rcnt[1]);
        double hmax = 0;
        for (double v : rH) hmax = std::max(hmax, std::fabs(v));
        for (int k = 0; k < 36; ++k) EXPECT(std::fabs(Hm[k] - rH[k]) <= 1e-9 * hmax);
        EXPECT(std::fabs(e - re) <= 1e-9 * std::fabs(re));
        // keyframe <-> keyframe driver with fast_matching (plslam_fast_matching layout == plo_fast_matching)
        static_assert(sizeof(plslam_fast_matching) == sizeof(plo_fast_matching), "fast-matching record layout");
        const int n1 = 400, n2 = 380;
        std::vector<double> Pp(3 * n1), plc(2 * n2);
        std::vector<uint8_t> dp = rand_desc(g, n1), dc = rand_desc(g, n2);
        for (int i = 0; i < n1; ++i) { Pp[3 * i + 2] = 2 + 20 * U(g); Pp[3 * i] = (752 * U(g) - K.cx) / K.fx * Pp[3 * i + 2]; Pp[3 * i + 1] = (480 * U(g) - K.cy) / K.fy * Pp[3 * i + 2]; }
        for (int j = 0; j < n2; ++j) {
            const int i = j % n1;
            plc[2 * j] = K.cx + K.fx * Pp[3 * i] / Pp[3 * i + 2] + 3 * U(g); plc[2 * j + 1] = K.cy + K.fy * Pp[3 * i + 1] / Pp[3 * i + 2] + 3 * U(g);
            std::vector<uint8_t> src(dp.begin() + (size_t)i * 32, dp.begin() + (size_t)i * 32 + 32), dst;
            noisy(g, src, dst);
            std::copy(dst.begin(), dst.end(), dc.begin() + (size_t)j * 32);
        }
        const double DT[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        plslam_fast_matching fm{1, GRID_COLS, GRID_ROWS, 3, GRID_COLS / 752.0, GRID_ROWS / 480.0, 0.75, 0.75};
        plo_fast_matching ofm{1, GRID_COLS, GRID_ROWS, 3, GRID_COLS / 752.0, GRID_ROWS / 480.0, 0.75, 0.75};
        std::vector<int32_t> km(n1), rkm(n1);
        int32_t kn = 0, used = 0, rused = 0;
        EXPECT(plslam_kf2kf_match_points(ctx, &K, DT, Pp.data(), dp.data(), n1, plc.data(), dc.data(), n2, 0.75f, 1, 20, &fm, km.data(), &kn, &used) == PLSLAM_OK);
        const int rkn = plo_kf2kf_match_points(&oK, DT, Pp.data(), dp.data(), n1, plc.data(), dc.data(), n2, 0.75f, 1, 20, &ofm, rkm.data(), &rused);
        EXPECT(kn == rkn && used == rused && kn > 100);
        for (int i = 0; i < n1; ++i) EXPECT(km[i] == rkm[i]);
        plslam_ctx_destroy(ctx);
    }
    std::printf(g_fail ? "HOST SHIM: %d FAILURES\n" : "HOST SHIM: all checks passed\n", g_fail);
    return g_fail ? 1 : 0;
}
