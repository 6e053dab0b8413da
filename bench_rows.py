"""Secondary records of bench.py for the SURVEY section-8 rows that are not the headline metric: every record is a short run of
one row's entry points on synthetic data of the BASELINE sizes, VERIFIED against the oracle over everything it produced (not
a sample), with the oracle's single-thread time on the same input beside it.  The oracle is imported here as the CHECKER
only (bench.py's contract); nothing of it is on the measured path.

  grid          a3 / f2   StVO::matchGrid (K14): one 1500 x 1500 point call, one 200 x 200 line call, a 1024-pair plan
  drivers       a6-a8     matchMap2KFPoints / Lines, matchKF2KFPoints / Lines: fast_matching and brute force, C3 sizes
  lba_iterate   f1        one LM iteration of the LBA plan with the blocks left on the device, C3 sizes
  lbd           f4        computeLBD (K18) + binary conversion (K11): one frame's 200 lines and 65 536 lines
  median_desc   a13 / f3  representative descriptors of a C3 map
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

_ROOT = os.path.dirname(os.path.abspath(__file__))


def _test_helpers():
    """The scene builders the parity tests use (tests/ travels with the repository)."""
    p = os.path.join(_ROOT, "tests")
    if p not in sys.path:
        sys.path.insert(0, p)
    import test_map2kf as TM                         # noqa: E402
    import test_match_grid_cpu as TG                 # noqa: E402
    return TM, TG


def _pct(ts_us):
    ts = np.asarray(ts_us, dtype=np.float64)
    return {"us_median": float(np.median(ts)), "us_p10": float(np.percentile(ts, 10)), "us_p90": float(np.percentile(ts, 90))}


def _wall(f, reps, warm=3):
    for _ in range(warm):
        f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e6)
    return ts


def _events(torch, stream, f, iters, warm=3):
    with torch.cuda.stream(stream):
        for _ in range(warm):
            f()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            f()
        e1.record(stream)
        stream.synchronize()
    return e0.elapsed_time(e1) / iters           # ms


def grid(ctx, dev, torch, O, stream):
    import plslam_amd
    from plslam_amd import grid as G
    _, TG = _test_helpers()
    W = (3, 3, 3, 3)
    rec = {"workload": "StVO::matchGrid (mapHandler.cpp:271,418,591,706): 64 x 48 grid, matching_f2f_ws = 3, nnr 0.75, mutual"}
    for name, mk, n in (("point_call_1500x1500", TG.point_case, 1500), ("line_call_200x200", TG.line_case, 200)):
        c = mk(11, n, n, G.GRID_COLS, G.GRID_ROWS)
        m, k = ctx.match_grid(window=W, nnr=0.75, mutual=True, **c)
        t0 = time.perf_counter()
        rm, rk = O.match_grid(window=W, nnr=0.75, mutual=True, **c)
        cpu = (time.perf_counter() - t0) * 1e6
        if not (np.array_equal(m, rm) and k == rk):
            raise SystemExit(f"secondary record grid/{name}: table differs from the oracle")
        cen = np.asarray(c["centres"], np.int32).reshape(-1, 2)
        rec[name] = dict(_pct(_wall(lambda: ctx.match_grid(window=W, nnr=0.75, mutual=True, **c), 30)),
                         cpu_oracle_1thread_us=cpu, matches=int(k),
                         candidate_pairs=int(G.pair_count(cen, c["cell_start"], G.GRID_COLS, G.GRID_ROWS, W)),
                         verified="table and count bit-exact vs the oracle")
    # a device-resident plan of 1024 frame pairs = 1024 point + 1024 line problems (8 distinct inputs)
    B = 1024
    keep, probs, outs, ref = [], [], [], []

    def up(a, dt):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        keep.append(t)
        return t
    base = [(TG.point_case(100 + s, 1500, 1500, G.GRID_COLS, G.GRID_ROWS), False) for s in range(4)] + \
           [(TG.line_case(200 + s, 200, 200, G.GRID_COLS, G.GRID_ROWS), True) for s in range(4)]
    ups = []
    for c, lines in base:
        cen = np.asarray(c["centres"], np.int32).reshape(c["d1"].shape[0], -1, 2)
        u = dict(d1=up(c["d1"], np.uint8), d2=up(c["d2"], np.uint8), cen=up(cen, np.int32), cs=up(c["cell_start"], np.int32),
                 it=up(c["cell_items"], np.int32), nc=cen.shape[1],
                 cap=G.store_capacity(cen, c["cell_start"], G.GRID_COLS, G.GRID_ROWS, W), lines=lines)
        if lines:
            u.update(a=up(c["dir1"], np.float64), b=up(c["dir2"], np.float64))
        ups.append(u)
        ref.append(O.match_grid(window=W, nnr=0.75, mutual=True, **c))
    for b in range(B):
        for kind in (0, 4):
            u = ups[kind + b % 4]
            n1, n2 = u["d1"].shape[0], u["d2"].shape[0]
            o, cnt = torch.empty(n1, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev)
            outs.append((kind + b % 4, o, cnt))
            q = dict(d1=u["d1"].data_ptr(), d2=u["d2"].data_ptr(), centres1=u["cen"].data_ptr(), cell_start=u["cs"].data_ptr(),
                     cell_items=u["it"].data_ptr(), n1=n1, n2=n2, n_centres=u["nc"], grid_cols=G.GRID_COLS,
                     grid_rows=G.GRID_ROWS, n_items=u["it"].shape[0], window=W, nnr=0.75, mutual=True,
                     pair_capacity=u["cap"], matches_12=o.data_ptr(), n_matches=cnt.data_ptr())
            if u["lines"]:
                q.update(dir1=u["a"].data_ptr(), dir2=u["b"].data_ptr(), sim_th=0.75)
            probs.append(q)
    plan = plslam_amd.GridPlan(ctx, probs)
    ms = _events(torch, stream, lambda: plan.run(stream.cuda_stream), 10)
    if plan.overflows(stream.cuda_stream) != 0:
        raise SystemExit("secondary record grid/plan: candidate store overflow")
    for case, o, cnt in outs:                                   # EVERY problem of the batch
        if not (np.array_equal(o.cpu().numpy(), ref[case][0]) and int(cnt.item()) == ref[case][1]):
            raise SystemExit("secondary record grid/plan: a table of the batch differs from the oracle")
    rec["plan_1024_frame_pairs"] = {"problems": len(probs), "ms_per_launch": ms, "frame_pairs_per_s": B / ms * 1e3,
                                    "verified": f"all {len(probs)} tables and counts bit-exact vs the oracle"}
    plan.close()
    return rec


def drivers(ctx, O, dev_tensors=None):
    import plslam_amd
    from plslam_amd import synth
    TM, _ = _test_helpers()
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), O.make_cam(**synth.EUROC)
    rec = {"workload": "MapHandler::matchMap2KFPoints / Lines (mapHandler.cpp:532-752) at C3 sizes and matchKF2KFPoints / Lines "
                       "(:234-530) at C2 sizes: one host-pointer call (what the local-mapping thread does per keyframe), with "
                       "fast_matching (matchGrid first, brute-force fall-back) and without"}
    for kind, n_map, n_kf in (("points", 10000, 1500), ("lines", 2000, 200)):
        s = TM.scene(n_map, n_kf, lines=(kind == "lines"), seed=n_map + 1)
        a = (s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"])
        for name, fm in (("fast_matching", TM.fast_cfg()), ("brute_force", TM.fast_cfg(enabled=0))):
            g = lambda: ctx.map2kf_match_fast(kind, cam, *a, 0.9, True, 1.5, 10, fm, kf_seg=s.get("kf_seg"))   # noqa: E731
            got = g()
            t0 = time.perf_counter()
            ref = O.map2kf_match_fast(kind, ocam, *a, 0.9, True, 1.5, 10, fm, kf_seg=s.get("kf_seg"))
            cpu = (time.perf_counter() - t0) * 1e6
            if not (np.array_equal(got[0], ref[0]) and got[1] == ref[1]):
                raise SystemExit(f"secondary record drivers/map2kf_{kind}_{name}: result differs from the oracle")
            rec[f"map2kf_{kind}_{n_map}x{n_kf}_{name}"] = dict(_pct(_wall(g, 20)), cpu_oracle_1thread_us=cpu,
                                                               associations=int(ref[1]),
                                                               verified="association table + inlier count bit-exact vs the oracle")
            if dev_tensors is not None:
                # the same call with the map side resident on the device (plslam_map2kf_match_*_dev)
                import torch
                d_lm, d_md, d_cd = (torch.from_numpy(np.ascontiguousarray(s[k_])).to(dev_tensors) for k_ in ("LM", "med", "cand"))
                gd = lambda: ctx.map2kf_match_dev(kind, cam, s["Twf"], d_lm.data_ptr(), d_md.data_ptr(), d_cd.data_ptr(), n_map,   # noqa: E731
                                                  s["kf_desc"], s["kf_feat"], s["kf_idx"], 0.9, True, 1.5, 10, fm, kf_seg=s.get("kf_seg"))
                gotd = gd()
                if not (np.array_equal(gotd[0], ref[0]) and gotd[1] == ref[1]):
                    raise SystemExit(f"secondary record drivers/map2kf_{kind}_{name} (map on the device): result differs from the oracle")
                rec[f"map2kf_{kind}_{n_map}x{n_kf}_{name}"]["map_on_device_us_median"] = _pct(_wall(gd, 20))["us_median"]
    for kind, n in (("points", 1500), ("lines", 200)):
        s = TM.kf_pair(n, n - 100, lines=(kind == "lines"), seed=n)
        a = (s["DT"], s["X"], s["d_prev"], s["feat"], s["d_curr"])
        for name, fm in (("fast_matching", TM.fast_cfg()), ("brute_force", TM.fast_cfg(enabled=0))):
            g = lambda: ctx.kf2kf_match(kind, cam, *a, 0.75, True, 20, fm)   # noqa: E731
            got = g()
            t0 = time.perf_counter()
            ref = O.kf2kf_match(kind, ocam, *a, 0.75, True, 20, fm)
            cpu = (time.perf_counter() - t0) * 1e6
            if not (np.array_equal(got[0], ref[0]) and got[1] == ref[1]):
                raise SystemExit(f"secondary record drivers/kf2kf_{kind}_{name}: result differs from the oracle")
            rec[f"kf2kf_{kind}_{n}_{name}"] = dict(_pct(_wall(g, 20)), cpu_oracle_1thread_us=cpu, matches=int(ref[1]),
                                                   verified="match table + count bit-exact vs the oracle")
            if dev_tensors is not None:
                # the same call with the keyframes' rows resident on the device (plslam_kf2kf_match_*_dev)
                import torch
                dX, dP, dC = (torch.from_numpy(np.ascontiguousarray(s[k_])).to(dev_tensors) for k_ in ("X", "d_prev", "d_curr"))
                gd = lambda: ctx.kf2kf_match_dev(kind, cam, s["DT"], dX.data_ptr(), dP.data_ptr(), n, s["feat"], dC.data_ptr(),   # noqa: E731
                                                 0.75, True, 20, fm)
                gotd = gd()
                if not (np.array_equal(gotd[0], ref[0]) and gotd[1] == ref[1]):
                    raise SystemExit(f"secondary record drivers/kf2kf_{kind}_{name} (rows on the device): result differs from the oracle")
                rec[f"kf2kf_{kind}_{n}_{name}"]["rows_on_device_us_median"] = _pct(_wall(gd, 20))["us_median"]
    return rec


def lba_iterate(ctx, O):
    import plslam_amd
    from plslam_amd import synth
    lm = synth.local_map()
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), O.make_cam(**synth.EUROC)
    plan = plslam_amd.LbaPlan(ctx, cam, 1e-7, 10, 9, 10000, 2000, lm["pt_lm"], lm["pt_kf"], lm["pt_kf"] - 1, lm["obs_uv"],
                              lm["ls_lm"], lm["ls_kf"], lm["ls_kf"] - 1, lm["l_obs"])
    err, g = plan.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"])
    # the checker: rows of the same pass from the oracle -> the weighted error and the gradient the plan returns
    t0 = time.perf_counter()
    pr = O.lba_point_rows(ocam, 1e-7, lm["T_kf_w"], lm["Xw"], lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
    lr = O.lba_line_rows(ocam, 1e-7, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"])
    cpu_rows_ms = 1e3 * (time.perf_counter() - t0)
    err_ref = float(np.sum(pr[2] * pr[2] * pr[3]) + np.sum(lr[2] * lr[2] * lr[3]))
    prow, lrow = plan.rows()
    for got, ref, what in ((prow[0], pr[0], "J_pose(points)"), (prow[1], pr[1], "J_lm(points)"), (prow[2], pr[2], "r(points)"),
                           (lrow[0], lr[0], "J_pose(lines)"), (lrow[1], lr[1], "J_lm(lines)"), (lrow[3], lr[3], "w(lines)")):
        if not np.allclose(got, ref, rtol=1e-6, atol=0):
            raise SystemExit(f"secondary record lba_iterate: {what} differs from the oracle beyond 1e-6 relative")
    if not np.isclose(err, err_ref, rtol=1e-9):
        raise SystemExit("secondary record lba_iterate: the weighted error differs from the oracle's rows")
    ts = _wall(lambda: plan.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"]), 30)
    # the same iteration with less crossing the host boundary: err only down (a device-side solver reads g where it is), and
    # the state resident too (nothing up: the solver updated it in place) -- three launches and an 8-byte download
    err_only = _pct(_wall(lambda: plan.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"], want_g=False), 30))["us_median"]
    # the same call with the state kept IN the plan's page-locked images (plslam_lba_plan_host_state): no staging copies
    hs = plan.host_state()
    hs["T_kf_w"][:] = np.asarray(lm["T_kf_w"], np.float64).reshape(-1, 16)
    hs["Xw"][:], hs["Lw"][:] = lm["Xw"], lm["Lw"]
    err_h, g_h = plan.iterate_dev(hs["T_kf_w"], hs["Xw"], hs["Lw"], g_out=hs["g"])
    if err_h != err or g_h is not hs["g"] or not np.array_equal(g_h, g):
        raise SystemExit("secondary record lba_iterate: the iteration on the page-locked images differs")
    in_place = _pct(_wall(lambda: plan.iterate_dev(hs["T_kf_w"], hs["Xw"], hs["Lw"], g_out=hs["g"]), 30))["us_median"]
    if plan.iterate_resident() != err:
        raise SystemExit("secondary record lba_iterate: the resident iteration's error differs")
    resident = _pct(_wall(plan.iterate_resident, 30))["us_median"]
    # the Schur step on the resident blocks (round 5): the reduced 54 x 54 camera system down, the pose step up, the landmark
    # steps on the device.  Checked against a float64 numpy Schur complement of the downloaded blocks, the full step against
    # the block equations themselves (every landmark's own row of the damped system).
    nkf, npt, nls, lam = 9, 10000, 2000, 1e-3
    n6 = 6 * nkf
    Bk = plan.blocks()
    S, b, nsing = plan.schur(lam)
    dp = np.linalg.solve(S, b)
    dxp, dxl = plan.backsub(dp)
    pkf, lkf = lm["pt_kf"] - 1, lm["ls_kf"] - 1
    resid_p = np.einsum("jab,jb->ja", Bk["H_pt"] * (1 + lam * np.eye(3))[None], dxp) - Bk["g"][n6:n6 + 3 * npt].reshape(npt, 3)
    resid_l = np.einsum("jab,jb->ja", Bk["H_ls"] * (1 + lam * np.eye(6))[None], dxl) - Bk["g"][n6 + 3 * npt:].reshape(nls, 6)
    resid_c = np.einsum("kab,kb->ka", Bk["H_pose"] * (1 + lam * np.eye(6))[None], dp.reshape(nkf, 6)) - Bk["g"][:n6].reshape(nkf, 6)
    for o in np.nonzero(pkf >= 0)[0]:
        resid_p[lm["pt_lm"][o]] += Bk["W_pt"][o] @ dp[6 * pkf[o]:6 * pkf[o] + 6]
        resid_c[pkf[o]] += Bk["W_pt"][o].T @ dxp[lm["pt_lm"][o]]
    for o in np.nonzero(lkf >= 0)[0]:
        resid_l[lm["ls_lm"][o]] += Bk["W_ls"][o] @ dp[6 * lkf[o]:6 * lkf[o] + 6]
        resid_c[lkf[o]] += Bk["W_ls"][o].T @ dxl[lm["ls_lm"][o]]
    scale = float(np.abs(Bk["g"]).max())
    worst = max(float(np.abs(resid_p).max()), float(np.abs(resid_l).max()), float(np.abs(resid_c).max())) / scale
    if nsing != 0 or not worst < 1e-8:
        raise SystemExit(f"secondary record lba_iterate: the Schur step does not solve the damped block system (residual {worst:.2e} of |g|max, "
                         f"{nsing} singular landmark blocks)")
    schur_us = _pct(_wall(lambda: plan.schur(lam), 30))["us_median"]
    back_us = _pct(_wall(lambda: plan.backsub(dp, apply=False, want=False), 30))["us_median"]

    def lm_iteration():
        plan.iterate_resident()
        S_, b_, _ = plan.schur(lam)
        plan.backsub(np.linalg.solve(S_, b_), apply=False, want=False)
    whole_us = _pct(_wall(lm_iteration, 30))["us_median"]

    def lm_iteration_two_calls():                           # round 6: plslam_lba_plan_iterate_schur + plslam_lba_plan_apply_step
        _, S_, b_, _n = plan.iterate_schur(lam)
        plan.apply_step(np.linalg.solve(S_, b_), None, apply=False)
    e2, S2, b2, _n2 = plan.iterate_schur(lam)
    if not (np.array_equal(S2, S) and np.array_equal(b2, b)):
        raise SystemExit("secondary record lba_iterate: plslam_lba_plan_iterate_schur does not return plslam_lba_plan_schur's system")
    ss = plan.apply_step(dp, None, apply=False)
    want_ss = float((dxp ** 2).sum() + (dxl ** 2).sum())
    if not abs(ss - want_ss) <= 1e-12 * want_ss:
        raise SystemExit(f"secondary record lba_iterate: plslam_lba_plan_apply_step's sum of squares {ss!r} != {want_ss!r}")
    two_us = _pct(_wall(lm_iteration_two_calls, 30))["us_median"]
    solve_us = _pct(_wall(lambda: np.linalg.solve(S, b), 30))["us_median"]
    plan.close()
    return dict(_pct(ts), err_only_us_median=err_only, state_resident_us_median=resident, state_in_page_locked_images_us_median=in_place,
                schur_step={"schur_us_median": schur_us, "backsub_us_median": back_us, "lm_iteration_blocks_resident_us_median": whole_us,
                            "lm_iteration_two_calls_us_median": two_us, "host_numpy_solve_us_median": solve_us,
                            "two_calls": "plslam_lba_plan_iterate_schur (H, g, err + the Schur step, one synchronisation) + numpy's solve of S + "
                                         "plslam_lba_plan_apply_step (back-substitution + sum DX^2, one synchronisation): the iteration as "
                                         "LbaPlanSolver::optimize runs it; host_numpy_solve = the share of numpy's 54 x 54 solve in both figures",
                            "reduced_system": f"{n6} x {n6}", "lambda": lam, "residual_of_the_damped_system_over_gmax": worst,
                            "what": "plslam_lba_plan_schur (landmark inverses, reduced system S, b: 29 kB down) / plslam_lba_plan_backsub "
                                    "(pose step up, landmark steps on the device) / one whole LM iteration with the state and the blocks "
                                    "resident: iterate_resident + schur + the host's dense solve of S + backsub (mapHandler.cpp:1544-1575)",
                            "verified": "the step (dp, dX) satisfies every block row of the damped normal equations assembled from the "
                                        "downloaded blocks to 1e-8 of max |g|"},
                workload="one Levenberg-Marquardt iteration of MapHandler::levMarquardtOptimizationLBA at C3 sizes "
                         "(mapHandler.cpp:1358-1540 rows + :1410-1429, :1519-1538 block assembly; N = 42 054): poses and landmarks "
                         "uploaded (0.34 MB, one copy), three launches (rows + cross blocks + error partials | landmark blocks + "
                         "keyframe chunk partials | keyframe blocks + error), the error and g downloaded (one copy); the blocks stay "
                         "on the device.  err_only: g stays too; state_resident: nothing uploaded either "
                         "(plslam_lba_plan_iterate_resident); state_in_page_locked_images: the caller keeps T / Xw / Lw and reads g in "
                         "the plan's own page-locked images (plslam_lba_plan_host_state): the same two copies over PCIe, none on the host", rows=60000, cpu_oracle_rows_only_1thread_ms=cpu_rows_ms,
                verified="all 60 000 rows within 1e-6 relative of the oracle, weighted error within 1e-9")


def lbd(ctx, dev, torch, O, stream):
    from plslam_amd import synth
    r = np.random.Generator(np.random.PCG64(4))
    W, H = 752, 480
    dx, dy = synth.gradient_images(r, W, H)
    tx, ty = torch.from_numpy(dx).to(dev), torch.from_numpy(dy).to(dev)
    rec = {"workload": "BinaryDescriptor::computeLBD (binary_descriptor_custom.cpp:1026-1372) + binary conversion (:653-668) on a "
                       "752 x 480 octave, device-resident"}
    for n in (200, 65536):
        lines = synth.lbd_lines(r, n, W, H, min_len=20, max_len=200, dtype=O.LBD_LINE_DTYPE)
        f = torch.empty((n, 72), dtype=torch.float32, device=dev)
        codes = torch.empty((n, 32), dtype=torch.uint8, device=dev)
        reps = 20 if n <= 1000 else 4
        ms_f = _events(torch, stream, lambda: ctx.lbd_compute_dev(tx.data_ptr(), ty.data_ptr(), W, H, lines, f.data_ptr(), 7,
                                                                 stream.cuda_stream), reps)
        ms_b = _events(torch, stream, lambda: ctx.lbd_binarise_dev(f.data_ptr(), n, codes.data_ptr(), stream.cuda_stream), reps)
        m = min(n, 4096)                                                           # the oracle leg is bounded; the binary codes
        t0 = time.perf_counter()                                                   # are checked for every line below
        ref = O.lbd_compute(dx, dy, lines[:m])
        cpu = (time.perf_counter() - t0) / m
        fh = f.cpu().numpy()
        if not np.array_equal(fh[:m].view(np.uint32), ref.view(np.uint32)):
            raise SystemExit("secondary record lbd: float descriptors differ from the oracle")
        if not np.array_equal(codes.cpu().numpy(), O.lbd_binarise(fh)):
            raise SystemExit("secondary record lbd: binary rows differ from the oracle")
        rec[f"lines_{n}"] = {"compute_us": 1e3 * ms_f, "binarise_us": 1e3 * ms_b, "lines_per_s": n / ((ms_f + ms_b) * 1e-3),
                             "binarise_bytes": n * 320,
                             "binarise_note": "the float rows were written by the kernel just before (21 MB at 65 536 lines, 64 kB at "
                                              "200): cache-resident -- the time is launch latency plus on-chip traffic, no HBM rate "
                                              "is derived from it",
                             "cpu_oracle_1thread_lines_per_s": 1.0 / cpu,
                             "verified": f"float descriptors of {m} lines word for word, binary rows of all {n} lines vs the oracle"}
    return rec


def median_desc(ctx, dev, torch, O, stream):
    from plslam_amd import synth
    rr = np.random.Generator(np.random.PCG64(2))
    n_lm, n_obs = 12000, 5
    lists = synth.random_desc(rr, n_lm * n_obs)
    off = (np.arange(n_lm + 1) * n_obs).astype(np.int32)
    dl, do = torch.from_numpy(lists).to(dev), torch.from_numpy(off).to(dev)
    di = torch.empty(n_lm, dtype=torch.int32, device=dev)
    dm = torch.empty((n_lm, 32), dtype=torch.uint8, device=dev)
    ms = _events(torch, stream, lambda: ctx.median_desc_batched_dev(dl.data_ptr(), do.data_ptr(), n_lm, n_lm * n_obs,
                                                                   di.data_ptr(), dm.data_ptr(), stream.cuda_stream), 30)
    t0 = time.perf_counter()
    ri, rm = O.median_desc_batched(lists, off)
    cpu_ms = 1e3 * (time.perf_counter() - t0)
    if not (np.array_equal(di.cpu().numpy(), ri) and np.array_equal(dm.cpu().numpy(), rm)):
        raise SystemExit("secondary record median_desc: result differs from the oracle")
    return {"workload": "MapPoint / MapLine::updateAverageDescDir (mapFeatures.cpp:51-93, :121-163) for a C3 map: 10 000 points + "
                        "2 000 lines, 5 observations each", "landmarks": n_lm, "gpu_us": 1e3 * ms,
            "landmarks_per_s": n_lm / (ms * 1e-3), "cpu_oracle_1thread_ms": cpu_ms,
            "verified": "index and descriptor of all 12 000 landmarks bit-exact vs the oracle"}
