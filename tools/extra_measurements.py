#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (not the bench metric):
  * PCIe-inclusive pairs/s of the host-pointer entry point (plslam_match_batched) on the C2 batch;
  * C3: map<->frame matching (10 000 x 1500 ORB + 2 000 x 200 LBD, mutual) kernel time;
  * C3: LBA row kernels (50 000 point rows, 10 000 line rows) device-resident: us per pass and GB/s
    against the algorithmic bytes (152 B / 208 B per row), with the CPU oracle timed beside it.
Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import plslam_amd  # noqa: E402
from plslam_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


STREAM = None   # a real (non-NULL) HIP stream: the ABI reads NULL as "the context's own stream"


def ev_time(fn, iters=50, warm=5):
    with torch.cuda.stream(STREAM):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(STREAM)
        for _ in range(iters):
            fn()
        e1.record(STREAM)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters   # ms


def main():
    out = {}
    ctx = plslam_amd.Context(0)
    dev = torch.device("cuda", 0)
    global STREAM
    STREAM = torch.cuda.Stream(device=dev)
    st = STREAM.cuda_stream
    assert st != 0

    # ---- PCIe-inclusive: host-pointer API on 128 C2 pairs --------------------------------------
    B = 128
    s = synth.stereo_stream(B, 1500, 200)
    d1 = np.concatenate([np.concatenate([s["orb_l"][i + 1], s["orb_l"][i]]) for i in range(B)])
    d2 = np.concatenate([np.concatenate([s["orb_r"][i + 1], s["orb_l"][i + 1]]) for i in range(B)])
    off = np.arange(0, (2 * B + 1) * 1500, 1500, dtype=np.int32)
    l1 = np.concatenate([np.concatenate([s["lbd_l"][i + 1], s["lbd_l"][i]]) for i in range(B)])
    l2 = np.concatenate([np.concatenate([s["lbd_r"][i + 1], s["lbd_l"][i + 1]]) for i in range(B)])
    offl = np.arange(0, (2 * B + 1) * 200, 200, dtype=np.int32)
    ctx.match_batched(d1, off, d2, off, 0.75, True)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ctx.match_batched(d1, off, d2, off, 0.75, True)
        ctx.match_batched(l1, offl, l2, offl, 0.75, True)
    dt = (time.perf_counter() - t0) / reps
    out["pcie_inclusive"] = {"pairs_per_s": B / dt, "ms_per_call_batch": 1e3 * dt, "pairs": B,
                            "note": "plslam_match_batched with pageable host buffers: plan build + H2D 13.9 MB + "
                                    "kernels + D2H 1.7 MB per 128 pairs"}

    # ---- C3 matching --------------------------------------------------------------------------
    r = np.random.Generator(np.random.PCG64(31))
    frame_p = synth.random_desc(r, 1500)
    map_p = np.concatenate([synth.noisy_copy(r, frame_p)[0], synth.random_desc(r, 8500)])
    frame_l = synth.random_desc(r, 200)
    map_l = np.concatenate([synth.noisy_copy(r, frame_l)[0], synth.random_desc(r, 1800)])
    t = {k: torch.from_numpy(v).to(dev) for k, v in dict(mp=map_p, fp=frame_p, ml=map_l, fl=frame_l).items()}
    m_p = torch.empty(10000, dtype=torch.int32, device=dev)
    m_l = torch.empty(2000, dtype=torch.int32, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    plan = ctx.plan([(t["mp"].data_ptr(), 10000, t["fp"].data_ptr(), 1500, 0.75, True, m_p.data_ptr(), cnt.data_ptr()),
                     (t["ml"].data_ptr(), 2000, t["fl"].data_ptr(), 200, 0.75, True, m_l.data_ptr(), cnt.data_ptr() + 4)])
    ms = ev_time(lambda: plan.run(st), iters=200, warm=10)
    em, en = O.match(map_p, frame_p, 0.75, True)
    assert np.array_equal(m_p.cpu().numpy(), em)
    t0 = time.perf_counter()
    O.match(map_p, frame_p, 0.75, True, L=O.native_lib())
    O.match(map_l, frame_l, 0.75, True, L=O.native_lib())
    cpu_ms = 1e3 * (time.perf_counter() - t0)
    out["c3_map_to_frame_match"] = {"gpu_ms": ms, "cpu_1thread_ms": cpu_ms, "directed_evals": 2 * (10000 * 1500 + 2000 * 200),
                                    "matches_points": int(cnt[0].item()), "info": plan.info()}
    plan.close()

    # ---- C3 LBA rows --------------------------------------------------------------------------
    lm = synth.local_map()
    cam = plslam_amd.make_cam(**synth.EUROC)
    ocam = O.make_cam(**synth.EUROC)
    g = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in lm.items()}
    npt, nls = lm["pt_lm"].shape[0], lm["ls_lm"].shape[0]
    Jp = torch.empty((npt, 6), dtype=torch.float64, device=dev)
    Jl = torch.empty((npt, 3), dtype=torch.float64, device=dev)
    rr = torch.empty(npt, dtype=torch.float64, device=dev)
    ww = torch.empty(npt, dtype=torch.float64, device=dev)
    Jp2 = torch.empty((nls, 6), dtype=torch.float64, device=dev)
    Jl2 = torch.empty((nls, 6), dtype=torch.float64, device=dev)
    r2 = torch.empty(nls, dtype=torch.float64, device=dev)
    w2 = torch.empty(nls, dtype=torch.float64, device=dev)

    def pts():
        ctx.lba_point_rows_dev(cam, 1e-7, g["T_kf_w"].data_ptr(), g["Xw"].data_ptr(), g["obs_uv"].data_ptr(),
                               g["pt_lm"].data_ptr(), g["pt_kf"].data_ptr(), npt, Jp.data_ptr(), Jl.data_ptr(),
                               rr.data_ptr(), ww.data_ptr(), st)

    def lns():
        ctx.lba_line_rows_dev(cam, 1e-7, False, g["T_kf_w"].data_ptr(), g["Lw"].data_ptr(), g["l_obs"].data_ptr(),
                              g["ls_lm"].data_ptr(), g["ls_kf"].data_ptr(), nls, Jp2.data_ptr(), Jl2.data_ptr(),
                              r2.data_ptr(), w2.data_ptr(), st)
    ms_p = ev_time(pts, iters=300, warm=20)
    ms_l = ev_time(lns, iters=300, warm=20)
    ms_both = ev_time(lambda: (pts(), lns()), iters=300, warm=20)
    e = O.lba_point_rows(ocam, 1e-7, lm["T_kf_w"], lm["Xw"], lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
    assert np.allclose(Jp.cpu().numpy(), e[0], rtol=1e-6, atol=0)
    t0 = time.perf_counter()
    for _ in range(20):
        O.lba_point_rows(ocam, 1e-7, lm["T_kf_w"], lm["Xw"], lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
        O.lba_line_rows(ocam, 1e-7, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"])
    cpu_pass_ms = 1e3 * (time.perf_counter() - t0) / 20
    out["c3_lba_rows"] = {
        "point_rows": npt, "line_rows": nls,
        "point_kernel_us": 1e3 * ms_p, "line_kernel_us": 1e3 * ms_l, "pass_us_back_to_back": 1e3 * ms_both,
        "point_GBps_algorithmic": npt * 152 / (ms_p * 1e-3) / 1e9, "line_GBps_algorithmic": nls * 208 / (ms_l * 1e-3) / 1e9,
        "bytes_per_pass": npt * 152 + nls * 208, "cpu_oracle_1thread_ms_per_pass": cpu_pass_ms,
        "note": "launch-latency bound at this size (9.7 MB/pass); times include the launch gap of back-to-back launches"}
    # a batch of many maps in one launch shows the kernel's streaming rate
    reps = 64
    big = {k: torch.cat([v] * reps) for k, v in g.items() if k in ("obs_uv", "pt_lm", "pt_kf")}
    nbig = npt * reps
    JpB = torch.empty((nbig, 6), dtype=torch.float64, device=dev)
    JlB = torch.empty((nbig, 3), dtype=torch.float64, device=dev)
    rB = torch.empty(nbig, dtype=torch.float64, device=dev)
    wB = torch.empty(nbig, dtype=torch.float64, device=dev)
    ms_big = ev_time(lambda: ctx.lba_point_rows_dev(cam, 1e-7, g["T_kf_w"].data_ptr(), g["Xw"].data_ptr(),
                                                    big["obs_uv"].data_ptr(), big["pt_lm"].data_ptr(),
                                                    big["pt_kf"].data_ptr(), nbig, JpB.data_ptr(), JlB.data_ptr(),
                                                    rB.data_ptr(), wB.data_ptr(), st), iters=50, warm=5)
    out["lba_point_rows_streaming"] = {"rows": nbig, "kernel_us": 1e3 * ms_big,
                                       "GBps_algorithmic": nbig * 152 / (ms_big * 1e-3) / 1e9,
                                       "frac_of_8TBps": nbig * 152 / (ms_big * 1e-3) / 8e12}
    bigl = {k: torch.cat([v] * reps) for k, v in g.items() if k in ("l_obs", "ls_lm", "ls_kf")}
    nbl = nls * reps
    JpL = torch.empty((nbl, 6), dtype=torch.float64, device=dev)
    JlL = torch.empty((nbl, 6), dtype=torch.float64, device=dev)
    rL = torch.empty(nbl, dtype=torch.float64, device=dev)
    wL = torch.empty(nbl, dtype=torch.float64, device=dev)
    ms_bigl = ev_time(lambda: ctx.lba_line_rows_dev(cam, 1e-7, False, g["T_kf_w"].data_ptr(), g["Lw"].data_ptr(),
                                                    bigl["l_obs"].data_ptr(), bigl["ls_lm"].data_ptr(),
                                                    bigl["ls_kf"].data_ptr(), nbl, JpL.data_ptr(), JlL.data_ptr(),
                                                    rL.data_ptr(), wL.data_ptr(), st), iters=50, warm=5)
    out["lba_line_rows_streaming"] = {"rows": nbl, "kernel_us": 1e3 * ms_bigl,
                                      "GBps_algorithmic": nbl * 208 / (ms_bigl * 1e-3) / 1e9,
                                      "frac_of_8TBps": nbl * 208 / (ms_bigl * 1e-3) / 8e12}
    # ---- one LM iteration at C3 through the LBA plan (upload X, rows, block assembly, download) ----
    plan = plslam_amd.LbaPlan(ctx, cam, 1e-7, 10, 9, 10000, 2000, lm["pt_lm"], lm["pt_kf"], lm["pt_kf"] - 1, lm["obs_uv"],
                              lm["ls_lm"], lm["ls_kf"], lm["ls_kf"] - 1, lm["l_obs"])
    plan.iterate(lm["T_kf_w"], lm["Xw"], lm["Lw"])
    t0 = time.perf_counter()
    for _ in range(20):
        plan.iterate(lm["T_kf_w"], lm["Xw"], lm["Lw"])
    out["c3_lba_plan_iteration"] = {"ms_per_iteration_host_to_host": 1e3 * (time.perf_counter() - t0) / 20,
                                    "N": 6 * 9 + 3 * 10000 + 6 * 2000,
                                    "note": "upload poses+landmarks (0.34 MB), K3/K4 rows, K7-K10 block assembly, download "
                                            "g + blocks (11.5 MB) to pageable host memory; the reference's dense H at this "
                                            "size would be 14 GB"}
    # ---- K11: LBD float -> binary rows (288 B read + 32 B written per line) ----
    for tag, nlines in (("frame", 200), ("c2_step", 4096 * 2 * 200), ("stream", 8 << 20)):
        f = torch.rand((nlines, 72), dtype=torch.float32, device=dev)
        codes = torch.empty((nlines, 32), dtype=torch.uint8, device=dev)
        ms_k = ev_time(lambda: ctx.lbd_binarise_dev(f.data_ptr(), nlines, codes.data_ptr(), st), iters=50, warm=5)
        out["lbd_binarise_" + tag] = {"lines": nlines, "kernel_us": 1e3 * ms_k,
                                      "GBps_algorithmic": nlines * 320 / (ms_k * 1e-3) / 1e9,
                                      "frac_of_8TBps": nlines * 320 / (ms_k * 1e-3) / 8e12}
        del f, codes
    fh = synth.lbd_float(np.random.Generator(np.random.PCG64(1)), 1 << 18)
    t0 = time.perf_counter()
    O.lbd_binarise(fh)
    out["lbd_binarise_cpu_oracle_1thread_lines_per_s"] = (1 << 18) / (time.perf_counter() - t0)
    # ---- K12/K13: representative descriptor of every landmark (C3 map: 10 000 x 5 + 2 000 x 5) ----
    rr = np.random.Generator(np.random.PCG64(2))
    for tag, n_lm, n_obs in (("c3_map", 12000, 5), ("big_map", 1 << 20, 8)):
        lists = synth.random_desc(rr, n_lm * n_obs)
        off = (np.arange(n_lm + 1) * n_obs).astype(np.int32)
        dl, do = torch.from_numpy(lists).to(dev), torch.from_numpy(off).to(dev)
        di = torch.empty(n_lm, dtype=torch.int32, device=dev)
        dm = torch.empty((n_lm, 32), dtype=torch.uint8, device=dev)
        ms_m = ev_time(lambda: ctx.median_desc_batched_dev(dl.data_ptr(), do.data_ptr(), n_lm, n_lm * n_obs,
                                                           di.data_ptr(), dm.data_ptr(), st), iters=30, warm=3)
        t0 = time.perf_counter()
        nsub = min(n_lm, 50000)
        O.median_desc_batched(lists[:nsub * n_obs], off[:nsub + 1])
        cpu_ms = 1e3 * (time.perf_counter() - t0) * n_lm / nsub
        out["median_desc_" + tag] = {"landmarks": n_lm, "obs_per_landmark": n_obs, "gpu_us": 1e3 * ms_m,
                                     "landmarks_per_s": n_lm / (ms_m * 1e-3), "cpu_oracle_1thread_ms": cpu_ms,
                                     "GBps_compulsory": (n_lm * n_obs * 32 + n_lm * 36) / (ms_m * 1e-3) / 1e9}
        del dl, do, di, dm
    plan.close()
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
