"""The stereo L<->R gates of StVO::StereoFrame (stvo-pl stereoFrame.cpp matchStereoPoints / matchStereoLines, [RECALL];
SURVEY 8 a4; thresholds = the reference's config keys config/config/config_kitti.yaml:25-36): oracle semantics on CPU,
HIP kernels K15/K16 against the oracle on the GPU (tables and disparities bit-exact)."""
import numpy as np
import pytest


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def stereo_points(seed, n_l=1500, n_r=1500, width=752, height=480):
    """Left key points, right ones displaced by a disparity along x with sub-pixel noise in y; a match table that is
    mostly right, sometimes wrong, sometimes empty."""
    r = _rng(seed)
    kp_l = np.stack([r.uniform(0, width, n_l), r.uniform(0, height, n_l)], 1).astype(np.float32)
    src = r.permutation(max(n_l, n_r))[:n_r] % max(n_l, 1)
    disp = r.uniform(-3, 60, n_r)
    kp_r = (kp_l[src] - np.stack([disp, r.normal(0, 0.8, n_r)], 1)).astype(np.float32)
    m12 = np.full(n_l, -1, np.int32)
    m12[src] = np.arange(n_r)
    wrong = r.random(n_l) < 0.1
    m12[wrong] = r.integers(0, max(n_r, 1), int(wrong.sum()))
    m12[r.random(n_l) < 0.2] = -1
    return m12, kp_l, kp_r


def stereo_lines(seed, n_l=200, n_r=200, width=752, height=480):
    r = _rng(seed)
    a = np.stack([r.uniform(0, width, n_l), r.uniform(0, height, n_l)], 1)
    ang = r.uniform(0, np.pi, n_l)
    ln = r.uniform(5, 150, n_l)
    ln[r.random(n_l) < 0.05] = 0.0                                     # zero-length segments
    ang[r.random(n_l) < 0.1] = 0.0                                     # horizontal segments (dy = 0: division by zero)
    seg_l = np.concatenate([a, a + np.stack([np.cos(ang), np.sin(ang)], 1) * ln[:, None]], 1)
    src = r.permutation(max(n_l, n_r))[:n_r] % max(n_l, 1)
    d0, d1 = r.uniform(-2, 50, n_r), r.uniform(0.5, 1.5, n_r)
    seg_r = seg_l[src].copy()
    seg_r[:, 0] -= d0
    seg_r[:, 2] -= d0 * d1
    seg_r += r.normal(0, 0.7, seg_r.shape)
    cut = r.random(n_r) < 0.3                                          # right segment only partly overlapping in y
    seg_r[cut, 2:] = seg_r[cut, :2] + (seg_r[cut, 2:] - seg_r[cut, :2]) * r.uniform(0.1, 0.9, (int(cut.sum()), 1))
    m12 = np.full(n_l, -1, np.int32)
    m12[src] = np.arange(n_r)
    wrong = r.random(n_l) < 0.1
    m12[wrong] = r.integers(0, max(n_r, 1), int(wrong.sum()))
    m12[r.random(n_l) < 0.15] = -1
    return m12, seg_l.astype(np.float32), seg_r.astype(np.float32)


def test_point_gate_semantics(oracle):
    kp_l = np.array([[100.0, 50.0], [100.0, 50.0], [100.0, 50.0], [100.0, 50.0], [10.0, 10.0]], np.float32)
    kp_r = np.array([[90.0, 50.5], [90.0, 51.5], [99.5, 50.0], [99.0, 50.0]], np.float32)
    m12 = np.array([0, 1, 2, 3, -1], np.int32)
    out, disp, n = oracle.stereo_point_gate(m12, kp_l, kp_r, 1.0, 1.0)
    # row 1: |dy| = 1.5 > 1; row 2: disparity 0.5 < 1; row 3: disparity exactly 1 (>=) passes
    assert out.tolist() == [0, -1, -1, 3, -1] and n == 2
    assert disp.tolist() == [10.0, 0.0, 0.0, 1.0, 0.0]
    # max_dist_epip = 0 (config_kitti.yaml:25) keeps only rows with identical y
    assert oracle.stereo_point_gate(m12, kp_l, kp_r, 0.0, 1.0)[0].tolist() == [-1, -1, -1, 3, -1]
    # the arithmetic is float: 0.1f + 0.2f style -- a y difference that is <= 1 only in float
    a = np.array([[5.0, 16777216.0]], np.float32)
    b = np.array([[1.0, 16777215.0]], np.float32)
    assert oracle.stereo_point_gate([0], a, b, 1.0, 1.0)[2] == 1


def test_line_gate_semantics(oracle):
    # a vertical pair shifted by 10 px: overlap 1, disparities (10, 10)
    seg_l = np.array([[100, 10, 100, 60]], np.float32)
    seg_r = np.array([[90, 10, 90, 60]], np.float32)
    out, d, n = oracle.stereo_line_gate([0], seg_l, seg_r, 1.0, 0.1, 0.75, 0.7)
    assert out.tolist() == [0] and n == 1 and d.tolist() == [[10.0, 10.0]]
    # the right segment covers half of the left one's rows: overlap 0.5 < 0.75 -> dropped
    assert oracle.stereo_line_gate([0], seg_l, np.array([[90, 10, 90, 35]], np.float32), 1.0, 0.1, 0.75, 0.7)[2] == 0
    assert oracle.stereo_line_gate([0], seg_l, np.array([[90, 10, 90, 35]], np.float32), 1.0, 0.1, 0.4, 0.7)[2] == 1
    # disparities 10 and 4: ratio 0.4 < 0.7 -> both -1 -> dropped
    assert oracle.stereo_line_gate([0], seg_l, np.array([[90, 10, 96, 60]], np.float32), 1.0, 0.1, 0.75, 0.7)[2] == 0
    # horizontal left segment
    assert oracle.stereo_line_gate([0], np.array([[100, 10, 160, 10.05]], np.float32), seg_r, 1.0, 0.1, 0.75, 0.7)[2] == 0
    # lineSegmentOverlapStereo itself: disjoint, contained, partial
    L = oracle.lib().plo_line_segment_overlap_stereo
    assert L(0, 10, 20, 30, 0.1) == 0.0 and L(0, 10, -5, 15, 0.1) == 10 / 15 and L(0, 10, 5, 20, 0.1) == 1.0
    assert L(3.0, 3.05, 0, 100, 0.1) == 1.0                            # horizontal: the test is skipped


@pytest.mark.gpu
@pytest.mark.parametrize("n_l,n_r", [(1500, 1500), (4000, 3500), (37, 64), (1, 1)])
def test_gpu_point_gate_bit_exact(ctx, oracle, n_l, n_r):
    m12, kp_l, kp_r = stereo_points(n_l + n_r, n_l, n_r)
    for th, md in ((1.0, 1.0), (0.0, 1.0), (2.5, 0.0), (0.5, -5.0)):
        got = ctx.stereo_point_gate(m12, kp_l, kp_r, th, md)
        ref = oracle.stereo_point_gate(m12, kp_l, kp_r, th, md)
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
        assert got[2] == ref[2] == int((ref[0] >= 0).sum())
    assert ref[2] > 0 or n_l < 10


@pytest.mark.gpu
@pytest.mark.parametrize("n_l,n_r", [(200, 200), (600, 640), (19, 7), (1, 1)])
def test_gpu_line_gate_bit_exact(ctx, oracle, n_l, n_r):
    m12, seg_l, seg_r = stereo_lines(n_l * 3 + n_r, n_l, n_r)
    tot = 0
    with np.errstate(all="ignore"):
        for md, hz, ov, ratio in ((1.0, 0.1, 0.75, 0.7), (1.0, 1.0, 0.75, 0.7), (0.0, 0.1, 0.2, 0.3), (-2.0, 0.0, 0.0, 0.0)):
            got = ctx.stereo_line_gate(m12, seg_l, seg_r, md, hz, ov, ratio)
            ref = oracle.stereo_line_gate(m12, seg_l, seg_r, md, hz, ov, ratio)
            np.testing.assert_array_equal(got[0], ref[0])
            np.testing.assert_array_equal(got[1], ref[1])
            assert got[2] == ref[2] == int((ref[0] >= 0).sum())
            tot += ref[2]
    assert tot > 0 or n_l < 10


@pytest.mark.gpu
def test_gpu_stereo_association_end_to_end(ctx, oracle):
    """Descriptors -> StVO::matchGrid with the stereo window {matching_s_ws, 0, 0, 0} -> gate: the frame's stereo
    points, all on the device path, against the oracle chain."""
    from plslam_amd import grid as G
    from plslam_amd import synth
    r = _rng(77)
    n = 1200
    _, kp_l, _ = stereo_points(5, n, n)
    disp = r.uniform(2, 40, n)
    kp_r = (kp_l - np.stack([disp, r.normal(0, 0.3, n)], 1)).astype(np.float32)
    d_l = synth.random_desc(r, n)
    d_r = d_l ^ np.packbits(r.random((n, 256)) < 0.05, axis=1)
    perm = r.permutation(n)
    kp_r, d_r = kp_r[perm], np.ascontiguousarray(d_r[perm])
    sc = np.array([G.GRID_COLS / 752.0, G.GRID_ROWS / 480.0])
    cs, items = G.fill_points(G.to_cells(kp_r.astype(np.float64) * sc))
    c = dict(centres=G.to_cells(kp_l.astype(np.float64) * sc), d1=d_l, cell_start=cs, cell_items=items, cols=G.GRID_COLS,
             rows=G.GRID_ROWS, d2=d_r, window=(10, 0, 0, 0), nnr=0.75, mutual=True)
    m_gpu, m_ref = ctx.match_grid(**c), oracle.match_grid(**c)
    np.testing.assert_array_equal(m_gpu[0], m_ref[0])
    got = ctx.stereo_point_gate(m_gpu[0], kp_l, kp_r, 1.0, 1.0)
    ref = oracle.stereo_point_gate(m_ref[0], kp_l, kp_r, 1.0, 1.0)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    truth = np.argsort(perm)                                           # left i1 <-> right truth[i1]
    kept = got[0] >= 0
    assert kept.sum() > 0.6 * n and (got[0][kept] == truth[kept]).mean() > 0.99


@pytest.mark.gpu
@pytest.mark.parametrize("pairs,n_orb,n_lbd,gates", [
    (64, 1500, 200, "kitti"),                                          # C2 sizes, the shipped thresholds
    (96, 300, 70, dict(max_dist_epip=1.0, min_disp=0.0, line_horiz_th=1.0, stereo_overlap_th=0.2, ls_min_disp_ratio=0.3)),
    (8, 2100, 33, dict(max_dist_epip=2.5, min_disp=-5.0, line_horiz_th=0.0, stereo_overlap_th=0.0, ls_min_disp_ratio=0.0))])
def test_gate_stage_of_the_batch_plan(ctx, oracle, pairs, n_orb, n_lbd, gates):
    """SURVEY 8 a4 in the batched path: ONE plslam_match_plan_run yields, for every pair of the batch, the L<->R match
    tables AND the stereo associations that survive StereoFrame's gates (device-resident tables in, no host round
    trip).  Every pair is compared with the oracle chain match -> gate bit for bit: table, disparities (as raw 64-bit
    words: NaN rows of degenerate lines included) and counts."""
    import torch
    import plslam_amd
    from plslam_amd import frontend, synth
    th = synth.KITTI_GATES if gates == "kitti" else gates
    s = synth.stereo_stream(pairs, n_orb, n_lbd, seed=991 + n_orb)
    geo = synth.stereo_geometry(s, seed=17)
    try:
        ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
        bm = frontend.StereoBatchMatcher(ctx, s, nnr_p=0.75, nnr_l=0.9, mutual=True, geometry=geo, gates=th, n_buffers=2)
    finally:
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)
    for k in range(3):                                                  # both buffers, re-run: idempotent
        bm.run_overlapped(k)
    bm.synchronize_all()
    for b in range(2):
        tab = bm.tables[b].cpu().numpy()
        st, sd, sc = bm.stereo_tabs[b].cpu().numpy(), bm.stereo_disps[b].cpu().numpy(), bm.stereo_cnts[b].cpu().numpy()
        sl = frontend.table_slices(n_orb, n_lbd)
        kept_p = kept_l = 0
        check = range(pairs) if b == 0 else range(0, pairs, 7)
        with np.errstate(all="ignore"):
            for i in check:
                mp, _ = oracle.match(s["orb_l"][i + 1], s["orb_r"][i + 1], 0.75, True)
                ml, _ = oracle.match(s["lbd_l"][i + 1], s["lbd_r"][i + 1], 0.9, True)
                np.testing.assert_array_equal(tab[i, sl["orb_lr"]], mp)
                np.testing.assert_array_equal(tab[i, sl["lbd_lr"]], ml)
                ep, dp, cp = oracle.stereo_point_gate(mp, geo["kp_l"][i + 1], geo["kp_r"][i + 1], th["max_dist_epip"],
                                                      th["min_disp"])
                el, dl, cl = oracle.stereo_line_gate(ml, geo["seg_l"][i + 1], geo["seg_r"][i + 1], th["min_disp"],
                                                     th["line_horiz_th"], th["stereo_overlap_th"], th["ls_min_disp_ratio"])
                np.testing.assert_array_equal(st[i, :n_orb], ep)
                np.testing.assert_array_equal(st[i, n_orb:], el)
                np.testing.assert_array_equal(sd[i, :n_orb].view(np.uint64), dp.view(np.uint64))
                np.testing.assert_array_equal(sd[i, n_orb:].view(np.uint64), dl.reshape(-1).view(np.uint64))
                assert sc[i].tolist() == [cp, cl]
                kept_p += cp
                kept_l += cl
        assert kept_p > 0 and (kept_l > 0 or n_lbd < 40)
    bm.close()


@pytest.mark.gpu
def test_failed_gate_request_leaves_the_plan_usable(ctx, oracle):
    """a plslam_match_plan_add_stereo_gates call that fails (a gate with a misaligned feature table, counters
    that are not one array) must leave the plan as it was -- round 3's version had cleared the host image of the problem
    table first and returned without re-uploading it: the device table kept gate indices, the next run passed no gate table,
    and the finalize kernel read through a null pointer.  And a plan whose stage is REMOVED (ngates = 0) runs without it."""
    import torch
    from plslam_amd import frontend, synth
    pairs, n_orb, n_lbd = 48, 300, 64
    s = synth.stereo_stream(pairs, n_orb, n_lbd, seed=5151)
    geo = synth.stereo_geometry(s, seed=3)
    th = dict(synth.KITTI_GATES)
    for post_fuse in (1, 2):                                  # the separate finalize kernel and the fused stage behind the scan
        ctx.set_option("post_fuse", post_fuse)
        try:
            bm = frontend.StereoBatchMatcher(ctx, s, nnr_p=0.75, nnr_l=0.9, mutual=True, geometry=geo, gates=th)
        finally:
            ctx.set_option("post_fuse", 0)
        bm.run()
        torch.cuda.synchronize()
        ref = [x.clone() for x in (bm.table, bm.stereo, bm.stereo_disp.view(torch.int64), bm.stereo_counts)]
        good = dict(matches_12=bm.table.data_ptr(), f_l=bm.g["kp_l"].data_ptr(), f_r=bm.g["kp_r"].data_ptr(), n_l=n_orb, n_r=n_orb,
                    lines=0, stereo_12=bm.stereo.data_ptr(), disp=bm.stereo_disp.data_ptr(), n_stereo=bm.stereo_counts.data_ptr(), **th)
        bad_align = dict(good, f_l=good["f_l"] + 4)                           # float2 rows must be 8-byte aligned
        bad_count = dict(good, n_stereo=bm.stereo_counts.data_ptr() + 40)     # counters of a request: one contiguous array
        for request in ([good, bad_align], [good, bad_count]):
            with pytest.raises(Exception):
                bm.plan.add_stereo_gates(request)
            for t in (bm.stereo, bm.stereo_counts):
                t.fill_(-7)
            bm.run()
            torch.cuda.synchronize()
            got = (bm.table, bm.stereo, bm.stereo_disp.view(torch.int64), bm.stereo_counts)
            assert all(torch.equal(a, b) for a, b in zip(got, ref)), post_fuse      # the old stage still runs, whole
        bm.plan.add_stereo_gates([])                                              # stage removed
        bm.stereo.fill_(-7)
        bm.run()
        torch.cuda.synchronize()
        assert torch.equal(bm.table, ref[0]) and bool((bm.stereo == -7).all())
        bm.close()


@pytest.mark.gpu
def test_device_pointer_gates(ctx, oracle):
    """plslam_stereo_point_gate_dev / _line_gate_dev: device tables in, device results out, on a caller's stream."""
    import ctypes as C
    import torch
    m12, kp_l, kp_r = stereo_points(3, 1500, 1400)
    ml, seg_l, seg_r = stereo_lines(4, 200, 210)
    dev = torch.device("cuda", ctx.device)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)     # noqa: E731
    d_m, d_a, d_b = t(m12), t(kp_l), t(kp_r)
    out = torch.empty(1500, dtype=torch.int32, device=dev)
    disp = torch.empty(1500, dtype=torch.float64, device=dev)
    cnt = torch.full((2,), 77, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    L = ctx._L
    assert L.plslam_stereo_point_gate_dev(ctx.handle, d_m.data_ptr(), 1500, d_a.data_ptr(), d_b.data_ptr(), 1400, 1.0, 1.0,
                                          out.data_ptr(), disp.data_ptr(), cnt.data_ptr(), st.cuda_stream) == 0
    e_m, e_a, e_b = t(ml), t(seg_l), t(seg_r)
    out_l = torch.empty(200, dtype=torch.int32, device=dev)
    disp_l = torch.empty((200, 2), dtype=torch.float64, device=dev)
    assert L.plslam_stereo_line_gate_dev(ctx.handle, e_m.data_ptr(), 200, e_a.data_ptr(), e_b.data_ptr(), 210, 1.0, 0.1, 0.75,
                                         0.7, out_l.data_ptr(), disp_l.data_ptr(), cnt.data_ptr() + 4, st.cuda_stream) == 0
    st.synchronize()
    rp = oracle.stereo_point_gate(m12, kp_l, kp_r, 1.0, 1.0)
    with np.errstate(all="ignore"):
        rl = oracle.stereo_line_gate(ml, seg_l, seg_r, 1.0, 0.1, 0.75, 0.7)
    np.testing.assert_array_equal(out.cpu().numpy(), rp[0])
    np.testing.assert_array_equal(disp.cpu().numpy().view(np.uint64), rp[1].view(np.uint64))
    np.testing.assert_array_equal(out_l.cpu().numpy(), rl[0])
    np.testing.assert_array_equal(disp_l.cpu().numpy().view(np.uint64), rl[1].view(np.uint64))
    assert cnt.cpu().numpy().tolist() == [rp[2], rl[2]]
    # argument checks: misaligned feature rows, counters that are not one array
    assert L.plslam_stereo_point_gate_dev(ctx.handle, d_m.data_ptr(), 1500, d_a.data_ptr() + 4, d_b.data_ptr(), 1400, 1.0, 1.0,
                                          out.data_ptr(), disp.data_ptr(), None, None) == -1


def golden_gate_cases():
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stereo_gates_golden.npz"))
    for c in range(3):
        for t, th in enumerate(g["point_thresholds"]):
            yield ("points", (g[f"p{c}_m12"], g[f"p{c}_kp_l"], g[f"p{c}_kp_r"]), tuple(th),
                   (g[f"p{c}_t{t}_stereo"], g[f"p{c}_t{t}_disp"], int(g[f"p{c}_t{t}_n"])))
        for t, th in enumerate(g["line_thresholds"]):
            yield ("lines", (g[f"l{c}_m12"], g[f"l{c}_seg_l"], g[f"l{c}_seg_r"]), tuple(th),
                   (g[f"l{c}_t{t}_stereo"], g[f"l{c}_t{t}_disp"], int(g[f"l{c}_t{t}_n"])))


def _same_gate(got, want):
    np.testing.assert_array_equal(got[0], want[0])
    assert np.array_equal(np.asarray(got[1], np.float32).view(np.uint32), np.asarray(want[1], np.float32).view(np.uint32))
    assert got[2] == want[2]


def test_oracle_reproduces_the_committed_gate_golden(oracle):
    with np.errstate(all="ignore"):
        for kind, inp, th, want in golden_gate_cases():
            got = (oracle.stereo_point_gate if kind == "points" else oracle.stereo_line_gate)(*inp, *th)
            _same_gate(got, want)


@pytest.mark.gpu
def test_gpu_committed_gate_golden(ctx):
    """GPU vs the committed fixture tests/golden/stereo_gates_golden.npz (no oracle involved at run time)."""
    for kind, inp, th, want in golden_gate_cases():
        got = (ctx.stereo_point_gate if kind == "points" else ctx.stereo_line_gate)(*inp, *th)
        _same_gate(got, want)
