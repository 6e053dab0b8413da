"""The map <-> keyframe association drivers (MapHandler::matchMap2KFPoints / Lines,
src/mapHandler.cpp:532-752, BF path): oracle restatement sanity on CPU, HIP path vs oracle on GPU."""
import numpy as np
import pytest

from plslam_amd import synth


def scene(n_map=3000, n_kf=900, lines=False, seed=1, frac_unmatched=0.6):
    """A local map seen from one keyframe: landmarks around the frustum (some outside), keyframe
    features = noisy re-observations of a subset (descriptor bit flips, ~1 px noise) + clutter."""
    r = np.random.Generator(np.random.PCG64(seed))
    K = synth.EUROC
    Twf = np.linalg.inv(synth.se3_exp([0.2, -0.1, 0.4, 0.02, -0.03, 0.01]))
    z = r.uniform(-2.0, 30.0, n_map)
    u = r.uniform(-0.2 * K["width"], 1.2 * K["width"], n_map)
    v = r.uniform(-0.2 * K["height"], 1.2 * K["height"], n_map)
    Xc = np.stack([(u - K["cx"]) / K["fx"] * np.abs(z), (v - K["cy"]) / K["fy"] * np.abs(z), z], 1)
    Tfw = np.linalg.inv(Twf)
    Xw = Xc @ Tfw[:3, :3].T + Tfw[:3, 3]
    med = synth.random_desc(r, n_map)
    cand = (r.random(n_map) < 0.9).astype(np.uint8)
    src = r.choice(n_map, size=n_kf, replace=False)
    kf_desc = med[src] ^ np.packbits(r.random((n_kf, 256)) < 0.05, axis=1)
    clutter = r.random(n_kf) < 0.3
    kf_desc[clutter] = synth.random_desc(r, int(clutter.sum()))
    kf_idx = np.where(r.random(n_kf) < frac_unmatched, -1, r.integers(0, 5000, n_kf)).astype(np.int32)
    if not lines:
        with np.errstate(all="ignore"):
            pl = np.stack([K["cx"] + K["fx"] * Xc[src, 0] / Xc[src, 2], K["cy"] + K["fy"] * Xc[src, 1] / Xc[src, 2]], 1)
        pl = np.nan_to_num(pl, posinf=0, neginf=0) + r.normal(0, 0.7, (n_kf, 2))
        return dict(Twf=Twf, LM=Xw, med=med, cand=cand, kf_desc=kf_desc, kf_feat=pl, kf_idx=kf_idx)
    Ew = Xw + r.uniform(-0.5, 0.5, (n_map, 3))
    Lw = np.concatenate([Xw, Ew], 1)
    Ec = Ew @ Twf[:3, :3].T + Twf[:3, 3]
    with np.errstate(all="ignore"):
        p = np.stack([K["cx"] + K["fx"] * Xc[src, 0] / Xc[src, 2], K["cy"] + K["fy"] * Xc[src, 1] / Xc[src, 2], np.ones(n_kf)], 1)
        q = np.stack([K["cx"] + K["fx"] * Ec[src, 0] / Ec[src, 2], K["cy"] + K["fy"] * Ec[src, 1] / Ec[src, 2], np.ones(n_kf)], 1)
    le = np.nan_to_num(np.cross(p, q))
    nrm = np.sqrt(le[:, 0] ** 2 + le[:, 1] ** 2)
    le = le / np.where(nrm > 0, nrm, 1.0)[:, None]
    le[:, 2] += r.normal(0, 0.7, n_kf)
    seg = np.nan_to_num(np.concatenate([p[:, :2], q[:, :2]], 1), posinf=0, neginf=0) + r.normal(0, 0.7, (n_kf, 4))
    return dict(Twf=Twf, LM=Lw, med=med, cand=cand, kf_desc=kf_desc, kf_feat=le, kf_idx=kf_idx, kf_seg=seg)


def test_oracle_driver_semantics(oracle):
    cam = oracle.make_cam(**synth.EUROC)
    s = scene(800, 300)
    out, n = oracle.map2kf_match("points", cam, s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"],
                                 s["kf_idx"], 0.9, True, 1.0, 10)
    ok = out >= 0
    assert n == ok.sum() and n > 20
    assert (s["cand"][ok] == 1).all()                          # only candidate landmarks are associated
    assert (s["kf_idx"][out[ok]] == -1).all()                  # ... with still-unmatched features (:565)
    vis = oracle.map_point_visible(cam, s["Twf"], s["LM"]).astype(bool)
    assert vis[ok].all()                                       # ... that project inside the image (:551)
    assert len(set(out[ok])) == n                              # mutual => one-to-one
    # nothing is matched when the candidate list is not larger than min_matches (:594-596)
    nq = int((s["cand"].astype(bool) & vis).sum())
    out2, n2 = oracle.map2kf_match("points", cam, s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"],
                                   s["kf_idx"], 0.9, True, 1.0, nq)
    assert n2 == 0 and (out2 == -1).all()
    out3, n3 = oracle.map2kf_match("points", cam, s["Twf"], s["LM"], s["med"], np.zeros_like(s["cand"]), s["kf_desc"],
                                   s["kf_feat"], s["kf_idx"], 0.9, True, 1.0, 10)
    assert n3 == 0
    # a tighter gate only removes associations
    out4, n4 = oracle.map2kf_match("points", cam, s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"],
                                   s["kf_idx"], 0.9, True, 0.3, 10)
    assert 0 < n4 < n and ((out4 == out) | (out4 == -1)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n_map,n_kf", [("points", 10000, 1500), ("lines", 2000, 200), ("points", 300, 40),
                                             ("lines", 50, 10)])
def test_gpu_driver_bit_exact(ctx, oracle, kind, n_map, n_kf):
    """BASELINE config 3 sizes: 10k map points vs a 1500-feature frame, 2k map lines vs 200."""
    import plslam_amd
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), oracle.make_cam(**synth.EUROC)
    s = scene(n_map, n_kf, lines=(kind == "lines"), seed=n_map)
    for nnr, mutual, th, mm in ((0.9, True, 0.5, 0), (0.75, True, 1.0, 10), (0.9, False, 2.0, 6), (0.9, True, 0.5, 1)):
        exp, en = oracle.map2kf_match(kind, ocam, s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"],
                                      s["kf_idx"], nnr, mutual, th, mm)
        got, gn = ctx.map2kf_match(kind, cam, s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"],
                                   s["kf_idx"], nnr, mutual, th, mm)
        assert np.array_equal(got, exp) and gn == en, (kind, nnr, mutual, th)
    assert en > 0 or n_map < 100


@pytest.mark.gpu
def test_gpu_driver_degenerate(ctx, oracle):
    import plslam_amd
    cam = plslam_amd.make_cam(**synth.EUROC)
    s = scene(200, 50)
    none = np.full(50, 7, np.int32)                            # every feature already matched -> T empty
    got, gn = ctx.map2kf_match("points", cam, s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], none,
                               0.9, True, 1.0, 10)
    assert gn == 0 and (got == -1).all()
    got, gn = ctx.map2kf_match("points", cam, s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"],
                               s["kf_idx"], 0.9, True, 1.0, 10 ** 6)
    assert gn == 0 and (got == -1).all()
    got, gn = ctx.map2kf_match("points", cam, s["Twf"], np.zeros((0, 3)), np.zeros((0, 32), np.uint8),
                               np.zeros(0, np.uint8), s["kf_desc"], s["kf_feat"], s["kf_idx"], 0.9, True, 1.0, 10)
    assert gn == 0 and got.shape == (0,)


def fast_cfg(ws=3, nnr_grid=0.75, enabled=1):
    """The shipped configuration: 64 x 48 grid over the 752 x 480 image, matching_f2f_ws = 3
    (config/config/config_euroc.yaml / config_kitti.yaml:59)."""
    K = synth.EUROC
    return dict(enabled=enabled, grid_cols=64, grid_rows=48, ws=ws, inv_width=64 / K["width"], inv_height=48 / K["height"],
                nnr_grid=nnr_grid, line_sim_th=0.75)


def test_oracle_fast_driver_semantics(oracle):
    """fast_matching: matchGrid first; StVO::match only replaces it when it found fewer than min_matches."""
    cam = oracle.make_cam(**synth.EUROC)
    s = scene(1500, 600)
    a = (s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"])
    out, n, used = oracle.map2kf_match_fast("points", cam, *a, 0.9, True, 1.5, 10, fast_cfg())
    assert n > 20 and used == 0 and n == (out >= 0).sum()
    # a grid result below min_matches is replaced by brute force: identical to the plain driver then
    vis = oracle.map_point_visible(cam, s["Twf"], s["LM"]).astype(bool)
    nq = int((s["cand"].astype(bool) & vis).sum())
    assert nq > n + 60                                        # room for a min_matches between the two
    # a grid result below min_matches: StVO::match runs ON THE SAME VECTOR -- rows it accepts are overwritten, rows it
    # rejects keep matchGrid's entry (stvo-pl's resize(), see plo_match_prior), all of them pass the epipolar gate after
    out2, n2, used2 = oracle.map2kf_match_fast("points", cam, *a, 0.9, True, 1.5, n + 50, fast_cfg())
    ref, nref = oracle.map2kf_match("points", cam, *a, 0.9, True, 1.5, n + 50)
    assert used2 == 1 and n2 == (out2 >= 0).sum() >= nref
    both = (ref >= 0)
    assert (out2[both] == ref[both]).all()                    # whatever brute force associates stands
    extra = (out2 >= 0) & ~both
    assert (out2[extra] == out[extra]).all()                  # the rest are matchGrid's entries that survived
    # ... but not when the candidate list itself is not larger than min_matches (:594)
    out2b, n2b, used2b = oracle.map2kf_match_fast("points", cam, *a, 0.9, True, 1.5, nq, fast_cfg())
    assert used2b == 0 and n2b == n and (out2b == out).all()
    # disabled == the plain driver
    out3, n3, used3 = oracle.map2kf_match_fast("points", cam, *a, 0.9, True, 1.5, 10, fast_cfg(enabled=0))
    ref3, nref3 = oracle.map2kf_match("points", cam, *a, 0.9, True, 1.5, 10)
    assert used3 == 1 and n3 == nref3 and (out3 == ref3).all()
    # min_matches = 0: nothing can be "fewer than 0", so the plain driver matches nothing (:594-596) ...
    assert oracle.map2kf_match("points", cam, *a, 0.9, True, 1.5, 0)[1] == 0
    # ... while the grid result stands on its own
    assert oracle.map2kf_match_fast("points", cam, *a, 0.9, True, 1.5, 0, fast_cfg())[1] == n
    # associations of the windowed matcher lie within the window (3 cells of 11.75 x 10 px) + the gate
    sl = scene(400, 150, lines=True)
    outl, nl, usedl = oracle.map2kf_match_fast("lines", cam, sl["Twf"], sl["LM"], sl["med"], sl["cand"], sl["kf_desc"],
                                               sl["kf_feat"], sl["kf_idx"], 0.9, True, 3.0, 2, fast_cfg(), kf_seg=sl["kf_seg"])
    assert nl == (outl >= 0).sum()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n_map,n_kf", [("points", 10000, 1500), ("lines", 2000, 200), ("points", 300, 40),
                                             ("lines", 50, 10)])
def test_gpu_fast_driver_bit_exact(ctx, oracle, kind, n_map, n_kf):
    """The shipped fast_matching configuration at BASELINE config 3 sizes: association tables, counts and the
    fall-back decision equal the oracle's."""
    import plslam_amd
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), oracle.make_cam(**synth.EUROC)
    s = scene(n_map, n_kf, lines=(kind == "lines"), seed=n_map + 1)
    a = (s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"])
    seen_used = set()
    for nnr, mutual, th, mm, fm in ((0.75, True, 1.5, 10, fast_cfg()), (0.9, False, 2.0, 6, fast_cfg(ws=1)),
                                    (0.9, True, 0.8, 250 if n_map >= 2000 else 12, fast_cfg()), (0.9, True, 1.0, 0, fast_cfg(ws=5, nnr_grid=0.9)),
                                    (0.8, True, 1.0, 5, fast_cfg(enabled=0))):
        got = ctx.map2kf_match_fast(kind, cam, *a, nnr, mutual, th, mm, fm, kf_seg=s.get("kf_seg"))
        ref = oracle.map2kf_match_fast(kind, ocam, *a, nnr, mutual, th, mm, fm, kf_seg=s.get("kf_seg"))
        np.testing.assert_array_equal(got[0], ref[0])
        assert got[1] == ref[1] == int((ref[0] >= 0).sum()) and got[2] == ref[2]
        seen_used.add(ref[2])
    assert seen_used == {0, 1} or n_map < 100


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n_map,n_kf", [("points", 10000, 1500), ("lines", 2000, 200), ("points", 300, 40)])
def test_gpu_driver_with_the_map_on_the_device(ctx, oracle, kind, n_map, n_kf):
    """plslam_map2kf_match_points_dev / _lines_dev: landmarks, representative descriptors and candidate flags stay on the GPU
    across calls (here: torch tensors, used by several calls with different keyframe-side settings); results equal the
    host-pointer drivers' and the oracle's, with fast_matching and without, fall-back taken and not."""
    import torch
    import plslam_amd
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), oracle.make_cam(**synth.EUROC)
    s = scene(n_map, n_kf, lines=(kind == "lines"), seed=n_map + 7)
    dev = torch.device("cuda", ctx.device)
    d_lm = torch.from_numpy(np.ascontiguousarray(s["LM"], np.float64)).to(dev)
    d_md = torch.from_numpy(np.ascontiguousarray(s["med"], np.uint8)).to(dev)
    d_cd = torch.from_numpy(np.ascontiguousarray(s["cand"], np.uint8)).to(dev)
    a = (s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"])
    seen_used = set()
    for nnr, mutual, th, mm, fm in ((0.75, True, 1.5, 10, fast_cfg()), (0.9, True, 0.8, 250 if n_map >= 2000 else 12, fast_cfg()),
                                    (0.8, True, 1.0, 5, fast_cfg(enabled=0)), (0.9, False, 2.0, 6, fast_cfg(ws=1))):
        got = ctx.map2kf_match_dev(kind, cam, s["Twf"], d_lm.data_ptr(), d_md.data_ptr(), d_cd.data_ptr(), n_map, s["kf_desc"],
                                   s["kf_feat"], s["kf_idx"], nnr, mutual, th, mm, fm, kf_seg=s.get("kf_seg"))
        ref = oracle.map2kf_match_fast(kind, ocam, *a, nnr, mutual, th, mm, fm, kf_seg=s.get("kf_seg"))
        np.testing.assert_array_equal(got[0], ref[0])
        assert got[1] == ref[1] and got[2] == ref[2]
        seen_used.add(ref[2])
    assert seen_used == {0, 1} or n_map < 2000
    # the device-side inputs are untouched
    assert np.array_equal(d_md.cpu().numpy(), s["med"]) and np.array_equal(d_cd.cpu().numpy(), s["cand"])


def kf_pair(n_prev=1500, n_curr=1400, lines=False, seed=3):
    """Two consecutive keyframes: stereo features of the previous one (3D, its camera frame), the relative pose DT and
    the current one's 2D features = re-observations of a subset (+ noise) and clutter."""
    r = np.random.Generator(np.random.PCG64(seed))
    K = synth.EUROC
    DT = synth.se3_exp([0.15, -0.05, 0.3, 0.01, -0.02, 0.015])
    z = r.uniform(1.5, 25.0, n_prev)
    u, v = r.uniform(-30, K["width"] + 30, n_prev), r.uniform(-30, K["height"] + 30, n_prev)
    P = np.stack([(u - K["cx"]) / K["fx"] * z, (v - K["cy"]) / K["fy"] * z, z], 1)
    P[r.random(n_prev) < 0.01, 2] = -1.0                       # a few behind the current camera after DT / degenerate
    proj = lambda X: np.stack([K["cx"] + K["fx"] * X[:, 0] / X[:, 2], K["cy"] + K["fy"] * X[:, 1] / X[:, 2]], 1)
    d_prev = synth.random_desc(r, n_prev)
    src = r.integers(0, n_prev, n_curr)
    d_curr = d_prev[src] ^ np.packbits(r.random((n_curr, 256)) < 0.05, axis=1)
    clutter = r.random(n_curr) < 0.3
    d_curr[clutter] = synth.random_desc(r, int(clutter.sum()))
    with np.errstate(all="ignore"):
        Pc = P @ DT[:3, :3].T + DT[:3, 3]
        pl = np.nan_to_num(proj(Pc[src]), posinf=0, neginf=0) + r.normal(0, 1.5, (n_curr, 2))
        if not lines:
            return dict(DT=DT, X=P, d_prev=d_prev, feat=pl, d_curr=np.ascontiguousarray(d_curr))
        E = P + r.uniform(-0.6, 0.6, (n_prev, 3))
        Ec = E @ DT[:3, :3].T + DT[:3, 3]
        seg = np.concatenate([pl, np.nan_to_num(proj(Ec[src]), posinf=0, neginf=0) + r.normal(0, 1.5, (n_curr, 2))], 1)
    return dict(DT=DT, X=np.concatenate([P, E], 1), d_prev=d_prev, feat=seg, d_curr=np.ascontiguousarray(d_curr))


def test_oracle_kf2kf_semantics(oracle):
    cam = oracle.make_cam(**synth.EUROC)
    s = kf_pair(600, 550)
    a = (s["DT"], s["X"], s["d_prev"], s["feat"], s["d_curr"])
    m, n, used = oracle.kf2kf_match("points", cam, *a, 0.75, True, 20, fast_cfg())
    assert used == 0 and n == (m >= 0).sum() > 100
    bf, nbf = oracle.match(s["d_prev"], s["d_curr"], 0.75, True)
    m2, n2, used2 = oracle.kf2kf_match("points", cam, *a, 0.75, True, 20, fast_cfg(enabled=0))
    assert used2 == 1 and n2 == nbf and (m2 == bf).all()          # fast_matching off: plain StVO::match
    m3, n3, used3 = oracle.kf2kf_match("points", cam, *a, 0.75, True, n + 10, fast_cfg())
    exp3, nexp3 = oracle.match_prior(s["d_prev"], s["d_curr"], 0.75, True, m)
    assert used3 == 1 and (m3 == exp3).all() and n3 == nexp3      # too few grid matches: StVO::match on the same vector
    assert (m3[bf >= 0] == bf[bf >= 0]).all() and (m3 >= 0).sum() >= nbf
    m4, n4, used4 = oracle.kf2kf_match("points", cam, *a, 0.75, True, 600, fast_cfg())
    assert used4 == 0 and (m4 == m).all()                         # ... unless a keyframe has no more than min_matches features
    # lines: pj_lines are pixels used as cells (:392-393): almost nothing falls into the 64 x 48 grid, the fall-back runs
    sl = kf_pair(200, 180, lines=True)
    ml, nl, usedl = oracle.kf2kf_match("lines", cam, sl["DT"], sl["X"], sl["d_prev"], sl["feat"], sl["d_curr"], 0.9, True, 10,
                                       fast_cfg())
    assert usedl == 1 and nl == oracle.match(sl["d_prev"], sl["d_curr"], 0.9, True)[1]


@pytest.mark.gpu
def test_gpu_kf2kf_lines_when_the_windowed_pass_can_succeed(ctx, oracle):
    """matchKF2KFLines hands matchGrid PIXEL coordinates (:392-393): on a real image nearly every window misses the 64 x 48 grid and
    the driver, knowing so from a host-side bound, enqueues StVO::match behind the windowed pass at once (round 6).  Here the camera
    is so small that the projected pixels ARE grid cells (fx = 30, 64 x 48 pixels): the bound is large, the windowed pass finds its
    matches, and the call must take the ordinary road -- with min_matches below what it finds (no StVO::match), above it (the
    fall-back after the count is known) and far above the bound's reach.  Bit-exact against the oracle every time."""
    import plslam_amd
    K = dict(fx=30.0, fy=30.0, cx=32.0, cy=24.0, b=0.11, width=64, height=48)
    cam, ocam = plslam_amd.make_cam(**K), oracle.make_cam(**K)
    r = np.random.Generator(np.random.PCG64(12))
    n_prev, n_curr = 220, 200
    DT = synth.se3_exp([0.02, -0.01, 0.05, 0.002, -0.003, 0.002])
    z = r.uniform(2.0, 12.0, n_prev)
    P = np.stack([r.uniform(-0.9, 0.9, n_prev) * z, r.uniform(-0.7, 0.7, n_prev) * z, z], 1)
    E = P + r.uniform(-0.3, 0.3, (n_prev, 3))
    proj = lambda X: np.stack([K["cx"] + K["fx"] * X[:, 0] / X[:, 2], K["cy"] + K["fy"] * X[:, 1] / X[:, 2]], 1)
    d_prev = synth.random_desc(r, n_prev)
    src = r.permutation(n_prev)[:n_curr]
    d_curr = d_prev[src] ^ np.packbits(r.random((n_curr, 256)) < 0.04, axis=1)
    Pc, Ec = P @ DT[:3, :3].T + DT[:3, 3], E @ DT[:3, :3].T + DT[:3, 3]
    seg = np.concatenate([proj(Pc[src]), proj(Ec[src])], 1) + r.normal(0, 0.2, (n_curr, 4))
    a = (DT, np.concatenate([P, E], 1), d_prev, seg, np.ascontiguousarray(d_curr))
    fm = dict(enabled=1, grid_cols=64, grid_rows=48, ws=3, inv_width=1.0, inv_height=1.0, nnr_grid=0.85, line_sim_th=0.75)
    base = oracle.kf2kf_match("lines", ocam, *a, 0.85, True, 5, fm)
    assert base[2] == 0 and base[1] > 40                      # the windowed pass alone: plenty of matches, no StVO::match
    seen = set()
    for mm in (5, base[1] + 20, 199, 10 ** 6):
        got = ctx.kf2kf_match("lines", cam, *a, 0.85, True, mm, fm)
        ref = oracle.kf2kf_match("lines", ocam, *a, 0.85, True, mm, fm)
        np.testing.assert_array_equal(got[0], ref[0])
        assert got[1] == ref[1] and got[2] == ref[2], mm
        seen.add(ref[2])
    assert seen == {0, 1}


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n_prev,n_curr", [("points", 1500, 1400), ("lines", 200, 180), ("points", 60, 50),
                                                ("lines", 30, 40), ("points", 4000, 4000)])
def test_gpu_kf2kf_driver_bit_exact(ctx, oracle, kind, n_prev, n_curr):
    import plslam_amd
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), oracle.make_cam(**synth.EUROC)
    s = kf_pair(n_prev, n_curr, lines=(kind == "lines"), seed=n_prev)
    a = (s["DT"], s["X"], s["d_prev"], s["feat"], s["d_curr"])
    seen = set()
    for nnr, mutual, mm, fm in ((0.75, True, 20, fast_cfg()), (0.9, False, 5, fast_cfg(ws=1)), (0.9, True, 10 ** 6, fast_cfg()),
                                (0.8, True, 10, fast_cfg(enabled=0)), (0.8, True, 0, fast_cfg(enabled=0)),
                                (0.75, True, max(min(n_prev, n_curr) - 1, 0), fast_cfg(ws=6, nnr_grid=0.9))):
        got = ctx.kf2kf_match(kind, cam, *a, nnr, mutual, mm, fm)
        ref = oracle.kf2kf_match(kind, ocam, *a, nnr, mutual, mm, fm)
        np.testing.assert_array_equal(got[0], ref[0])
        assert got[1] == ref[1] and got[2] == ref[2]
        # the same with the rows resident on the device (plslam_kf2kf_match_*_dev): nothing but the grid goes up
        import torch
        dX, dP, dC = (torch.from_numpy(np.ascontiguousarray(s[k])).cuda() for k in ("X", "d_prev", "d_curr"))
        gotd = ctx.kf2kf_match_dev(kind, cam, s["DT"], dX.data_ptr(), dP.data_ptr(), n_prev, s["feat"], dC.data_ptr(), nnr, mutual, mm, fm)
        np.testing.assert_array_equal(gotd[0], ref[0])
        assert gotd[1] == ref[1] and gotd[2] == ref[2]
        # the count is the number of entries, except after the fall-back ON matchGrid's vector: entries that were kept
        # were never counted, kept entries that fail the consistency loop are subtracted (the reference's arithmetic)
        assert ref[1] == int((ref[0] >= 0).sum()) or (ref[2] == 1 and fm["enabled"] and mutual)
        seen.add(ref[2])
    assert seen == {0, 1}


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["points", "lines"])
def test_gpu_fast_driver_without_visible_candidates_and_on_a_large_map(ctx, oracle, kind):
    """The one-launch form of the fast_matching drivers with NOTHING to match -- every candidate flag zero, or every landmark
    behind the camera: the row count the kernels read from the device is zero, matchGrid's list is empty -- and with a map far
    larger than one workgroup's share (70 001 landmarks: k_visible_compact's workgroups chain their counts over 274 links, the
    last one partly filled), through the host-pointer and the device-resident entry points."""
    import torch
    import plslam_amd
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), oracle.make_cam(**synth.EUROC)
    lines = kind == "lines"
    dev = torch.device("cuda", ctx.device)

    def both(s, cand, lm, nnr=0.75, mm=10, fm=None):
        fm = fm or fast_cfg()
        a = (s["Twf"], lm, s["med"], cand, s["kf_desc"], s["kf_feat"], s["kf_idx"])
        ref = oracle.map2kf_match_fast(kind, ocam, *a, nnr, True, 1.5, mm, fm, kf_seg=s.get("kf_seg"))
        got = ctx.map2kf_match_fast(kind, cam, *a, nnr, True, 1.5, mm, fm, kf_seg=s.get("kf_seg"))
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (lm, s["med"], cand)]
        gotd = ctx.map2kf_match_dev(kind, cam, s["Twf"], d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), lm.shape[0], s["kf_desc"],
                                    s["kf_feat"], s["kf_idx"], nnr, True, 1.5, mm, fm, kf_seg=s.get("kf_seg"))
        for g in (got, gotd):
            np.testing.assert_array_equal(g[0], ref[0])
            assert g[1] == ref[1] and g[2] == ref[2]
        return ref

    s = scene(3000, 400, lines=lines, seed=77)
    ref = both(s, np.zeros_like(s["cand"]), s["LM"])                       # no candidate flag set
    assert ref[1] == 0 and (ref[0] == -1).all()
    behind = np.array(s["LM"], np.float64)
    Tfw = np.linalg.inv(s["Twf"])
    for e in range(2 if lines else 1):                                      # every landmark 50 m behind the camera
        Xc = np.stack([np.zeros(3000), np.zeros(3000), np.full(3000, -50.0)], 1)
        behind[:, 3 * e:3 * e + 3] = Xc @ Tfw[:3, :3].T + Tfw[:3, 3]
    ref = both(s, s["cand"], behind)
    assert ref[1] == 0 and (ref[0] == -1).all()
    big = scene(70001, 600, lines=lines, seed=78)
    ref = both(big, big["cand"], big["LM"], nnr=0.9, mm=5)
    assert ref[1] > 0
    # a 1 M-landmark map (ADVICE r4 / VERDICT r5: the count chain was an O(b) spin per workgroup): 3 907 workgroups of
    # k_visible_compact find their list offsets by decoupled look-back; candidate flags on every 150th landmark keep the list --
    # which must come out ascending and complete -- at a size the checker matches in a moment
    huge = scene(1_000_000, 600, lines=lines, seed=79)
    sparse = np.zeros_like(huge["cand"])
    sparse[::150] = huge["cand"][::150]
    sparse[-1] = 1                                                          # (the last, partly filled workgroup has work too)
    ref = both(huge, sparse, huge["LM"], nnr=0.9, mm=5)
    assert ref[1] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["points", "lines"])
def test_gpu_brute_force_driver_in_one_synchronisation(ctx, oracle, kind):
    """Without fast_matching the drivers run as ONE launch sequence too: the candidate list and its length stay on the device,
    the matcher is a two-launch column-split plan sized for the whole map whose kernels read the row count themselves.  Cases
    the C3-sized tests do not reach: no candidate at all / every landmark behind the camera (row count zero), fewer candidates
    than min_matches (the matcher's table must not be used), one visible candidate, a 70 001-landmark map, a map of a few
    landmarks (one column range), and a context whose options the plan cannot honour (it falls back to the step-by-step form).
    Host-pointer and device-resident entry points against the oracle."""
    import torch
    import plslam_amd
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), oracle.make_cam(**synth.EUROC)
    lines = kind == "lines"
    dev = torch.device("cuda", ctx.device)
    off = fast_cfg(enabled=0)

    def both(s, cand, lm, nnr=0.75, mm=10, th=1.5):
        a = (s["Twf"], lm, s["med"], cand, s["kf_desc"], s["kf_feat"], s["kf_idx"])
        ref = oracle.map2kf_match(kind, ocam, *a, nnr, True, th, mm)
        got = ctx.map2kf_match(kind, cam, *a, nnr, True, th, mm)
        d = [torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (lm, s["med"], cand)]
        gotd = ctx.map2kf_match_dev(kind, cam, s["Twf"], d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), lm.shape[0], s["kf_desc"],
                                    s["kf_feat"], s["kf_idx"], nnr, True, th, mm, off, kf_seg=s.get("kf_seg"))
        for g in (got, gotd):
            np.testing.assert_array_equal(g[0], ref[0])
            assert g[1] == ref[1]
        return ref

    s = scene(3000, 400, lines=lines, seed=177)
    n = s["LM"].shape[0]
    ref = both(s, s["cand"], s["LM"])
    assert ref[1] > 0
    ref = both(s, np.zeros_like(s["cand"]), s["LM"])                       # no candidate flag set: row count zero
    assert ref[1] == 0 and (ref[0] == -1).all()
    behind = np.array(s["LM"], np.float64)
    Tfw = np.linalg.inv(s["Twf"])
    for e in range(2 if lines else 1):                                      # every landmark 50 m behind the camera
        Xc = np.stack([np.zeros(n), np.zeros(n), np.full(n, -50.0)], 1)
        behind[:, 3 * e:3 * e + 3] = Xc @ Tfw[:3, :3].T + Tfw[:3, 3]
    ref = both(s, s["cand"], behind)
    assert ref[1] == 0 and (ref[0] == -1).all()
    few = np.zeros_like(s["cand"])
    few[np.flatnonzero(s["cand"])[:40]] = 1                                 # at most 40 candidates: below min_matches = 60 ...
    ref = both(s, few, s["LM"], mm=60)
    assert ref[1] == 0 and (ref[0] == -1).all()
    both(s, few, s["LM"], mm=3)                                             # ... and above another
    one = np.zeros_like(s["cand"])
    one[np.flatnonzero(s["cand"])[7]] = 1
    both(s, one, s["LM"], mm=1)
    tiny = scene(90, 30, lines=lines, seed=178)                             # one column range, one row block
    both(tiny, tiny["cand"], tiny["LM"], nnr=0.9, mm=2)
    big = scene(70001, 600, lines=lines, seed=179)
    ref = both(big, big["cand"], big["LM"], nnr=0.9, mm=5)
    assert ref[1] > 0
    try:                                                                    # options the device-count plan refuses
        for key, val in (("scan_variant", plslam_amd.SCAN_LANE_PER_QUERY), ("mfma_form", 5), ("split_post", 1), ("col_split", 1)):
            ctx.set_option(key, val)
            ref = both(s, s["cand"], s["LM"], mm=5)
            assert ref[1] > 0
            ctx.set_option(key, 0)
    finally:
        for key in ("scan_variant", "mfma_form", "split_post", "col_split"):
            ctx.set_option(key, 0)
