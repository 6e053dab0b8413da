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
    return dict(Twf=Twf, LM=Lw, med=med, cand=cand, kf_desc=kf_desc, kf_feat=le, kf_idx=kf_idx)


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
    for nnr, mutual, th, mm in ((0.75, True, 1.0, 10), (0.9, False, 2.0, 6), (0.9, True, 0.5, 0)):
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
