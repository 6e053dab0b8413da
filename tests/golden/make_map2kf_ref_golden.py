"""Generates tests/golden/map2kf_ref_golden.npz.  Run from the repo root IN THE BUILD CONTAINER (needs /root/reference):
    python tests/golden/make_map2kf_ref_golden.py

Outputs of the REFERENCE'S OWN SOURCE TEXT for the loops either side of the map <-> key-frame descriptor match and for the
representative descriptor: matchMap2KFPoints visibility pre-filter (src/mapHandler.cpp:545-558) and gate (:601-629),
matchMap2KFLines (:647-663, :716-749) -- cut out of the file where it lies and compiled textually
(oracle/ref_wrap_map2kf.cpp) -- and MapPoint / MapLine::updateAverageDescDir (src/mapFeatures.cpp:51-93, :121-163;
mapFeatures.cpp compiled as is, oracle/ref_wrap_mapfeatures.cpp).  Written only if the C oracle agrees exactly.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from plslam_amd import synth  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "map2kf_ref_golden.npz")


def main():
    K = synth.EUROC
    cam = O.make_cam(**K)
    r = np.random.Generator(np.random.PCG64(20240))
    Twf = np.linalg.inv(synth.se3_exp(r.normal(0, 0.08, 6)))
    n, nt = 400, 160
    X = np.stack([r.uniform(-4, 4, n), r.uniform(-3, 3, n), r.uniform(-2, 20, n)], 1)
    Lw = np.concatenate([X, X + r.normal(0, 0.4, X.shape)], 1)
    ref = O.ref_map_visible("points", cam, Twf, X, 1.0 / K["width"], 1.0 / K["height"])
    if ref is None:
        raise SystemExit("oracle/_ref lacks ref_map_visible: run `make -C oracle ref` with /root/reference present")
    vis_p, pj_p = ref
    vis_l, pj_l = O.ref_map_visible("lines", cam, Twf, Lw, 1.0 / K["width"], 1.0 / K["height"])
    assert np.array_equal(vis_p, O.map_point_visible(cam, Twf, X)) and np.array_equal(vis_l, O.map_line_visible(cam, Twf, Lw))
    Xc = X @ Twf[:3, :3].T + Twf[:3, 3]
    uv = np.stack([K["cx"] + K["fx"] * Xc[:, 0] / Xc[:, 2], K["cy"] + K["fy"] * Xc[:, 1] / Xc[:, 2]], 1)
    Xv, Lv = X[vis_p.astype(bool)], Lw[vis_l.astype(bool)]
    m12p = r.integers(-1, nt, len(Xv)).astype(np.int32)
    pl = r.uniform(0, 700, (nt, 2))
    ok = m12p >= 0
    pl[m12p[ok]] = uv[vis_p.astype(bool)][ok] + r.normal(0, 0.8, (int(ok.sum()), 2))
    m12l = r.integers(-1, nt, len(Lv)).astype(np.int32)
    le = r.normal(0, 1, (nt, 3))
    le /= np.linalg.norm(le[:, :2], axis=1, keepdims=True)
    le[:, 2] *= 200.0
    out = dict(Twf=Twf.reshape(16), X=X, Lw=Lw, vis_p=vis_p, vis_l=vis_l, pj_p=pj_p, pj_l=pj_l, m12p=m12p, pl=pl, m12l=m12l, le=le,
               th_p=np.array([1.0, 2.5]), th_l=np.array([1.0, 40.0]))
    for k, th in enumerate(out["th_p"]):
        mask, cnt = O.ref_map2kf_gate("points", cam, Twf, Xv, m12p, pl, th)
        om, oc = O.map2kf_point_gate(cam, Twf, Xv, m12p, pl, th)
        assert np.array_equal(mask, om) and cnt == oc and 0 < cnt < ok.sum()
        out[f"gate_p{k}"], out[f"count_p{k}"] = mask, np.array([cnt])
    for k, th in enumerate(out["th_l"]):
        mask, cnt = O.ref_map2kf_gate("lines", cam, Twf, Lv, m12l, le, th)
        om, oc = O.map2kf_line_gate(cam, Twf, Lv, m12l, le, th)
        assert np.array_equal(mask, om) and cnt == oc and cnt > 0
        out[f"gate_l{k}"], out[f"count_l{k}"] = mask, np.array([cnt])
    # representative descriptors: 60 landmarks, 1..12 observations, some tie-heavy, some with duplicated observations
    lists, offs, idx_p, idx_l = [], [0], [], []
    for j in range(60):
        m = int(r.integers(1, 13))
        d = synth.tie_stress_desc(r, m) if j % 3 == 0 else synth.random_desc(r, m)
        if j % 5 == 0 and m > 2:
            d[int(r.integers(1, m))] = d[0]
        a, b = O.ref_median_desc(d, "point"), O.ref_median_desc(d, "line")
        assert a == b == O.median_desc(d)
        lists.append(d)
        offs.append(offs[-1] + m)
        idx_p.append(a)
    out.update(med_desc_lists=np.concatenate(lists), med_offsets=np.array(offs, np.int32), med_idx=np.array(idx_p, np.int32))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
