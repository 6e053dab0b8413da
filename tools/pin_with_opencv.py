#!/usr/bin/env python3
"""Settles the "parity unpinned" part of DESIGN.md section 3 on a machine that has OpenCV (and, optionally, a stvo-pl
build): replays the committed golden vectors through the REAL cv::BFMatcher and reports every difference.

What it checks with `cv2` alone (pip opencv-python is enough):
  * tests/golden/match_golden.npz -- for every case: cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=False).knnMatch(q, t, k=2)
    against the stored (knn_idx, knn_dist): the tie order of kNN-2 (lowest trainIdx among equal distances, second neighbour
    may share the first one's distance) is the one thing the oracle RECALLS about OpenCV's batchDistance;
  * the ratio test and the mutual check exactly as stvo-pl's matchNNR / match are recalled (accept iff
    matches[i][0].distance < matches[i][1].distance * nnr in float; keep i1 -> i2 iff matches_21[i2] == i1), computed from
    cv2's OWN knn results, against the stored m12 tables for nnr in (0.6, 0.75, 0.9) x mutual in (0, 1).
What needs a stvo-pl checkout: tools/pin_stvo/ -- `export_cases.py` writes the goldens (match, grid, stereo gates) as flat
binary arrays + a manifest, `pin_stvo.cpp` (Makefile beside it: STVO_PL_DIR + pkg-config OpenCV) links stvo-pl's own
matching.cpp / gridStructure.cpp / config.cpp and replays every match and grid case through StVO::match and both
StVO::matchGrid overloads; the stereo-gate cases (tests/golden/stereo_gates_golden.npz) are exported with inputs, thresholds
and expected tables for a comparison inside StereoFrame::matchStereoPoints / matchStereoLines, which cannot be called on
their own.  (A Python module exposing match / matchGrid can be plugged in here instead: `--stvo-module NAME`.)

Exit code 0 = every replayed vector agrees (or nothing could be replayed: no cv2 -- says so), 1 = differences (listed).
This container has neither OpenCV nor stvo-pl: here the script only reports that.  It never touches the product.
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def replay_match_golden(cv2):
    g = np.load(os.path.join(GOLD, "match_golden.npz"))
    bad = []
    names = sorted({k.split("/")[0] for k in g.files})
    bf = cv2.BFMatcher(cv2.NORM_HAMMING, crossCheck=False)

    def knn(q, t):
        if len(q) == 0 or len(t) == 0:
            return np.full((len(q), 2), -1, np.int32), np.full((len(q), 2), np.iinfo(np.int32).max, np.int32)
        res = bf.knnMatch(q, t, k=2)
        idx = np.full((len(q), 2), -1, np.int32)
        dist = np.full((len(q), 2), np.iinfo(np.int32).max, np.int32)
        for i, ms in enumerate(res):
            for k, m in enumerate(ms[:2]):
                idx[i, k], dist[i, k] = m.trainIdx, int(m.distance)
        return idx, dist

    def nnr_table(idx, dist, nnr):
        out = np.full(len(idx), -1, np.int32)
        ok = (idx[:, 1] >= 0) & (dist[:, 0].astype(np.float32) < dist[:, 1].astype(np.float32) * np.float32(nnr))
        out[ok] = idx[ok, 0]
        return out

    for n in names:
        q, t = g[f"{n}/q"], g[f"{n}/t"]
        idx, dist = knn(q, t)
        if not np.array_equal(dist, g[f"{n}/knn_dist"]):
            bad.append((n, "knn distances", int((dist != g[f'{n}/knn_dist']).sum())))
        if not np.array_equal(idx, g[f"{n}/knn_idx"]):
            bad.append((n, "knn indices (tie order)", int((idx != g[f'{n}/knn_idx']).sum())))
        idx21, dist21 = knn(t, q)
        for nnr in (0.6, 0.75, 0.9):
            m12, m21 = nnr_table(idx, dist, nnr), nnr_table(idx21, dist21, nnr)
            for mut in (0, 1):
                key = f"{n}/m12_nnr{nnr}_mut{mut}"
                if key not in g.files:
                    continue
                got = m12.copy()
                if mut:
                    sel = got >= 0
                    got[sel] = np.where(m21[got[sel]] == np.nonzero(sel)[0], got[sel], -1)
                if not np.array_equal(got, g[key]):
                    bad.append((n, f"match table nnr {nnr} mutual {mut}", int((got != g[key]).sum())))
    return len(names), bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stvo-module", default=None, help="importable module exposing stvo-pl's match / matchGrid / stereo gates")
    a = ap.parse_args()
    try:
        cv2 = importlib.import_module("cv2")
    except ImportError:
        print("cv2 is not importable here: nothing replayed (the unpinned items of DESIGN.md section 3 stay unpinned).")
        return 0
    n, bad = replay_match_golden(cv2)
    print(f"match_golden.npz: {n} cases replayed through cv2.BFMatcher {cv2.__version__}: {len(bad)} difference(s)")
    for b in bad:
        print("  ", b)
    table = [("match_golden.npz", n, n - len({b[0] for b in bad}), len({b[0] for b in bad}), "cv2.BFMatcher")]
    if a.stvo_module:
        try:
            stvo = importlib.import_module(a.stvo_module)
        except ImportError as e:
            print(f"--stvo-module {a.stvo_module}: {e}")
            return 1 if bad else 0
        g = np.load(os.path.join(GOLD, "grid_golden.npz"))
        cases = sorted({k.split("/")[0] for k in g.files})
        gbad = 0
        for c in cases:
            kw = {k.split("/", 1)[1]: g[k] for k in g.files if k.startswith(c + "/") and not k.endswith("/m12")}
            got = np.asarray(stvo.matchGrid(**kw), np.int32)
            gbad += int(not np.array_equal(got, g[f"{c}/m12"]))
        print(f"grid_golden.npz: {len(cases)} cases through {a.stvo_module}.matchGrid: {gbad} differ")
        bad += [("grid", gbad)] if gbad else []
        table.append(("grid_golden.npz", len(cases), len(cases) - gbad, gbad, a.stvo_module))
    else:
        print("matchGrid / stereo gates / SE(3) helpers: need a stvo-pl build (make -C tools/pin_stvo check, or --stvo-module); not replayed.")
        table += [(f, 0, 0, 0, "needs stvo-pl: make -C tools/pin_stvo check STVO_PL_DIR=...")
                  for f in ("grid_golden.npz", "stereo_gates_golden.npz", "se3_helpers_golden.npz")]
    print(f"\n{'golden file':28s} {'replayed':>8s} {'PASS':>6s} {'FAIL':>6s}  through")
    for f, n_, ok, ko, how in table:
        print(f"{f:28s} {n_:8d} {ok:6d} {ko:6d}  {how}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
