#!/usr/bin/env python3
"""The two-launch matchGrid path lists its candidates in whatever order the scheduling makes it; nothing downstream may depend on
it.  Repeats the driver calls and the plain matchGrid call and compares every result with the first one.  usage: [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import plslam_amd  # noqa: E402
from plslam_amd import synth  # noqa: E402
import test_map2kf as TM  # noqa: E402
import test_match_grid_cpu as TG  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ctx = plslam_amd.Context(0)
cam = plslam_amd.make_cam(**synth.EUROC)
fm = TM.fast_cfg()
calls = {}
for kind, n_map, n_kf in (("points", 10000, 1500), ("lines", 2000, 200), ("points", 3000, 900)):
    s = TM.scene(n_map, n_kf, lines=(kind == "lines"), seed=n_map + 1)
    a = (s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"])
    calls[f"map2kf_{kind}_{n_map}"] = (lambda kind=kind, a=a, s=s: ctx.map2kf_match_fast(kind, cam, *a, 0.9, True, 1.5, 10, fm, kf_seg=s.get("kf_seg")))
for kind, n in (("points", 1500), ("lines", 600)):
    s = TM.kf_pair(n, n - 100, lines=(kind == "lines"), seed=n)
    a = (s["DT"], s["X"], s["d_prev"], s["feat"], s["d_curr"])
    calls[f"kf2kf_{kind}_{n}"] = (lambda kind=kind, a=a: ctx.kf2kf_match(kind, cam, *a, 0.75, True, 20, fm))
c = TG.point_case(7, 1500, 1500, 64, 48)
calls["match_grid_1500"] = lambda: ctx.match_grid(window=(3, 3, 3, 3), nnr=0.75, mutual=True, **c)
bad = 0
for name, f in calls.items():
    ref = f()
    diff = 0
    for _ in range(reps):
        got = f()
        diff += int(not (np.array_equal(got[0], ref[0]) and got[1] == ref[1]))
    print(f"{name}: {reps} repetitions, {diff} differ from the first")
    bad += diff
sys.exit(1 if bad else 0)
