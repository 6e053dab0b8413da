#!/usr/bin/env python3
"""One driver call pattern repeated, for rocprofv3 --kernel-trace (tools/kt.sh): where the time of a map<->keyframe /
keyframe<->keyframe call goes.  usage: driver_trace.py map2kf_points|map2kf_lines|kf2kf_points|kf2kf_lines [fast 0|1] [reps]
Prints the wall time per call; the kernel trace gives the device side of it."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import plslam_amd  # noqa: E402
from plslam_amd import synth  # noqa: E402
import test_map2kf as TM  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "map2kf_points"
fast = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
min_matches = int(sys.argv[4]) if len(sys.argv) > 4 else None       # default: 10 (map<->KF) / 20 (KF<->KF) as the bench's records
ctx = plslam_amd.Context(0)
cam = plslam_amd.make_cam(**synth.EUROC)
fm = TM.fast_cfg(enabled=fast)
kind = "lines" if what.endswith("lines") else "points"
if what.startswith("map2kf"):
    n_map, n_kf = (10000, 1500) if kind == "points" else (2000, 200)
    s = TM.scene(n_map, n_kf, lines=(kind == "lines"), seed=n_map + 1)
    a = (s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"])
    call = lambda: ctx.map2kf_match_fast(kind, cam, *a, 0.9, True, 1.5, 10 if min_matches is None else min_matches, fm, kf_seg=s.get("kf_seg"))   # noqa: E731
    extra = f"candidates {int(np.count_nonzero(s['cand']))} of {n_map}, unmatched keyframe features {int((s['kf_idx'] == -1).sum())}"
else:
    n = 1500 if kind == "points" else 200
    s = TM.kf_pair(n, n - 100, lines=(kind == "lines"), seed=n)
    a = (s["DT"], s["X"], s["d_prev"], s["feat"], s["d_curr"])
    call = lambda: ctx.kf2kf_match(kind, cam, *a, 0.75, True, 20 if min_matches is None else min_matches, fm)   # noqa: E731
    extra = ""
for _ in range(5):
    out = call()
t0 = time.perf_counter()
for _ in range(reps):
    call()
dt = (time.perf_counter() - t0) / reps
print(f"{what} fast_matching {fast}: {1e6 * dt:.1f} us per call over {reps} calls; result count {out[1]}, match() used {out[2]}; {extra}")
