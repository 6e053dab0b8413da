#!/usr/bin/env python3
"""Latency of ONE host-pointer plslam_match_grid call (the SLAM loop's call pattern, src/mapHandler.cpp:271,418,591,706) for the
KF<->KF point problem (1500 x 1500, 64 x 48 grid, window 3) and the line problem (200 x 200), per value of the context option
zero_copy_kb.  usage: grid_call_latency.py [points|lines|both] [calls] [zero_copy_kb ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import plslam_amd  # noqa: E402
from plslam_amd import grid as G  # noqa: E402
from oracle import oracle as O  # noqa: E402
from test_match_grid_cpu import line_case, point_case  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "both"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 200
zcs = [int(x) for x in sys.argv[3:]] or [0, 64, 1024]
ctx = plslam_amd.Context(0)
W = (3, 3, 3, 3)
for name, mk, n in (("points_1500x1500", point_case, 1500), ("lines_200x200", line_case, 200)):
    if which not in ("both", name.split("_")[0]):
        continue
    c = mk(11, n, n, G.GRID_COLS, G.GRID_ROWS)
    rm, rk = O.match_grid(window=W, nnr=0.75, mutual=True, **c)
    for zc in zcs:
        ctx.set_option("grid_dense", 0 if zc < 0 else 1)       # (a negative value: the general kernels, no zero copy)
        zc = max(zc, 0)
        ctx.set_option("zero_copy_kb", zc)
        for _ in range(10):
            m, k = ctx.match_grid(window=W, nnr=0.75, mutual=True, **c)
        assert np.array_equal(m, rm) and k == rk
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            ctx.match_grid(window=W, nnr=0.75, mutual=True, **c)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        print(f"{name} dense {ctx.get_option('grid_dense')} zero_copy_kb {zc:5d}: median {np.median(ts):6.1f} us  p10 {np.percentile(ts, 10):6.1f}  p90 {np.percentile(ts, 90):6.1f}  ({k} matches, bit-exact vs the oracle)")
ctx.set_option("zero_copy_kb", 64)
