#!/usr/bin/env python3
"""Single-call latency of the fused map<->keyframe / keyframe<->keyframe drivers (host pointers, what the local-mapping
thread calls per keyframe) at BASELINE config 3 sizes, against the oracle on one host thread.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import plslam_amd  # noqa: E402
from plslam_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402
import test_map2kf as T  # noqa: E402


def timed(f, reps=30):
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e6)


def main():
    ctx = plslam_amd.Context(0)
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), O.make_cam(**synth.EUROC)
    out = {}
    for kind, n_map, n_kf in (("points", 10000, 1500), ("lines", 2000, 200)):
        s = T.scene(n_map, n_kf, lines=(kind == "lines"), seed=n_map + 1)
        a = (s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"])
        for name, fm in (("fast_matching", T.fast_cfg()), ("brute_force", T.fast_cfg(enabled=0))):
            g = lambda: ctx.map2kf_match_fast(kind, cam, *a, 0.9, True, 1.5, 10, fm, kf_seg=s.get("kf_seg"))
            got = g()
            t0 = time.perf_counter()
            ref = O.map2kf_match_fast(kind, ocam, *a, 0.9, True, 1.5, 10, fm, kf_seg=s.get("kf_seg"))
            cpu = (time.perf_counter() - t0) * 1e6
            assert np.array_equal(got[0], ref[0]) and got[1] == ref[1]
            out[f"map2kf_{kind}_{n_map}x{n_kf}_{name}"] = {"gpu_call_us_median": timed(g), "cpu_oracle_1thread_us": cpu,
                                                         "associations": int(ref[1])}
    for kind, n in (("points", 1500), ("lines", 200)):
        s = T.kf_pair(n, n - 100, lines=(kind == "lines"), seed=n)
        a = (s["DT"], s["X"], s["d_prev"], s["feat"], s["d_curr"])
        for name, fm in (("fast_matching", T.fast_cfg()), ("brute_force", T.fast_cfg(enabled=0))):
            g = lambda: ctx.kf2kf_match(kind, cam, *a, 0.75, True, 20, fm)
            got = g()
            t0 = time.perf_counter()
            ref = O.kf2kf_match(kind, ocam, *a, 0.75, True, 20, fm)
            cpu = (time.perf_counter() - t0) * 1e6
            assert np.array_equal(got[0], ref[0]) and got[1] == ref[1]
            out[f"kf2kf_{kind}_{n}_{name}"] = {"gpu_call_us_median": timed(g), "cpu_oracle_1thread_us": cpu, "matches": int(ref[1])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
