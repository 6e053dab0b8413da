#!/usr/bin/env python3
"""Single-call latency of the host-pointer entry point plslam_match (the in-loop SLAM use: one
StVO::match per call, descriptors in host memory), against the CPU oracle on one thread."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plslam_amd  # noqa: E402
from plslam_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    ctx = plslam_amd.Context(0)
    r = np.random.Generator(np.random.PCG64(3))
    out = {}
    for name, n1, n2, nnr in (("orb_1500x1500", 1500, 1500, 0.75), ("lbd_200x200", 200, 200, 0.75),
                              ("map_10000x1500", 10000, 1500, 0.75), ("orb_800x800_kitti", 800, 800, 0.75)):
        d1 = synth.random_desc(r, n1)
        d2 = np.concatenate([synth.noisy_copy(r, d1)[0][: min(n1, n2)], synth.random_desc(r, max(0, n2 - n1))])[:n2]
        for _ in range(5):
            ctx.match(d1, d2, nnr, True)
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            m, n = ctx.match(d1, d2, nnr, True)
            ts.append(time.perf_counter() - t0)
        L = O.native_lib()
        t0 = time.perf_counter()
        em, en = O.match(d1, d2, nnr, True, L=L)
        cpu = time.perf_counter() - t0
        assert np.array_equal(m, em)
        ts = np.array(ts) * 1e6
        out[name] = {"gpu_call_us_median": float(np.median(ts)), "gpu_call_us_p10": float(np.percentile(ts, 10)),
                     "gpu_call_us_p90": float(np.percentile(ts, 90)), "cpu_oracle_1thread_us": cpu * 1e6}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
