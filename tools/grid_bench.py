#!/usr/bin/env python3
"""K14 (StVO::matchGrid) measurements on one MI355X:
  * single-call latency of the host-pointer entry point plslam_match_grid for the KF<->KF point problem at BASELINE
    config 2's size (1500 x 1500 rows, 64 x 48 grid, matching_f2f_ws = 3) and the line problem (200 x 200), against
    the oracle on one host thread and against the brute-force plslam_match on the same descriptors;
  * batch throughput of the device-resident plan (one workgroup per problem): problems/s and candidate pairs/s.
Prints one JSON object."""
import json
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


def main():
    import torch
    ctx = plslam_amd.Context(0)
    dev = torch.device("cuda", 0)
    out = {}
    W = (3, 3, 3, 3)
    for name, mk, n in (("points_1500x1500_ws3", point_case, 1500), ("lines_200x200_ws3", line_case, 200)):
        c = mk(11, n, n, G.GRID_COLS, G.GRID_ROWS)
        for _ in range(5):
            ctx.match_grid(window=W, nnr=0.75, mutual=True, **c)
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            m, k = ctx.match_grid(window=W, nnr=0.75, mutual=True, **c)
            ts.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        rm, rk = O.match_grid(window=W, nnr=0.75, mutual=True, **c)
        cpu = time.perf_counter() - t0
        assert np.array_equal(m, rm) and k == rk
        tb = []
        for _ in range(20):
            t0 = time.perf_counter()
            ctx.match(c["d1"], c["d2"], 0.75, True)
            tb.append(time.perf_counter() - t0)
        cen = np.asarray(c["centres"], np.int32).reshape(-1, 2)
        ts = np.array(ts) * 1e6
        out[name] = {"pairs": G.pair_count(cen, c["cell_start"], G.GRID_COLS, G.GRID_ROWS, W), "matches": int(k),
                     "gpu_call_us_median": float(np.median(ts)), "gpu_call_us_p10": float(np.percentile(ts, 10)),
                     "gpu_call_us_p90": float(np.percentile(ts, 90)), "cpu_oracle_1thread_us": cpu * 1e6,
                     "bf_match_call_us_median": float(np.median(tb) * 1e6)}

    # batch: B frame pairs, each a point problem (1500) and a line problem (200), device-resident
    for B in (64, 1024):
        keep, probs, pairs = [], [], 0

        def up(a, dt):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
            keep.append(t)
            return t

        base = [(point_case(100 + s, 1500, 1500, G.GRID_COLS, G.GRID_ROWS), False) for s in range(4)] + \
               [(line_case(200 + s, 200, 200, G.GRID_COLS, G.GRID_ROWS), True) for s in range(4)]
        ups = []
        for c, lines in base:
            cen = np.asarray(c["centres"], np.int32).reshape(c["d1"].shape[0], -1, 2)
            u = dict(d1=up(c["d1"], np.uint8), d2=up(c["d2"], np.uint8), cen=up(cen, np.int32),
                     cs=up(c["cell_start"], np.int32), it=up(c["cell_items"], np.int32), nc=cen.shape[1],
                     cap=G.store_capacity(cen, c["cell_start"], G.GRID_COLS, G.GRID_ROWS, W),
                     pairs=G.pair_count(cen, c["cell_start"], G.GRID_COLS, G.GRID_ROWS, W), lines=lines)
            if lines:
                u.update(a=up(c["dir1"], np.float64), b=up(c["dir2"], np.float64))
            ups.append(u)
        for b in range(B):
            for kind in (0, 4):
                u = ups[kind + b % 4]
                n1, n2 = u["d1"].shape[0], u["d2"].shape[0]
                o, cnt = torch.empty(n1, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev)
                keep += [o, cnt]
                q = dict(d1=u["d1"].data_ptr(), d2=u["d2"].data_ptr(), centres1=u["cen"].data_ptr(),
                         cell_start=u["cs"].data_ptr(), cell_items=u["it"].data_ptr(), n1=n1, n2=n2, n_centres=u["nc"],
                         grid_cols=G.GRID_COLS, grid_rows=G.GRID_ROWS, n_items=u["it"].shape[0], window=W, nnr=0.75, mutual=True,
                         pair_capacity=u["cap"], matches_12=o.data_ptr(), n_matches=cnt.data_ptr())
                if u["lines"]:
                    q.update(dir1=u["a"].data_ptr(), dir2=u["b"].data_ptr(), sim_th=0.75)
                probs.append(q)
                pairs += u["pairs"]
        plan = plslam_amd.GridPlan(ctx, probs)
        s = torch.cuda.Stream(device=dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s):
            for _ in range(3):
                plan.run(s.cuda_stream)
            e0.record(s)
            for _ in range(10):
                plan.run(s.cuda_stream)
            e1.record(s)
        s.synchronize()
        ms = e0.elapsed_time(e1) / 10
        assert plan.overflows(s.cuda_stream) == 0
        out[f"plan_{B}_frame_pairs"] = {"problems": len(probs), "candidate_pairs": pairs, "ms_per_launch": ms,
                                        "frame_pairs_per_s": B / ms * 1e3, "candidate_pairs_per_s": pairs / ms * 1e3}
        plan.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
