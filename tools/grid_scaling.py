#!/usr/bin/env python3
"""K14 batch scaling: ms per launch of a device-resident plan of N identical-size problems (points 1500x1500 or lines
200x200, window 3) for N = 128 ... 4096: does a problem cost the same when all 256 CUs are busy?"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import plslam_amd  # noqa: E402
from plslam_amd import grid as G  # noqa: E402
from test_match_grid_cpu import line_case, point_case  # noqa: E402


def main():
    import torch
    ctx = plslam_amd.Context(0)
    dev = torch.device("cuda", 0)
    W = (3, 3, 3, 3)
    out = {}
    # GRID_POINTS_N / GRID_BATCHES: other point-problem sizes / launch sizes for experiments (e.g. "1000", "1,128,256")
    n_points = int(os.environ.get("GRID_POINTS_N", "1500"))
    batches = tuple(int(x) for x in os.environ.get("GRID_BATCHES", "128,256,1024,4096").split(","))
    for kind, mk, n in (("points", point_case, n_points), ("lines", line_case, 200)):
        keep = []

        def up(a, dt):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
            keep.append(t)
            return t

        ups = []
        NV = int(os.environ.get("GRID_VARIANTS", "4"))          # distinct input sets the problems cycle over
        base_cases = [mk(100 + s, n, n, G.GRID_COLS, G.GRID_ROWS) for s in range(4)]
        for s in range(NV):
            c = base_cases[s % 4]                                # same content, own device buffers
            cen = np.asarray(c["centres"], np.int32).reshape(n, -1, 2)
            u = dict(d1=up(c["d1"], np.uint8), d2=up(c["d2"], np.uint8), cen=up(cen, np.int32), cs=up(c["cell_start"], np.int32),
                     it=up(c["cell_items"], np.int32), nc=cen.shape[1],
                     cap=G.store_capacity(cen, c["cell_start"], G.GRID_COLS, G.GRID_ROWS, W))
            if "dir1" in c:
                u.update(a=up(c["dir1"], np.float64), b=up(c["dir2"], np.float64))
            ups.append(u)
        for B in batches:
            probs, outs = [], []
            for b in range(B):
                u = ups[b % NV]
                o, cnt = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev)
                outs += [o, cnt]
                q = dict(d1=u["d1"].data_ptr(), d2=u["d2"].data_ptr(), centres1=u["cen"].data_ptr(), cell_start=u["cs"].data_ptr(),
                         cell_items=u["it"].data_ptr(), n1=n, n2=n, n_centres=u["nc"], grid_cols=G.GRID_COLS,
                         grid_rows=G.GRID_ROWS, n_items=u["it"].shape[0], window=W, nnr=0.75, mutual=True,
                         pair_capacity=u["cap"], matches_12=o.data_ptr(), n_matches=cnt.data_ptr())
                if "a" in u:
                    q.update(dir1=u["a"].data_ptr(), dir2=u["b"].data_ptr(), sim_th=0.75)
                probs.append(q)
            plan = plslam_amd.GridPlan(ctx, probs)
            s = torch.cuda.Stream(device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(s):
                for _ in range(2):
                    plan.run(s.cuda_stream)
                e0.record(s)
                for _ in range(5):
                    plan.run(s.cuda_stream)
                e1.record(s)
            s.synchronize()
            ms = e0.elapsed_time(e1) / 5
            out[f"{kind}_{B}_v{NV}"] = {"ms_per_launch": ms, "problems_per_s": B / ms * 1e3, "us_per_problem_per_cu": ms * 1e3 / max(B / 256, 1)}
            plan.close()
            del outs
    print(json.dumps(out))


if __name__ == "__main__":
    main()
