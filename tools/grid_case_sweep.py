import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import numpy as np
import plslam_amd
from oracle import oracle as O
from test_match_grid_cpu import line_case, point_case
import test_gpu_match_grid as T
ctx = plslam_amd.Context(0)
r = T._rng(99)
bad = 0
for it in range(60):
    lines = it % 3 == 0
    n1 = int(r.choice([1, 2, 63, 64, 65, 255, 256, 257, 700, 1024, 1025, 2048, 2049, 4096, 4097, 8192]))
    n2 = int(r.choice([1, 2, 40, 333, 512, 1024, 1025, 1500, 2048]))
    cols, rows = [(1, 1), (2, 3), (7, 5), (16, 12), (64, 48)][it % 5]
    w = tuple(int(x) for x in r.integers(0, 4, 4)) if it % 4 else (cols, cols, rows, rows)
    nnr = float(r.choice([0.6, 0.75, 0.9, 1.5]))
    c = (line_case if lines else point_case)(7000 + it, n1, n2, cols, rows, ties=it % 2 == 1)
    got = ctx.match_grid(window=w, nnr=nnr, mutual=True, **c)
    ref = O.match_grid(window=w, nnr=nnr, mutual=True, **c)
    d = np.nonzero(got[0] != ref[0])[0]
    if len(d) or got[1] != ref[1]:
        bad += 1
        print("case", it, n1, n2, (cols, rows), w, nnr, "rows differing", d[:8], got[0][d[:8]], ref[0][d[:8]], got[1], ref[1])
print("bad cases", bad)
