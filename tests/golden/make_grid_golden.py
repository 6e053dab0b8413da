"""Generates tests/golden/grid_golden.npz.  Run from the repo root:  python tests/golden/make_grid_golden.py

The windowed matcher StVO::matchGrid lives in the un-vendored stvo-pl (call sites src/mapHandler.cpp:271,418,591,706);
the reference holds no vectors for it.  The match tables below are produced by the order-free numpy formulation
(oracle.oracle.np_match_grid: prefix minima per column) -- independent of both the C oracle's sequential loop and the
HIP kernel -- so the committed file pins all three to one another.
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from test_match_grid_cpu import line_case, point_case  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    out = {}
    names = []
    specs = [("points", 220, 200, 16, 12, (2, 2, 2, 2), False), ("points", 150, 180, 8, 6, (3, 0, 0, 0), True),
             ("lines", 90, 80, 16, 12, (2, 2, 2, 2), False), ("lines", 60, 70, 8, 6, (1, 1, 1, 1), True),
             ("points", 64, 3, 2, 2, (1, 1, 1, 1), True)]
    for k, (kind, n1, n2, cols, rows, w, ties) in enumerate(specs):
        c = (point_case if kind == "points" else line_case)(9000 + k, n1, n2, cols, rows, ties)
        name = f"c{k}"
        names.append(name)
        for key in ("centres", "d1", "cell_start", "cell_items", "d2", "dir1", "dir2"):
            if key in c:
                out[f"{name}_{key}"] = np.asarray(c[key])
        out[f"{name}_meta"] = np.array([cols, rows, *w, 1 if kind == "lines" else 0], np.int32)
        for mutual in (0, 1):
            for nnr in (0.75, 0.9):
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    m, n = O.np_match_grid(window=w, nnr=nnr, mutual=bool(mutual), **c)
                out[f"{name}_m{mutual}_r{int(nnr * 100)}"] = m
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "grid_golden.npz"), **out)
    print("wrote grid_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
