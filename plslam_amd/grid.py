"""Host-side mirror of what the reference's CALLERS do around StVO::matchGrid: turning pixel positions into
grid cells and filling the GridStructure (src/mapHandler.cpp:252-269 points, :381-416 lines, :580-589, :683-704).
Plumbing for tests and the bench -- the matcher itself is plslam_match_grid (include/plslam_hip.h); the C++ form
of the same helpers is plslam_amd/host/stvo_match.hpp.

GridStructure(rows, cols) of stvo-pl ([RECALL]) is addressed grid[x][y] with 0 <= x < cols, 0 <= y < rows; the CSR
form used by the C ABI gives cell (x, y) the id x*rows + y and keeps each cell's push_back order.
"""
from __future__ import annotations

import numpy as np

GRID_ROWS, GRID_COLS = 48, 64          # stvo-pl gridStructure.h [RECALL]


def to_cells(xy):
    """double -> int as std::pair<int,int>(double, double) does it: truncation toward zero."""
    return np.trunc(np.asarray(xy, np.float64)).astype(np.int32)


def fill_points(cells, cols=GRID_COLS, rows=GRID_ROWS):
    """grid.at(x, y).push_back(idx) for idx = 0..n-1; out-of-range cells go to a dummy list (dropped)."""
    c = np.asarray(cells, np.int32).reshape(-1, 2)
    ok = (c[:, 0] >= 0) & (c[:, 0] < cols) & (c[:, 1] >= 0) & (c[:, 1] < rows)
    ids = c[ok, 0].astype(np.int64) * rows + c[ok, 1]
    order = np.argsort(ids, kind="stable")
    start = np.zeros(cols * rows + 1, np.int32)
    np.cumsum(np.bincount(ids, minlength=cols * rows), out=start[1:])
    return start, np.nonzero(ok)[0][order].astype(np.int32)


def line_coords(x1, y1, x2, y2):
    """getLineCoords (stvo-pl gridStructure.cpp, [RECALL]): Bresenham cells, the last x excluded."""
    steep = abs(y2 - y1) > abs(x2 - x1)
    if steep:
        x1, y1, x2, y2 = y1, x1, y2, x2
    if x1 > x2:
        x1, x2, y1, y2 = x2, x1, y2, y1
    dx, dy = x2 - x1, abs(y2 - y1)
    error = dx / 2.0
    ystep = 1 if y1 < y2 else -1
    y, out = int(y1), []
    for x in range(int(x1), int(x2)):
        out.append((y, x) if steep else (x, y))
        error -= dy
        if error < 0:
            y += ystep
            error += dx
    return out


def fill_lines(segments, cols=GRID_COLS, rows=GRID_ROWS):
    """Every Bresenham cell of segment idx receives idx, idx ascending (src/mapHandler.cpp:398-411).
    segments: n x 4 (x1, y1, x2, y2) in grid units (pixels * inv_width / inv_height)."""
    seg = np.asarray(segments, np.float64).reshape(-1, 4)
    ids, items = [], []
    for idx, (x1, y1, x2, y2) in enumerate(seg):
        for x, y in line_coords(x1, y1, x2, y2):
            if 0 <= x < cols and 0 <= y < rows:
                ids.append(x * rows + y)
                items.append(idx)
    ids, items = np.asarray(ids, np.int64), np.asarray(items, np.int32)
    order = np.argsort(ids, kind="stable")
    start = np.zeros(cols * rows + 1, np.int32)
    np.cumsum(np.bincount(ids, minlength=cols * rows), out=start[1:])
    return start, items[order]


def directions(segments):
    """normalize() of the callers' direction vectors (:402-404); a zero-length segment gives NaN."""
    seg = np.asarray(segments, np.float64).reshape(-1, 4)
    v = np.stack([seg[:, 2] - seg[:, 0], seg[:, 3] - seg[:, 1]], 1)
    with np.errstate(invalid="ignore", divide="ignore"):
        return v / np.sqrt(v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1])[:, None]


def row_item_counts(centres, cell_start, cols, rows, window):
    """Grid items inside the window(s) of every row (duplicates / out-of-range items included). centres: n1 x c x 2."""
    c = np.asarray(centres, np.int64)
    c = c.reshape(c.shape[0], -1, 2)
    cs = np.asarray(cell_start, np.int64)
    w = [int(v) for v in window]
    cnt = np.zeros(c.shape[0], np.int64)
    lo = np.clip(c[:, :, 1] - w[2], 0, rows)
    hi = np.clip(c[:, :, 1] + w[3] + 1, 0, rows)
    for dx in range(-min(w[0], cols), min(w[1], cols) + 1):
        x = c[:, :, 0] + dx
        ok = (x >= 0) & (x < cols) & (lo < hi)
        xs = np.where(ok, x, 0)
        cnt += np.where(ok, cs[xs * rows + hi] - cs[xs * rows + lo], 0).sum(axis=1)
    return cnt


def pair_count(centres, cell_start, cols, rows, window):
    """(row, candidate) pairs of a problem, duplicates included."""
    return int(row_item_counts(centres, cell_start, cols, rows, window).sum())


def store_capacity(centres, cell_start, cols, rows, window, mutual=True):
    """plslam_grid_problem.pair_capacity: rows go in blocks of 1024; a block needs 1024 slots per candidate of its
    fullest row (only mutual problems store candidates)."""
    if not mutual:
        return 0
    cnt = row_item_counts(centres, cell_start, cols, rows, window)
    return int(sum(1024 * int(cnt[b:b + 1024].max()) for b in range(0, len(cnt), 1024)))
