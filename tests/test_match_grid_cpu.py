"""CPU tests of the windowed matcher's test infrastructure (StVO::matchGrid; call sites
src/mapHandler.cpp:271,418,591,706 -- the function itself is in the un-vendored stvo-pl, [RECALL]):
  * the oracle's literal sequential restatement (plo_match_grid) against the order-free formulation the
    device kernel uses (np_match_grid): the loop-carried `if (d < distances[i2]) ... else continue;` equals
    "i1 is the first row among those at least as close to i2";
  * the host-side grid helpers (plslam_amd/grid.py) against the oracle's.
"""
import warnings

import numpy as np
import pytest

from plslam_amd import grid as G
from plslam_amd import synth


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def point_case(seed, n1, n2, cols, rows, ties=False):
    f = synth.grid_frame_pair(_rng(seed), n1, n2, ties=ties)
    sx, sy = cols / f["width"], rows / f["height"]
    c1 = G.to_cells(f["px1"] * [sx, sy])
    cs, items = G.fill_points(G.to_cells(f["px2"] * [sx, sy]), cols, rows)
    return dict(centres=c1, d1=f["d1"], cell_start=cs, cell_items=items, cols=cols, rows=rows, d2=f["d2"])


def line_case(seed, n1, n2, cols, rows, ties=False):
    f = synth.grid_frame_pair(_rng(seed), n1, n2, ties=ties, lines=True)
    sc = np.array([cols / f["width"], rows / f["height"]] * 2)
    s1, s2 = f["seg1"] * sc, f["seg2"] * sc
    c1 = G.to_cells(s1).reshape(-1, 2, 2)
    cs, items = G.fill_lines(s2, cols, rows)
    # upstream derives the query direction from the INTEGER end points (zero vectors -> NaN -> never skipped)
    d1 = G.directions(c1.reshape(-1, 4).astype(np.float64))
    return dict(centres=c1, d1=f["d1"], cell_start=cs, cell_items=items, cols=cols, rows=rows, d2=f["d2"],
                dir1=d1, dir2=G.directions(s2), sim_th=0.75)


@pytest.mark.parametrize("mutual", [False, True])
@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("kind", ["points", "lines"])
def test_sequential_oracle_equals_order_free_form(oracle, kind, ties, mutual):
    mk = point_case if kind == "points" else line_case
    for seed, (n1, n2, cols, rows, w) in enumerate([(150, 140, 16, 12, (2, 2, 2, 2)), (90, 120, 8, 6, (3, 0, 0, 0)),
                                                    (40, 30, 2, 2, (1, 1, 1, 1)), (60, 1, 4, 4, (4, 4, 4, 4))]):
        c = mk(seed + 10 * ties, n1, n2, cols, rows, ties)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = oracle.match_grid(window=w, nnr=0.75, mutual=mutual, **c)
            b = oracle.np_match_grid(window=w, nnr=0.75, mutual=mutual, **c)
        np.testing.assert_array_equal(a[0], b[0])
        assert a[1] == b[1] == int((a[0] >= 0).sum())


def test_sequential_dependence_is_real(oracle):
    """Row 0 and row 1 both see item 0; row 1 is closer.  With mutual the later, closer row takes it and row 0,
    which claimed it first, is dropped by the final check; a row that is NOT closer is skipped at the candidate
    (so its second candidate becomes its lone -- accepted -- best)."""
    z = np.zeros((1, 32), np.uint8)
    d2 = np.concatenate([z, z]); d2[1, 0] = 0xFF                      # items 0, 1
    d1 = np.zeros((2, 32), np.uint8); d1[0, 1] = 0x01                 # row 0: d=(1, 9); row 1: d=(0, 8)
    cs, items = G.fill_points([[0, 0], [0, 0]], 1, 1)
    c1 = np.zeros((2, 1, 2), np.int32)
    m, n = oracle.match_grid(c1, d1, cs, items, 1, 1, d2, (0, 0, 0, 0), 0.75, True)
    assert m.tolist() == [-1, 0] and n == 1
    # swap the rows: the first row is the closer one; the second row's candidate 0 is skipped (1 !< 0), so is
    # candidate 1 (9 !< 8): no live candidate -> no match for it
    m, n = oracle.match_grid(c1, d1[::-1].copy(), cs, items, 1, 1, d2, (0, 0, 0, 0), 0.75, True)
    assert m.tolist() == [0, -1] and n == 1
    # without mutual each row simply takes its nearest (ratio 0/8 and 1/9 pass)
    m, n = oracle.match_grid(c1, d1, cs, items, 1, 1, d2, (0, 0, 0, 0), 0.75, False)
    assert m.tolist() == [0, 0] and n == 2


def test_lone_candidate_is_accepted_and_tie_goes_to_lowest_index(oracle):
    d2 = synth.random_desc(_rng(1), 3)
    d2[2] = d2[1]                                                     # items 1 and 2 identical
    d1 = d2[[0, 1]].copy()
    cs, items = G.fill_points([[0, 0], [3, 3], [3, 3]], 4, 4)
    m, n = oracle.match_grid([[0, 0], [3, 3]], d1, cs, items, 4, 4, d2, (0, 0, 0, 0), 0.75, False)
    # row 0: one candidate (best_d2 = INT_MAX) -> accepted; row 1: d = (0, 0): 0 < 0 * 0.75 fails
    assert m.tolist() == [0, -1] and n == 1
    d1[1, 0] ^= 1                                                     # d = (1, 1) -> still rejected by the ratio
    assert oracle.match_grid([[0, 0], [3, 3]], d1, cs, items, 4, 4, d2, (0, 0, 0, 0), 0.75, False)[0].tolist() == [0, -1]
    # with nnr > 1 the tie is accepted and must go to the LOWEST index (the definition this repo adds)
    assert oracle.match_grid([[0, 0], [3, 3]], d1, cs, items, 4, 4, d2, (0, 0, 0, 0), 1.5, False)[0].tolist() == [0, 1]


def test_grid_helpers_match_the_oracle(oracle):
    r = _rng(3)
    xy = np.stack([r.integers(-2, 70, 500), r.integers(-2, 52, 500)], 1)
    a, b = G.fill_points(xy), oracle.grid_fill_points(xy, G.GRID_COLS, G.GRID_ROWS)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    for _ in range(200):
        s = r.uniform(-3, 70, 4)
        assert G.line_coords(*s) == [tuple(v) for v in oracle.get_line_coords(*s).tolist()]
    seg = r.uniform(-2, 66, (80, 4))
    a, b = G.fill_lines(seg), oracle.grid_fill_lines(seg, G.GRID_COLS, G.GRID_ROWS)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])
    with np.errstate(all="ignore"):
        np.testing.assert_array_equal(G.directions(seg), oracle.normalize2(np.stack([seg[:, 2] - seg[:, 0], seg[:, 3] - seg[:, 1]], 1)))
    assert np.isnan(G.directions([[1, 1, 1, 1]])).all()


def test_pair_count_is_the_candidate_enumeration(oracle):
    c = point_case(5, 300, 280, 16, 12)
    for w in [(2, 2, 2, 2), (3, 0, 0, 0), (0, 0, 0, 5), (20, 20, 20, 20)]:
        n = 0
        cs = c["cell_start"]
        for x, y in c["centres"]:
            for x_ in range(max(0, x - w[0]), min(16, x + w[1] + 1)):
                lo, hi = max(0, y - w[2]), min(12, y + w[3] + 1)
                if lo < hi:
                    n += cs[x_ * 12 + hi] - cs[x_ * 12 + lo]
        assert G.pair_count(c["centres"], cs, 16, 12, w) == n
        assert G.store_capacity(c["centres"], cs, 16, 12, w) >= n and G.store_capacity(c["centres"], cs, 16, 12, w, False) == 0


GRID_GOLD = __import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "grid_golden.npz")


def golden_cases():
    g = np.load(GRID_GOLD)
    for name in g["names"]:
        cols, rows, w0, w1, w2, w3, lines = [int(v) for v in g[f"{name}_meta"]]
        c = dict(centres=g[f"{name}_centres"], d1=g[f"{name}_d1"], cell_start=g[f"{name}_cell_start"],
                 cell_items=g[f"{name}_cell_items"], cols=cols, rows=rows, d2=g[f"{name}_d2"])
        if lines:
            c.update(dir1=g[f"{name}_dir1"], dir2=g[f"{name}_dir2"], sim_th=0.75)
        for mutual in (0, 1):
            for nnr in (0.75, 0.9):
                yield c, (w0, w1, w2, w3), nnr, bool(mutual), g[f"{name}_m{mutual}_r{int(nnr * 100)}"]


def test_oracle_reproduces_the_committed_grid_goldens(oracle):
    """tests/golden/grid_golden.npz (made by the order-free numpy form, tests/golden/make_grid_golden.py) vs the C
    oracle's literal sequential loop."""
    k = 0
    for c, w, nnr, mutual, want in golden_cases():
        m, n = oracle.match_grid(window=w, nnr=nnr, mutual=mutual, **c)
        np.testing.assert_array_equal(m, want)
        assert n == int((want >= 0).sum())
        k += int((want >= 0).sum())
    assert k > 300


def test_hypothesis_sequential_vs_order_free(oracle):
    """Property-based: arbitrary small grids, windows, centres (also far outside the grid), items (also out of range
    and repeated across cells), descriptors drawn from 3 patterns (ties everywhere): the literal sequential loop and
    the order-free form agree."""
    from hypothesis import given, settings, strategies as st

    pats = np.array([[0] * 32, [0xFF] + [0] * 31, [0x0F] * 32], np.uint8)

    @settings(max_examples=150, deadline=None)
    @given(st.integers(1, 4), st.integers(1, 4), st.integers(0, 12), st.integers(0, 10), st.integers(1, 2),
           st.tuples(*[st.integers(0, 3)] * 4), st.booleans(), st.sampled_from([0.5, 0.75, 1.0, 1.5]), st.integers(0, 2 ** 31))
    def run(cols, rows, n1, n2, nc, w, mutual, nnr, seed):
        r = _rng(seed)
        d1, d2 = pats[r.integers(0, 3, n1)], pats[r.integers(0, 3, n2)]
        cen = np.stack([r.integers(-3, cols + 3, (n1, nc)), r.integers(-3, rows + 3, (n1, nc))], 2).astype(np.int32)
        # a hand-made CSR grid: every cell gets a random multiset of item ids, some outside [0, n2)
        lens = r.integers(0, 4, cols * rows)
        cs = np.zeros(cols * rows + 1, np.int32)
        np.cumsum(lens, out=cs[1:])
        items = r.integers(-1, n2 + 2, int(cs[-1])).astype(np.int32)
        kw = dict(centres=cen, d1=d1.reshape(-1, 32), cell_start=cs, cell_items=items, cols=cols, rows=rows,
                  d2=d2.reshape(-1, 32), window=w, nnr=nnr, mutual=mutual)
        a, b = oracle.match_grid(**kw), oracle.np_match_grid(**kw)
        assert a[1] == b[1] and np.array_equal(a[0], b[0])

    run()


def test_pair_capacity_helpers_of_the_abi():
    """plslam_grid_pair_capacity = the documented store size (the Python restatement in plslam_amd.grid agrees);
    plslam_grid_pair_capacity_bound never falls below it.  Pure host code: runs without a device."""
    from plslam_amd import capi
    r = _rng(41)
    for it in range(40):
        lines = it % 2 == 1
        n1, n2 = int(r.integers(1, 2600)), int(r.integers(1, 1800))
        cols, rows = [(1, 1), (3, 2), (16, 12), (64, 48)][it % 4]
        w = tuple(int(x) for x in r.integers(0, 5, 4)) if it % 5 else (2 ** 31 - 1,) * 4
        c = (line_case if lines else point_case)(9000 + it, n1, n2, cols, rows)
        cen = np.asarray(c["centres"], np.int32).reshape(n1, -1, 2)
        exact = capi.grid_pair_capacity(cen, c["cell_start"], cols, rows, w)
        assert exact == G.store_capacity(cen, c["cell_start"], cols, rows, w)
        bound = capi.grid_pair_capacity(cen, c["cell_start"], cols, rows, w, bound=True)
        assert bound >= exact, (it, bound, exact)
        assert capi.grid_pair_capacity(cen, c["cell_start"], cols, rows, w, mutual=False) == 0
        assert capi.grid_pair_capacity(cen, c["cell_start"], cols, rows, w, mutual=False, bound=True) == 0
