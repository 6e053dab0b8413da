"""GPU parity tests for K14, the windowed matcher StVO::matchGrid (plslam_match_grid / plslam_grid_plan_*; call sites
src/mapHandler.cpp:271,418,591,706): match tables bit-exact against the oracle's literal sequential restatement --
points and lines, with and without Config::bestLRMatches(), tie-heavy descriptors, degenerate grids."""
import numpy as np
import pytest

import plslam_amd
from plslam_amd import grid as G
from plslam_amd import synth
from test_match_grid_cpu import line_case, point_case

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["dense_small", "dense_small_copied", "general", "general_in_place"], autouse=True)
def small_problem_kernel(request, ctx):
    """Every test of this module four times: with a lone problem of at most 256 x 256 rows on the one-workgroup dense kernel (round
    6, k_match_grid_dense: the default) reading its upload image where it lies in page-locked host memory (the default,
    zero_copy_kb = 64) or copied to the device first; and with that kernel off, i.e. on the general kernels (records / candidate
    list / passes) every other problem takes anyway -- behind the copy (their default) or reading the image in place."""
    ctx.set_option("grid_dense", 1 if request.param.startswith("dense_small") else 0)
    ctx.set_option("zero_copy_kb", {"dense_small": 64, "dense_small_copied": 0, "general": 64, "general_in_place": -1024}[request.param])
    yield request.param
    ctx.set_option("grid_dense", 1)
    ctx.set_option("zero_copy_kb", 64)


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def _same(ctx, oracle, c, w, nnr, mutual):
    got = ctx.match_grid(window=w, nnr=nnr, mutual=mutual, **c)
    ref = oracle.match_grid(window=w, nnr=nnr, mutual=mutual, **c)
    np.testing.assert_array_equal(got[0], ref[0])
    assert got[1] == ref[1]
    assert nnr > 1.0 or ref[1] == int((ref[0] >= 0).sum())           # (nnr > 1: upstream also counts rows without a live candidate)
    return ref


@pytest.mark.parametrize("mutual", [False, True])
@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("kind", ["points", "lines"])
def test_random_scenes_vs_oracle(ctx, oracle, kind, ties, mutual):
    mk = point_case if kind == "points" else line_case
    total = 0
    for seed, (n1, n2, cols, rows, w) in enumerate([(1500, 1500, 64, 48, (3, 3, 3, 3)), (800, 900, 64, 48, (5, 0, 0, 0)),
                                                    (300, 280, 16, 12, (2, 2, 2, 2)), (200, 150, 8, 6, (3, 0, 1, 0)),
                                                    (50, 40, 2, 2, (1, 1, 1, 1)), (2000, 3, 4, 4, (4, 4, 4, 4)),
                                                    (1, 1, 64, 48, (64, 64, 48, 48))]):
        c = mk(100 + seed + 10 * ties, n1, n2, cols, rows, ties)
        for nnr in (0.75, 0.9):
            total += _same(ctx, oracle, c, w, nnr, mutual)[1]
    assert total > 0


def test_many_small_random_problems_vs_oracle(ctx, oracle):
    """Sweep of shapes for the LDS-resident mutual path (candidates filtered and kept per wave, candidate-parallel record
    passes): coarse grids (long column lists, many duplicates of a line's item across cells), windows from one cell to the
    whole grid, ties, 1 .. 8192 rows on 256- and 1024-lane workgroups -- on both sides of the packed-candidate limit (row and
    column numbers of 23 bits together: 4096 x 2048, 8192 x 1024 fit, 4097 x 1025 and 8192 x 1500 do not)."""
    r = _rng(99)
    for it in range(60):
        lines = it % 3 == 0
        n1 = int(r.choice([1, 2, 63, 64, 65, 255, 256, 257, 700, 1024, 1025, 2048, 2049, 4096, 4097, 8192]))
        n2 = int(r.choice([1, 2, 40, 333, 512, 1024, 1025, 1500, 2048]))
        cols, rows = [(1, 1), (2, 3), (7, 5), (16, 12), (64, 48)][it % 5]
        w = tuple(int(x) for x in r.integers(0, 4, 4)) if it % 4 else (cols, cols, rows, rows)
        c = (line_case if lines else point_case)(7000 + it, n1, n2, cols, rows, ties=it % 2 == 1)
        _same(ctx, oracle, c, w, float(r.choice([0.6, 0.75, 0.9, 1.5])), True)


@pytest.mark.parametrize("kind", ["points", "lines"])
def test_tables_too_large_for_lds_run_on_global_scratch(ctx, oracle, kind):
    """A 400 x 100 grid (40 001 cell offsets) or 20 000 rows exceed the LDS budget of a workgroup: same code, tables in
    global memory."""
    mk = point_case if kind == "points" else line_case
    for seed, (n1, n2, cols, rows, w) in enumerate([(900, 800, 400, 100, (12, 12, 6, 6)), (20000, 600, 64, 48, (2, 2, 2, 2))]):
        c = mk(300 + seed, n1, n2, cols, rows)
        for mutual in (False, True):
            _same(ctx, oracle, c, w, 0.8, mutual)


def test_c2_frame_pair_finds_the_true_matches(ctx, oracle):
    """BASELINE config 2 shape: 1500 ORB rows per image, 64 x 48 grid, matching_f2f_ws = 3
    (config/config/config_kitti.yaml:59)."""
    c = point_case(7, 1500, 1500, G.GRID_COLS, G.GRID_ROWS)
    m, n = _same(ctx, oracle, c, (3, 3, 3, 3), 0.75, True)
    assert n > 300                                                   # ~70 % of the rows have a true match nearby
    # every reported match is inside the window of its row
    x2 = {int(i): k // G.GRID_ROWS for k in range(G.GRID_COLS * G.GRID_ROWS)
          for i in c["cell_items"][c["cell_start"][k]:c["cell_start"][k + 1]]}
    for i1 in np.nonzero(m >= 0)[0]:
        assert abs(x2[int(m[i1])] - int(c["centres"][i1][0])) <= 3


def test_everything_in_one_cell_long_column_lists(ctx, oracle):
    """All rows and all items share one cell: every column list has n1 entries (> one wave), the prefix-minimum
    bins see every distance value; this is brute force restricted by the sequential rule."""
    for ties in (False, True):
        f = synth.grid_frame_pair(_rng(21 + ties), 700, 300, ties=ties)
        cs, items = G.fill_points(np.zeros((300, 2), np.int32), 3, 3)
        c = dict(centres=np.ones((700, 1, 2), np.int32), d1=f["d1"], cell_start=cs, cell_items=items, cols=3, rows=3,
                 d2=f["d2"])
        for mutual in (False, True):
            _same(ctx, oracle, c, (1, 1, 1, 1), 0.9, mutual)


def test_extreme_distances_and_lone_candidates(ctx, oracle):
    z = np.zeros((4, 32), np.uint8)
    o = np.full((4, 32), 0xFF, np.uint8)
    cs, items = G.fill_points([[0, 0], [0, 0], [1, 1], [1, 1]], 2, 2)
    for d1, d2 in ((z, o), (o, z), (z, z), (o, o)):                  # distances 256 / 0 only
        for mutual in (False, True):
            for nnr in (0.75, 2.0):
                _same(ctx, oracle, dict(centres=[[0, 0], [1, 1], [0, 1], [5, 5]], d1=d1, cell_start=cs, cell_items=items,
                                        cols=2, rows=2, d2=d2), (0, 0, 0, 0), nnr, mutual)
    # a lone candidate passes the ratio test (best_d2 = INT_MAX)
    r = _rng(3)
    d2 = synth.random_desc(r, 1)
    d1 = synth.random_desc(r, 1)
    cs, items = G.fill_points([[2, 2]], 4, 4)
    m, n = ctx.match_grid([[2, 2]], d1, cs, items, 4, 4, d2, (0, 0, 0, 0), 0.75, True)
    assert m.tolist() == [0] and n == 1


def test_sequential_rule_examples(ctx, oracle):
    z = np.zeros((1, 32), np.uint8)
    d2 = np.concatenate([z, z]); d2[1, 0] = 0xFF
    d1 = np.zeros((2, 32), np.uint8); d1[0, 1] = 0x01
    cs, items = G.fill_points([[0, 0], [0, 0]], 1, 1)
    c1 = np.zeros((2, 1, 2), np.int32)
    assert ctx.match_grid(c1, d1, cs, items, 1, 1, d2, (0, 0, 0, 0), 0.75, True)[0].tolist() == [-1, 0]
    assert ctx.match_grid(c1, d1[::-1].copy(), cs, items, 1, 1, d2, (0, 0, 0, 0), 0.75, True)[0].tolist() == [0, -1]
    assert ctx.match_grid(c1, d1, cs, items, 1, 1, d2, (0, 0, 0, 0), 0.75, False)[0].tolist() == [0, 0]


def test_empty_and_out_of_range_inputs(ctx, oracle):
    r = _rng(5)
    d1, d2 = synth.random_desc(r, 5), synth.random_desc(r, 4)
    cs0 = np.zeros(13, np.int32)
    none = np.zeros(0, np.int32)
    m, n = ctx.match_grid(np.zeros((5, 1, 2), np.int32), d1, cs0, none, 4, 3, d2, (1, 1, 1, 1), 0.75, True)
    assert (m == -1).all() and n == 0                                # empty grid
    m, n = ctx.match_grid(np.zeros((0, 1, 2), np.int32), d1[:0], cs0, none, 4, 3, d2, (1, 1, 1, 1), 0.75, True)
    assert m.shape == (0,) and n == 0                                # no rows
    cs, items = G.fill_points([[0, 0], [1, 1], [2, 2], [3, 2]], 4, 3)
    m, n = ctx.match_grid(np.zeros((5, 1, 2), np.int32), d1, cs, items, 4, 3, d2[:0], (1, 1, 1, 1), 0.75, True)
    assert (m == -1).all() and n == 0                                # desc2 empty: every item is out of range
    # items outside [0, n2) are skipped (`if (i2 < 0 || i2 >= desc2.rows) continue;`); far-away and negative centres
    items2 = np.array([0, 7, -3, 3], np.int32)
    cen = np.array([[0, 0], [1, 1], [-2147483648, 2147483647], [2147483647, -2147483648], [3, 2]], np.int32)
    c = dict(centres=cen, d1=d1, cell_start=cs, cell_items=items2, cols=4, rows=3, d2=d2)
    for w in ((1, 1, 1, 1), (0, 0, 0, 0), (2147483647, 2147483647, 2147483647, 2147483647)):
        _same(ctx, oracle, c, w, 0.9, True)


def test_bad_arguments_are_refused(ctx):
    r = _rng(6)
    d = synth.random_desc(r, 4)
    cs, items = G.fill_points([[0, 0]] * 4, 2, 2)
    ok = dict(centres=np.zeros((4, 1, 2), np.int32), d1=d, cell_start=cs, cell_items=items, cols=2, rows=2, d2=d,
              nnr=0.75)
    with pytest.raises(plslam_amd.PlslamError):
        ctx.match_grid(window=(1, -1, 0, 0), **ok)
    bad = dict(ok, cell_start=cs[::-1].copy())
    with pytest.raises(plslam_amd.PlslamError):
        ctx.match_grid(window=(1, 1, 1, 1), **bad)
    with pytest.raises(plslam_amd.PlslamError):
        ctx.match_grid(window=(1, 1, 1, 1), **dict(ok, cols=0))


def test_host_rows_need_no_alignment(ctx, oracle):
    """Host descriptor rows are staged into aligned device memory: a view at an odd byte offset (a cv::Mat ROI) is
    accepted and gives the same table."""
    c = point_case(77, 300, 260, 16, 12)
    ref = _same(ctx, oracle, c, (2, 2, 2, 2), 0.8, True)

    def odd(a):
        buf = np.zeros(a.size + 16, np.uint8)
        off = 3 + (-buf.ctypes.data) % 16                       # 3 bytes past a 16-byte boundary
        v = buf[off:off + a.size].reshape(a.shape)
        v[:] = a
        assert v.ctypes.data % 16 == 3 and v.flags.c_contiguous
        return v

    got = ctx.match_grid(window=(2, 2, 2, 2), nnr=0.8, mutual=True, **dict(c, d1=odd(c["d1"]), d2=odd(c["d2"])))
    np.testing.assert_array_equal(got[0], ref[0])
    assert got[1] == ref[1]


def test_plan_batch_device_resident_and_overflow(ctx, oracle):
    """64 problems of mixed kind in ONE launch through plslam_grid_plan_*; a problem whose pair_capacity is too
    small reports an overflow, matches nothing and leaves the others intact.  (Problem 11 has more rows than the
    candidate words of the LDS-resident form can name -- row and column numbers of 23 bits together, at most 14 for the row --:
    such a problem keeps its candidates in the global store alone; smaller ones use it only for what the LDS cannot hold.)"""
    import torch
    dev = torch.device("cuda", ctx.device)
    keep, probs, refs = [], [], []

    def up(a, dt):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(dev)
        keep.append(t)
        return t

    for b in range(64):
        lines = b % 3 == 0
        n1, n2 = 50 + 37 * (b % 7), 40 + 29 * (b % 5)
        if b == 11:
            n1 = 17000
        c = (line_case if lines else point_case)(500 + b, n1, n2, 16, 12, ties=b % 2 == 1)
        w = (2, 2, 2, 2) if b % 4 else (4, 0, 0, 0)
        mutual = b % 5 != 0
        refs.append(oracle.match_grid(window=w, nnr=0.8, mutual=mutual, **c))
        cen = np.asarray(c["centres"], np.int32).reshape(n1, -1, 2)
        cap = G.store_capacity(cen, c["cell_start"], 16, 12, w, mutual)
        out, cnt = torch.full((n1,), -7, dtype=torch.int32, device=dev), torch.full((1,), -7, dtype=torch.int32, device=dev)
        keep += [out, cnt]
        q = dict(d1=up(c["d1"], np.uint8).data_ptr(), d2=up(c["d2"], np.uint8).data_ptr(),
                 centres1=up(cen, np.int32).data_ptr(), cell_start=up(c["cell_start"], np.int32).data_ptr(),
                 cell_items=up(c["cell_items"] if len(c["cell_items"]) else np.zeros(1), np.int32).data_ptr(),
                 n1=n1, n2=n2, n_centres=cen.shape[1], grid_cols=16, grid_rows=12, n_items=len(c["cell_items"]), window=w, nnr=0.8, mutual=mutual,
                 pair_capacity=cap if b != 11 else 0, matches_12=out.data_ptr(), n_matches=cnt.data_ptr())
        if lines:
            q.update(dir1=up(c["dir1"], np.float64).data_ptr(), dir2=up(c["dir2"], np.float64).data_ptr(), sim_th=c["sim_th"])
        probs.append((q, out, cnt))
    plan = plslam_amd.GridPlan(ctx, [p[0] for p in probs])
    stream = torch.cuda.Stream(device=dev)
    for rep in range(2):                                             # a plan is re-runnable
        plan.run(stream.cuda_stream)
        assert plan.overflows(stream.cuda_stream) == 1
        for b, ((q, out, cnt), ref) in enumerate(zip(probs, refs)):
            if b == 11:                                               # a mutual problem (only those store candidates)
                assert (out.cpu().numpy() == -1).all() and int(cnt.item()) == -1
                continue
            np.testing.assert_array_equal(out.cpu().numpy(), ref[0], err_msg=f"problem {b}")
            assert int(cnt.item()) == ref[1]
    plan.close()


def test_committed_grid_goldens(ctx):
    """GPU vs the committed fixtures tests/golden/grid_golden.npz (no oracle involved at run time)."""
    from test_match_grid_cpu import golden_cases
    for c, w, nnr, mutual, want in golden_cases():
        m, n = ctx.match_grid(window=w, nnr=nnr, mutual=mutual, **c)
        np.testing.assert_array_equal(m, want)
        assert n == int((want >= 0).sum())


@pytest.mark.parametrize("n1,n2", [(1500, 1500), (4000, 4000)])
def test_full_window_without_mutual_is_brute_force(ctx, n1, n2):
    """Size-independent property at BASELINE sizes (C2 / C5 rows), no oracle involved: with a window that covers the
    whole grid and bestLRMatches off, every row's candidates are ALL items, so matchGrid must return what the
    brute-force matcher (StVO::match without the mutual check, a different kernel family) returns -- the ratio tests
    `d0 < d1 * 0.75` agree in fp32 and fp64 for every integer pair."""
    f = synth.grid_frame_pair(_rng(n1), n1, n2)
    sc = [G.GRID_COLS / f["width"], G.GRID_ROWS / f["height"]]
    cells2 = np.clip(G.to_cells(f["px2"] * sc), 0, [G.GRID_COLS - 1, G.GRID_ROWS - 1])   # every item inside the grid
    cs, items = G.fill_points(cells2)
    c = dict(centres=G.to_cells(f["px1"] * sc), d1=f["d1"], cell_start=cs, cell_items=items, cols=G.GRID_COLS,
             rows=G.GRID_ROWS, d2=f["d2"])
    m, n = ctx.match_grid(window=(G.GRID_COLS + 8, G.GRID_COLS + 8, G.GRID_ROWS + 8, G.GRID_ROWS + 8), nnr=0.75, mutual=False, **c)
    bf, nbf = ctx.match(f["d1"], f["d2"], 0.75, mutual=False)
    np.testing.assert_array_equal(m, bf)
    assert n == nbf > 0.3 * min(n1, n2)
    # and the mutual result is reproducible call after call (atomics, but only min / count combinations)
    a = ctx.match_grid(window=(3, 3, 3, 3), nnr=0.75, mutual=True, **c)
    b = ctx.match_grid(window=(3, 3, 3, 3), nnr=0.75, mutual=True, **c)
    np.testing.assert_array_equal(a[0], b[0])
    assert a[1] == b[1] > 0


def test_fuzz_small_arbitrary_grids(ctx, oracle):
    """400 arbitrary small problems (hand-made CSR grids with repeated and out-of-range items, centres far outside the
    grid, one or two centres per row, tie-only descriptors, nnr on both sides of 1) against the sequential oracle."""
    pats = np.array([[0] * 32, [0xFF] + [0] * 31, [0x0F] * 32], np.uint8)
    hits = 0
    for seed in range(400):
        r = _rng(10_000 + seed)
        cols, rows, nc = int(r.integers(1, 5)), int(r.integers(1, 5)), int(r.integers(1, 3))
        n1, n2 = int(r.integers(0, 70)), int(r.integers(0, 40))
        d1, d2 = pats[r.integers(0, 3, n1)].reshape(-1, 32), pats[r.integers(0, 3, n2)].reshape(-1, 32)
        if seed % 3 == 0 and n1 and n2:                              # also real distances, not only ties
            d1, d2 = d1 ^ np.packbits(r.random((n1, 256)) < 0.1, axis=1), d2 ^ np.packbits(r.random((n2, 256)) < 0.1, axis=1)
        cen = np.stack([r.integers(-3, cols + 3, (n1, nc)), r.integers(-3, rows + 3, (n1, nc))], 2).astype(np.int32)
        lens = r.integers(0, 6, cols * rows)
        cs = np.zeros(cols * rows + 1, np.int32)
        np.cumsum(lens, out=cs[1:])
        items = r.integers(-1, n2 + 2, int(cs[-1])).astype(np.int32)
        kw = dict(centres=cen, d1=d1, cell_start=cs, cell_items=items, cols=cols, rows=rows, d2=d2,
                  window=tuple(int(v) for v in r.integers(0, 4, 4)), nnr=float(r.choice([0.5, 0.75, 1.0, 1.5])),
                  mutual=bool(seed & 1))
        if nc == 2 and seed % 4 == 0 and n1 and n2:
            with np.errstate(all="ignore"):
                kw.update(dir1=G.directions(cen.reshape(-1, 4).astype(np.float64)), dir2=G.directions(r.normal(size=(n2, 4))), sim_th=0.6)
        got, ref = ctx.match_grid(**kw), oracle.match_grid(**kw)
        np.testing.assert_array_equal(got[0], ref[0], err_msg=f"seed {seed}")
        assert got[1] == ref[1], seed
        hits += int((ref[0] >= 0).sum())
    assert hits > 500


def test_two_launch_path_does_not_depend_on_the_candidate_order(ctx, oracle):
    """One problem alone on the chip (>= 512 rows, LDS-resident, mutual) gets its distances from k_grid_candidates on many
    workgroups: the candidate list's order is whatever the scheduling made it.  Every repetition must return the first
    one's table and count -- which is the oracle's (tie stress: equal distances everywhere)."""
    c = point_case(11, 1500, 1400, G.GRID_COLS, G.GRID_ROWS, ties=True)
    ref = oracle.match_grid(window=(3, 3, 3, 3), nnr=0.8, mutual=True, **c)
    for _ in range(25):
        m, n = ctx.match_grid(window=(3, 3, 3, 3), nnr=0.8, mutual=True, **c)
        np.testing.assert_array_equal(m, ref[0])
        assert n == ref[1]


def _regrid(c, edit):
    """The case's grid with its cell lists edited: edit(lists) gets one Python list per cell."""
    cs, items = np.asarray(c["cell_start"]), np.asarray(c["cell_items"])
    lists = [list(items[cs[k]:cs[k + 1]]) for k in range(len(cs) - 1)]
    edit(lists)
    out = dict(c)
    out["cell_start"] = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int32)
    out["cell_items"] = np.array([i for x in lists for i in x] + [0], np.int32)[:max(1, sum(len(x) for x in lists))]
    return out


def test_lone_problem_records_path_every_branch(ctx, oracle):
    """One problem alone on the chip (>= 128 rows, mutual, row and column numbers of 23 bits together): k_grid_records lists each
    (cell, column) run's records in the run's own 8 words, k_match_grid folds them.  The branches: every column with ONE run
    (points: the list is folded as it stands); an item in two cells or twice in one cell (two runs: the tables are wiped and
    the list is bucketed by column); item numbers outside [0, n2) and cells that hold only those; runs of more than 8 records
    (whole-grid windows over 1024+ rows: the words beyond the eighth are listed from the end of the store); a row count that
    does not fill the last wave's quarter; windows that reach nothing."""
    r = _rng(4242)
    c = point_case(31, 1500, 1400, G.GRID_COLS, G.GRID_ROWS)
    w3 = (3, 3, 3, 3)
    _same(ctx, oracle, c, w3, 0.75, True)                                      # one run per column
    ncell = G.GRID_COLS * G.GRID_ROWS

    def dup_across(lists):                                                     # 200 items copied into a neighbouring cell
        for k in r.choice(ncell - 1, 200, replace=False):
            if lists[k]:
                lists[k + 1].append(lists[k][0])
    _same(ctx, oracle, _regrid(c, dup_across), w3, 0.75, True)

    def dup_within(lists):                                                     # ... and twice in their own
        for k in r.choice(ncell, 200, replace=False):
            if lists[k]:
                lists[k].append(lists[k][-1])
    _same(ctx, oracle, _regrid(c, dup_within), w3, 0.9, True)

    def strays(lists):                                                         # numbers no desc2 row answers to
        for k in r.choice(ncell, 300, replace=False):
            lists[k].insert(0, int(r.choice([-1, -7, 1400, 1401, 2 ** 31 - 1])))
    _same(ctx, oracle, _regrid(c, strays), w3, 0.75, True)

    # long runs: every row reaches every cell
    for n1, n2, cols, rows in ((1024, 700, 7, 5), (1531, 300, 16, 12), (2050, 64, 2, 3)):
        for mk in (point_case, line_case):
            cc = mk(900 + n1, n1, n2, cols, rows, ties=(n1 % 2 == 0))
            _same(ctx, oracle, cc, (cols, cols, rows, rows), 0.75, True)
            _same(ctx, oracle, cc, (0, 0, 0, 0), 0.9, True)                  # one cell each
    # windows that reach nothing: every centre far outside the grid
    far = dict(c)
    far["centres"] = np.asarray(c["centres"]) + 10 ** 6
    ref = _same(ctx, oracle, far, w3, 0.75, True)
    assert ref[1] == 0
