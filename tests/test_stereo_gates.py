"""The stereo L<->R gates of StVO::StereoFrame (stvo-pl stereoFrame.cpp matchStereoPoints / matchStereoLines, [RECALL];
SURVEY 8 a4; thresholds = the reference's config keys config/config/config_kitti.yaml:25-36): oracle semantics on CPU,
HIP kernels K15/K16 against the oracle on the GPU (tables and disparities bit-exact)."""
import numpy as np
import pytest


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


def stereo_points(seed, n_l=1500, n_r=1500, width=752, height=480):
    """Left key points, right ones displaced by a disparity along x with sub-pixel noise in y; a match table that is
    mostly right, sometimes wrong, sometimes empty."""
    r = _rng(seed)
    kp_l = np.stack([r.uniform(0, width, n_l), r.uniform(0, height, n_l)], 1).astype(np.float32)
    src = r.permutation(max(n_l, n_r))[:n_r] % max(n_l, 1)
    disp = r.uniform(-3, 60, n_r)
    kp_r = (kp_l[src] - np.stack([disp, r.normal(0, 0.8, n_r)], 1)).astype(np.float32)
    m12 = np.full(n_l, -1, np.int32)
    m12[src] = np.arange(n_r)
    wrong = r.random(n_l) < 0.1
    m12[wrong] = r.integers(0, max(n_r, 1), int(wrong.sum()))
    m12[r.random(n_l) < 0.2] = -1
    return m12, kp_l, kp_r


def stereo_lines(seed, n_l=200, n_r=200, width=752, height=480):
    r = _rng(seed)
    a = np.stack([r.uniform(0, width, n_l), r.uniform(0, height, n_l)], 1)
    ang = r.uniform(0, np.pi, n_l)
    ln = r.uniform(5, 150, n_l)
    ln[r.random(n_l) < 0.05] = 0.0                                     # zero-length segments
    ang[r.random(n_l) < 0.1] = 0.0                                     # horizontal segments (dy = 0: division by zero)
    seg_l = np.concatenate([a, a + np.stack([np.cos(ang), np.sin(ang)], 1) * ln[:, None]], 1)
    src = r.permutation(max(n_l, n_r))[:n_r] % max(n_l, 1)
    d0, d1 = r.uniform(-2, 50, n_r), r.uniform(0.5, 1.5, n_r)
    seg_r = seg_l[src].copy()
    seg_r[:, 0] -= d0
    seg_r[:, 2] -= d0 * d1
    seg_r += r.normal(0, 0.7, seg_r.shape)
    cut = r.random(n_r) < 0.3                                          # right segment only partly overlapping in y
    seg_r[cut, 2:] = seg_r[cut, :2] + (seg_r[cut, 2:] - seg_r[cut, :2]) * r.uniform(0.1, 0.9, (int(cut.sum()), 1))
    m12 = np.full(n_l, -1, np.int32)
    m12[src] = np.arange(n_r)
    wrong = r.random(n_l) < 0.1
    m12[wrong] = r.integers(0, max(n_r, 1), int(wrong.sum()))
    m12[r.random(n_l) < 0.15] = -1
    return m12, seg_l.astype(np.float32), seg_r.astype(np.float32)


def test_point_gate_semantics(oracle):
    kp_l = np.array([[100.0, 50.0], [100.0, 50.0], [100.0, 50.0], [100.0, 50.0], [10.0, 10.0]], np.float32)
    kp_r = np.array([[90.0, 50.5], [90.0, 51.5], [99.5, 50.0], [99.0, 50.0]], np.float32)
    m12 = np.array([0, 1, 2, 3, -1], np.int32)
    out, disp, n = oracle.stereo_point_gate(m12, kp_l, kp_r, 1.0, 1.0)
    # row 1: |dy| = 1.5 > 1; row 2: disparity 0.5 < 1; row 3: disparity exactly 1 (>=) passes
    assert out.tolist() == [0, -1, -1, 3, -1] and n == 2
    assert disp.tolist() == [10.0, 0.0, 0.0, 1.0, 0.0]
    # max_dist_epip = 0 (config_kitti.yaml:25) keeps only rows with identical y
    assert oracle.stereo_point_gate(m12, kp_l, kp_r, 0.0, 1.0)[0].tolist() == [-1, -1, -1, 3, -1]
    # the arithmetic is float: 0.1f + 0.2f style -- a y difference that is <= 1 only in float
    a = np.array([[5.0, 16777216.0]], np.float32)
    b = np.array([[1.0, 16777215.0]], np.float32)
    assert oracle.stereo_point_gate([0], a, b, 1.0, 1.0)[2] == 1


def test_line_gate_semantics(oracle):
    # a vertical pair shifted by 10 px: overlap 1, disparities (10, 10)
    seg_l = np.array([[100, 10, 100, 60]], np.float32)
    seg_r = np.array([[90, 10, 90, 60]], np.float32)
    out, d, n = oracle.stereo_line_gate([0], seg_l, seg_r, 1.0, 0.1, 0.75, 0.7)
    assert out.tolist() == [0] and n == 1 and d.tolist() == [[10.0, 10.0]]
    # the right segment covers half of the left one's rows: overlap 0.5 < 0.75 -> dropped
    assert oracle.stereo_line_gate([0], seg_l, np.array([[90, 10, 90, 35]], np.float32), 1.0, 0.1, 0.75, 0.7)[2] == 0
    assert oracle.stereo_line_gate([0], seg_l, np.array([[90, 10, 90, 35]], np.float32), 1.0, 0.1, 0.4, 0.7)[2] == 1
    # disparities 10 and 4: ratio 0.4 < 0.7 -> both -1 -> dropped
    assert oracle.stereo_line_gate([0], seg_l, np.array([[90, 10, 96, 60]], np.float32), 1.0, 0.1, 0.75, 0.7)[2] == 0
    # horizontal left segment
    assert oracle.stereo_line_gate([0], np.array([[100, 10, 160, 10.05]], np.float32), seg_r, 1.0, 0.1, 0.75, 0.7)[2] == 0
    # lineSegmentOverlapStereo itself: disjoint, contained, partial
    L = oracle.lib().plo_line_segment_overlap_stereo
    assert L(0, 10, 20, 30, 0.1) == 0.0 and L(0, 10, -5, 15, 0.1) == 10 / 15 and L(0, 10, 5, 20, 0.1) == 1.0
    assert L(3.0, 3.05, 0, 100, 0.1) == 1.0                            # horizontal: the test is skipped


@pytest.mark.gpu
@pytest.mark.parametrize("n_l,n_r", [(1500, 1500), (4000, 3500), (37, 64), (1, 1)])
def test_gpu_point_gate_bit_exact(ctx, oracle, n_l, n_r):
    m12, kp_l, kp_r = stereo_points(n_l + n_r, n_l, n_r)
    for th, md in ((1.0, 1.0), (0.0, 1.0), (2.5, 0.0), (0.5, -5.0)):
        got = ctx.stereo_point_gate(m12, kp_l, kp_r, th, md)
        ref = oracle.stereo_point_gate(m12, kp_l, kp_r, th, md)
        np.testing.assert_array_equal(got[0], ref[0])
        np.testing.assert_array_equal(got[1], ref[1])
        assert got[2] == ref[2] == int((ref[0] >= 0).sum())
    assert ref[2] > 0 or n_l < 10


@pytest.mark.gpu
@pytest.mark.parametrize("n_l,n_r", [(200, 200), (600, 640), (19, 7), (1, 1)])
def test_gpu_line_gate_bit_exact(ctx, oracle, n_l, n_r):
    m12, seg_l, seg_r = stereo_lines(n_l * 3 + n_r, n_l, n_r)
    tot = 0
    with np.errstate(all="ignore"):
        for md, hz, ov, ratio in ((1.0, 0.1, 0.75, 0.7), (1.0, 1.0, 0.75, 0.7), (0.0, 0.1, 0.2, 0.3), (-2.0, 0.0, 0.0, 0.0)):
            got = ctx.stereo_line_gate(m12, seg_l, seg_r, md, hz, ov, ratio)
            ref = oracle.stereo_line_gate(m12, seg_l, seg_r, md, hz, ov, ratio)
            np.testing.assert_array_equal(got[0], ref[0])
            np.testing.assert_array_equal(got[1], ref[1])
            assert got[2] == ref[2] == int((ref[0] >= 0).sum())
            tot += ref[2]
    assert tot > 0 or n_l < 10


@pytest.mark.gpu
def test_gpu_stereo_association_end_to_end(ctx, oracle):
    """Descriptors -> StVO::matchGrid with the stereo window {matching_s_ws, 0, 0, 0} -> gate: the frame's stereo
    points, all on the device path, against the oracle chain."""
    from plslam_amd import grid as G
    from plslam_amd import synth
    r = _rng(77)
    n = 1200
    _, kp_l, _ = stereo_points(5, n, n)
    disp = r.uniform(2, 40, n)
    kp_r = (kp_l - np.stack([disp, r.normal(0, 0.3, n)], 1)).astype(np.float32)
    d_l = synth.random_desc(r, n)
    d_r = d_l ^ np.packbits(r.random((n, 256)) < 0.05, axis=1)
    perm = r.permutation(n)
    kp_r, d_r = kp_r[perm], np.ascontiguousarray(d_r[perm])
    sc = np.array([G.GRID_COLS / 752.0, G.GRID_ROWS / 480.0])
    cs, items = G.fill_points(G.to_cells(kp_r.astype(np.float64) * sc))
    c = dict(centres=G.to_cells(kp_l.astype(np.float64) * sc), d1=d_l, cell_start=cs, cell_items=items, cols=G.GRID_COLS,
             rows=G.GRID_ROWS, d2=d_r, window=(10, 0, 0, 0), nnr=0.75, mutual=True)
    m_gpu, m_ref = ctx.match_grid(**c), oracle.match_grid(**c)
    np.testing.assert_array_equal(m_gpu[0], m_ref[0])
    got = ctx.stereo_point_gate(m_gpu[0], kp_l, kp_r, 1.0, 1.0)
    ref = oracle.stereo_point_gate(m_ref[0], kp_l, kp_r, 1.0, 1.0)
    np.testing.assert_array_equal(got[0], ref[0])
    np.testing.assert_array_equal(got[1], ref[1])
    truth = np.argsort(perm)                                           # left i1 <-> right truth[i1]
    kept = got[0] >= 0
    assert kept.sum() > 0.6 * n and (got[0][kept] == truth[kept]).mean() > 0.99
