"""GPU parity tests for K12/K13, the batched representative descriptor of every landmark
(MapPoint/MapLine::updateAverageDescDir, src/mapFeatures.cpp:51-84, :121-157): winner indices and
rows bit-exact against the oracle, including ties (first row wins), single and empty lists."""
import numpy as np
import pytest

import plslam_amd
from plslam_amd import synth

pytestmark = pytest.mark.gpu


def _rng(seed):
    return np.random.Generator(np.random.PCG64(seed))


@pytest.mark.parametrize("n_lm,max_obs,ties,empty", [(1, 1, False, 0.0), (1, 2, False, 0.0), (7, 3, True, 0.0),
                                                      (300, 8, False, 0.05), (300, 8, True, 0.05),
                                                      (64, 70, False, 0.0), (1000, 12, True, 0.2)])
def test_batched_vs_oracle(ctx, oracle, n_lm, max_obs, ties, empty):
    d, off = synth.landmark_desc_lists(_rng(n_lm + max_obs), n_lm, max_obs=max_obs, empty_frac=empty, ties=ties)
    idx, md = ctx.median_desc_batched(d, off)
    ridx, rmd = oracle.median_desc_batched(d, off)
    np.testing.assert_array_equal(idx, ridx)
    np.testing.assert_array_equal(md, rmd)
    idx2, none = ctx.median_desc_batched(d, off, want_desc=False)
    assert none is None
    np.testing.assert_array_equal(idx2, ridx)


def test_all_lists_empty_and_no_landmarks(ctx):
    idx, md = ctx.median_desc_batched(np.zeros((0, 32), np.uint8), np.zeros(6, np.int32))
    assert (idx == -1).all() and not md.any()
    idx, md = ctx.median_desc_batched(np.zeros((0, 32), np.uint8), np.zeros(1, np.int32))
    assert idx.shape == (0,) and md.shape == (0, 32)


def test_identical_observations_first_row_wins(ctx, oracle):
    """All rows equal => every row's median is 0 => index 0 (strict '<', :78)."""
    row = synth.random_desc(_rng(2), 1)
    d = np.repeat(row, 9, axis=0)
    off = np.array([0, 4, 9], np.int32)
    idx, md = ctx.median_desc_batched(d, off)
    assert idx.tolist() == [0, 0]
    np.testing.assert_array_equal(md, np.repeat(row, 2, axis=0))
    # two observations: both rows have median d(0,1) -> first wins, whatever the distance
    d2 = synth.random_desc(_rng(3), 2)
    assert ctx.median_desc_batched(d2, np.array([0, 2], np.int32))[0].tolist() == [0] == [oracle.median_desc(d2)]


def test_one_long_list(ctx, oracle):
    """A landmark observed 700 times (far beyond any real map): the bisection has no size limit."""
    d, off = synth.landmark_desc_lists(_rng(9), 1, max_obs=1)
    r = _rng(10)
    d = np.repeat(d, 700, axis=0) ^ np.packbits(r.random((700, 256)) < 0.1, axis=1)
    off = np.array([0, 700], np.int32)
    idx, md = ctx.median_desc_batched(d, off)
    assert idx[0] == oracle.median_desc(d) == oracle.np_median_desc(d)
    np.testing.assert_array_equal(md[0], d[idx[0]])


def test_bad_offsets_are_refused(ctx):
    d = synth.random_desc(_rng(1), 4)
    with pytest.raises(plslam_amd.PlslamError):
        ctx.median_desc_batched(d, np.array([1, 4], np.int32))          # offsets[0] != 0
    with pytest.raises(plslam_amd.PlslamError):
        ctx.median_desc_batched(d, np.array([0, 3, 2, 4], np.int32))    # decreasing


def test_c3_map_device_resident_into_map2kf(ctx, oracle):
    """BASELINE config 3 scale: 10 000 landmarks x 5 observations on the device -> representative rows
    -> (as `med_desc`) the map<->keyframe driver; both stages equal the oracle's."""
    import torch
    from test_map2kf import scene
    s = scene(10000, 1500, seed=4)
    r = _rng(12)
    n_lm, n_obs = 10000, 5
    # observation lists whose representatives are the scene's map descriptors' neighbourhood
    lists = np.repeat(s["med"], n_obs, axis=0) ^ np.packbits(r.random((n_lm * n_obs, 256)) < 0.03, axis=1)
    off = (np.arange(n_lm + 1) * n_obs).astype(np.int32)
    d_desc, d_off = torch.from_numpy(lists).cuda(), torch.from_numpy(off).cuda()
    d_idx = torch.empty(n_lm, dtype=torch.int32, device="cuda")
    d_med = torch.empty((n_lm, 32), dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    ctx.median_desc_batched_dev(d_desc.data_ptr(), d_off.data_ptr(), n_lm, n_lm * n_obs, d_idx.data_ptr(),
                                d_med.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    ridx, rmd = oracle.median_desc_batched(lists, off)
    np.testing.assert_array_equal(d_idx.cpu().numpy(), ridx)
    med = d_med.cpu().numpy()
    np.testing.assert_array_equal(med, rmd)
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), oracle.make_cam(**synth.EUROC)
    args = (s["Twf"], s["LM"], med, s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"], 0.9, True, 1.0, 10)
    got, n = ctx.map2kf_match("points", cam, *args)
    exp, nr = oracle.map2kf_match("points", ocam, *args)
    assert n == nr and n > 50
    np.testing.assert_array_equal(got, exp)
