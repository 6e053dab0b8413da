"""GPU parity tests for the local-BA rows (K3/K4) and the map<->KF gates (K5/K6): tolerance 1e-6
relative as BASELINE.json's north_star states (observed: bit-exact, also asserted at 1e-12),
inlier masks bit-exact."""
import numpy as np
import pytest

import plslam_amd
from plslam_amd import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-6     # the contract (north_star: "residuals/Jacobians agree within 1e-6 relative")


def _cams():
    from oracle import oracle as O
    return plslam_amd.make_cam(**synth.EUROC), O.make_cam(**synth.EUROC)


def _close(a, b, rtol):
    scale = np.maximum(np.abs(b), 1e-300)
    return np.max(np.abs(a - b) / scale) <= rtol if a.size else True


def test_c3_point_and_line_rows(ctx, oracle):
    """BASELINE config 3: 10 KFs, 10 000 points x 5 obs (50 000 rows), 2 000 lines x 5 obs (10 000 rows)."""
    lm = synth.local_map()
    cam, ocam = _cams()
    got = ctx.lba_point_rows(cam, 1e-7, lm["T_kf_w"], lm["Xw"], lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
    exp = oracle.lba_point_rows(ocam, 1e-7, lm["T_kf_w"], lm["Xw"], lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
    for g, e in zip(got, exp):
        assert np.all(np.isfinite(g))
        assert _close(g, e, RTOL)
        assert _close(g, e, 1e-12)        # tighter than the contract: same op order, no FMA contraction
    got = ctx.lba_line_rows(cam, 1e-7, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"])
    exp = oracle.lba_line_rows(ocam, 1e-7, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"])
    for g, e in zip(got, exp):
        assert _close(g, e, RTOL) and _close(g, e, 1e-12)


def test_line_rows_iteration_pass_compat(ctx, oracle):
    lm = synth.local_map(n_kf=6, n_pt=0, n_ls=500, obs_per_lm=4)
    cam, ocam = _cams()
    got = ctx.lba_line_rows(cam, 1e-3, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"],
                            compat_iter_pass=True)
    exp = oracle.lba_line_rows(ocam, 1e-3, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"],
                               compat_iter_pass=True)
    for g, e in zip(got, exp):
        assert _close(g, e, RTOL) and _close(g, e, 1e-12)


def test_rows_degenerate_and_empty(ctx, oracle):
    cam, ocam = _cams()
    T = np.eye(4).reshape(1, 16)
    X = np.array([[0.0, 0.0, 1e-5], [0.1, -0.2, 5.0], [1.0, 1.0, -3.0]])
    uv = np.array([[synth.EUROC["cx"], synth.EUROC["cy"]], [380.0, 230.0], [100.0, 50.0]])
    got = ctx.lba_point_rows(cam, 1e-7, T, X, uv, [0, 1, 2], [0, 0, 0])
    exp = oracle.lba_point_rows(ocam, 1e-7, T, X, uv, [0, 1, 2], [0, 0, 0])
    for g, e in zip(got, exp):
        assert np.array_equal(g, e)
    got = ctx.lba_point_rows(cam, 1e-7, T, X, np.zeros((0, 2)), [], [])
    assert all(g.shape[0] == 0 for g in got)
    with pytest.raises(plslam_amd.PlslamError):          # out-of-range indices are rejected, not read
        ctx.lba_point_rows(cam, 1e-7, T, X, uv, [0, 1, 3], [0, 0, 0])
    with pytest.raises(plslam_amd.PlslamError):
        ctx.lba_point_rows(cam, 1e-7, T, X, uv, [0, 1, 2], [0, 1, 0])


def test_rows_linearity_property_full_size(ctx):
    """Size-independent property at C3 size: rows depend on the pose only through T^-1 X -- moving
    world and poses by one rigid transform leaves r, w, J_pose unchanged and rotates J_lm."""
    lm = synth.local_map()
    cam, _ = _cams()
    G = synth.se3_exp([0.3, -0.2, 0.1, 0.02, -0.01, 0.03])
    T2 = np.einsum("ij,kjl->kil", G, lm["T_kf_w"].reshape(-1, 4, 4)).reshape(-1, 16)
    X2 = lm["Xw"] @ G[:3, :3].T + G[:3, 3]
    a = ctx.lba_point_rows(cam, 1e-7, lm["T_kf_w"], lm["Xw"], lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
    b = ctx.lba_point_rows(cam, 1e-7, T2, X2, lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
    assert np.allclose(a[2], b[2], rtol=1e-7, atol=1e-9) and np.allclose(a[3], b[3], rtol=1e-7)
    assert np.allclose(a[0], b[0], rtol=1e-5, atol=1e-7)
    assert np.allclose(a[1] @ G[:3, :3].T, b[1], rtol=1e-5, atol=1e-7)


def test_gates_and_visibility_bit_exact(ctx, oracle):
    r = np.random.Generator(np.random.PCG64(8))
    cam, ocam = _cams()
    K = synth.EUROC
    Twf = np.linalg.inv(synth.se3_exp([0.1, -0.05, 0.3, 0.01, 0.02, -0.01]))
    n = 10000
    X = np.stack([r.uniform(-6, 6, n), r.uniform(-4, 4, n), r.uniform(-1, 30, n)], 1)
    assert np.array_equal(ctx.map_point_visible(cam, Twf, X), oracle.map_point_visible(ocam, Twf, X))
    Lw = np.concatenate([X[:2000], X[:2000] + r.uniform(-1, 1, (2000, 3))], 1)
    assert np.array_equal(ctx.map_line_visible(cam, Twf, Lw), oracle.map_line_visible(ocam, Twf, Lw))
    Xc = X @ Twf[:3, :3].T + Twf[:3, 3]
    with np.errstate(all="ignore"):
        uv = np.stack([K["cx"] + K["fx"] * Xc[:, 0] / Xc[:, 2], K["cy"] + K["fy"] * Xc[:, 1] / Xc[:, 2]], 1)
    nt = 1500
    pick = r.integers(0, n, nt)
    pl = np.nan_to_num(uv[pick]) + r.normal(0, 0.7, (nt, 2))    # thresholds straddled: |err| ~ 1 px
    m12 = np.full(n, -1, np.int32)
    m12[pick] = np.arange(nt)
    mask, cnt = ctx.map2kf_point_gate(cam, Twf, X, m12, pl, 1.0)
    emask, ecnt = oracle.map2kf_point_gate(ocam, Twf, X, m12, pl, 1.0)
    assert np.array_equal(mask, emask) and cnt == ecnt and 0 < cnt < nt
    le = r.normal(0, 1, (200, 3))
    le /= np.linalg.norm(le[:, :2], axis=1, keepdims=True)
    m12l = r.integers(-1, 200, 2000).astype(np.int32)
    mask, cnt = ctx.map2kf_line_gate(cam, Twf, Lw, m12l, le, 1.0)
    emask, ecnt = oracle.map2kf_line_gate(ocam, Twf, Lw, m12l, le, 1.0)
    assert np.array_equal(mask, emask) and cnt == ecnt
