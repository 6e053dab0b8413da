"""K17: the pose-only Gauss-Newton system of the loop-closure relative pose (MapHandler::computeRelativePoseGN /
computeRelativePoseRobustGN iteration body, src/mapHandler.cpp:3324-3424, :3588-3689).  CPU: the oracle's rows against
finite differences of the residual (the reference's Jacobian is that of a perturbation applied in the camera frame,
P_ -> expmap(delta) * T_inc * P; its update T_inc <- T_inc * inverse(expmap(x)) (:3428) agrees with it at T_inc = I,
where the iteration starts); GPU: H, g, e, N against the oracle within 1e-9 relative (contract 1e-6)."""
import numpy as np
import pytest

from plslam_amd import synth


def scene(n_pt=300, n_ls=80, seed=2):
    r = np.random.Generator(np.random.PCG64(seed))
    K = synth.EUROC
    T = synth.se3_exp([0.05, -0.03, 0.08, 0.01, 0.02, -0.015])
    z = r.uniform(2, 25, n_pt + 2 * n_ls)
    u, v = r.uniform(0, K["width"], z.size), r.uniform(0, K["height"], z.size)
    X = np.stack([(u - K["cx"]) / K["fx"] * z, (v - K["cy"]) / K["fy"] * z, z], 1)
    Xc = X @ T[:3, :3].T + T[:3, 3]
    proj = lambda Q: np.stack([K["cx"] + K["fx"] * Q[:, 0] / Q[:, 2], K["cy"] + K["fy"] * Q[:, 1] / Q[:, 2]], 1)
    P, S, E = X[:n_pt], X[n_pt:n_pt + n_ls], X[n_pt + n_ls:]
    pl_obs = proj(Xc[:n_pt]) + r.normal(0, 1.0, (n_pt, 2))
    ps = np.concatenate([proj(Xc[n_pt:n_pt + n_ls]), np.ones((n_ls, 1))], 1)
    pe = np.concatenate([proj(Xc[n_pt + n_ls:]), np.ones((n_ls, 1))], 1)
    le = np.cross(ps, pe)
    le /= np.sqrt(le[:, 0] ** 2 + le[:, 1] ** 2)[:, None]
    le[:, 2] += r.normal(0, 1.0, n_ls)
    return dict(T=T @ synth.se3_exp(r.normal(0, 0.01, 6)), P=P, pl_obs=pl_obs, pt_in=(r.random(n_pt) < 0.85).astype(np.uint8),
                sPeP=np.concatenate([S, E], 1), le_obs=le, ls_in=(r.random(n_ls) < 0.8).astype(np.uint8))


def _args(s):
    return s["T"], s["P"], s["pl_obs"], s["pt_in"], s["sPeP"], s["le_obs"], s["ls_in"]


def test_oracle_rows_are_the_derivative_of_the_residual(oracle):
    """One observation at a time: g = J r w => J; compare with central differences of r under expmap(d) * T_inc."""
    # fx == fy: the reference scales BOTH image coordinates with fx (fgz2 = fx / z^2), which is exact only then
    K = dict(synth.EUROC, fy=synth.EUROC["fx"])
    cam = oracle.make_cam(**K)
    s = scene(6, 4, seed=5)
    none_p, none_l = np.zeros(6, np.uint8), np.zeros(4, np.uint8)

    def r_point(T, i):
        G = T[:3, :3] @ s["P"][i] + T[:3, 3]
        p = np.array([K["cx"] + K["fx"] * G[0] / G[2], K["cy"] + K["fy"] * G[1] / G[2]])
        return np.linalg.norm(p - s["pl_obs"][i])

    def r_line(T, i):
        out = []
        for X in (s["sPeP"][i, :3], s["sPeP"][i, 3:]):
            G = T[:3, :3] @ X + T[:3, 3]
            out.append(s["le_obs"][i] @ np.array([K["cx"] + K["fx"] * G[0] / G[2], K["cy"] + K["fy"] * G[1] / G[2], 1.0]))
        return np.linalg.norm(out)

    for kind, n, rf in (("p", 6, r_point), ("l", 4, r_line)):
        for i in range(n):
            mp, ml = none_p.copy(), none_l.copy()
            (mp if kind == "p" else ml)[i] = 1
            H, g, e, cnt = oracle.pose_gn_accumulate(cam, 1e-7, s["T"], s["P"], s["pl_obs"], mp, s["sPeP"], s["le_obs"], ml)
            r = rf(s["T"], i)
            w = 1.0 / (1.0 + r * r)
            assert cnt == ((1, 0) if kind == "p" else (0, 1)) and np.isclose(e, r * r * w, rtol=1e-12)
            J = g / (r * w)
            np.testing.assert_allclose(H, np.outer(J, J) * w, rtol=1e-9, atol=1e-12)
            fd = np.zeros(6)
            for k in range(6):
                d = np.zeros(6)
                d[k] = 1e-6
                Tp = synth.se3_exp(d) @ s["T"]
                Tm = synth.se3_exp(-d) @ s["T"]
                fd[k] = (rf(Tp, i) - rf(Tm, i)) / 2e-6
            # J = +dr/d(delta), tangent order [t, w]: sign and ordering of the six components pinned
            np.testing.assert_allclose(J, fd, rtol=1e-5, atol=1e-6 * np.abs(fd).max())


@pytest.mark.gpu
@pytest.mark.parametrize("n_pt,n_ls", [(300, 80), (1500, 200), (7, 0), (0, 5), (1, 1)])
def test_gpu_system_matches_oracle(ctx, oracle, n_pt, n_ls):
    import plslam_amd
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), oracle.make_cam(**synth.EUROC)
    s = scene(n_pt, n_ls, seed=n_pt + n_ls)
    for th in (1e-7, 1e-3):
        H, g, e, n = ctx.pose_gn_accumulate(cam, th, *_args(s))
        rH, rg, re, rn = oracle.pose_gn_accumulate(ocam, th, *_args(s))
        assert n == rn == (int(s["pt_in"].sum()), int(s["ls_in"].sum()))
        scale = max(np.abs(rH).max(), 1e-300)
        np.testing.assert_allclose(H, rH, rtol=0, atol=1e-9 * scale)
        np.testing.assert_allclose(g, rg, rtol=0, atol=1e-9 * max(np.abs(rg).max(), 1e-300))
        assert abs(e - re) <= 1e-9 * max(abs(re), 1e-300)
        np.testing.assert_array_equal(H, H.T)


@pytest.mark.gpu
def test_gpu_gn_iterations_converge_like_the_oracle(ctx, oracle):
    """The loop of :3322-3435 with the system from the device vs from the oracle: same pose after 5 iterations."""
    import plslam_amd
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), oracle.make_cam(**synth.EUROC)
    s = scene(400, 100, seed=9)
    poses = []
    for f, c in ((ctx.pose_gn_accumulate, cam), (oracle.pose_gn_accumulate, ocam)):
        T = np.eye(4)
        for _ in range(5):
            H, g, e, n = f(c, 1e-7, T, *_args(s)[1:])
            x = np.linalg.solve(H, g)
            T = T @ np.linalg.inv(synth.se3_exp(x))
        poses.append(T)
    np.testing.assert_allclose(poses[0], poses[1], rtol=0, atol=1e-9)
    assert np.abs(poses[0] - np.eye(4)).max() > 1e-3               # it did move


@pytest.mark.gpu
def test_gpu_committed_golden(ctx):
    """No oracle at run time: tests/golden/pose_gn_golden.npz (tests/golden/make_k17_k18_golden.py)."""
    import os
    import plslam_amd
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pose_gn_golden.npz"))
    cam = plslam_amd.make_cam(**synth.EUROC)
    H, gg, e, n = ctx.pose_gn_accumulate(cam, 1e-7, g["T"], g["P"], g["pl_obs"], g["pt_in"], g["sPeP"], g["le_obs"], g["ls_in"])
    assert n == tuple(g["n"])
    np.testing.assert_allclose(H, g["H"], rtol=0, atol=1e-9 * np.abs(g["H"]).max())
    np.testing.assert_allclose(gg, g["g"], rtol=0, atol=1e-9 * np.abs(g["g"]).max())
    assert abs(e - float(g["e"])) <= 1e-9 * abs(float(g["e"]))
