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


def test_rows_with_more_pose_slots_than_the_workgroups_cache_has_lines(ctx, oracle):
    """The row kernels keep their workgroup's pose matrices in a direct-mapped LDS cache of 32 lines (round 6): 70 key frames whose
    observations are shuffled put several slots on one line in every workgroup -- the losers read global memory.  Same rows as
    the oracle's, for the row kernels and for the plan's fused iteration (err and gradient against the dense accumulation)."""
    lm = synth.local_map(n_kf=70, n_pt=3000, n_ls=600, obs_per_lm=6, seed=5)
    cam, ocam = _cams()
    rng = np.random.Generator(np.random.PCG64(9))
    pp, pl = rng.permutation(lm["pt_lm"].shape[0]), rng.permutation(lm["ls_lm"].shape[0])
    pt = {k: lm[k][pp] for k in ("obs_uv", "pt_lm", "pt_kf")}
    ls = {k: lm[k][pl] for k in ("l_obs", "ls_lm", "ls_kf")}
    got = ctx.lba_point_rows(cam, 1e-7, lm["T_kf_w"], lm["Xw"], pt["obs_uv"], pt["pt_lm"], pt["pt_kf"])
    exp = oracle.lba_point_rows(ocam, 1e-7, lm["T_kf_w"], lm["Xw"], pt["obs_uv"], pt["pt_lm"], pt["pt_kf"])
    for g, e in zip(got, exp):
        assert _close(g, e, 1e-12)
    for compat in (False, True):
        got = ctx.lba_line_rows(cam, 1e-7, lm["T_kf_w"], lm["Lw"], ls["l_obs"], ls["ls_lm"], ls["ls_kf"], compat_iter_pass=compat)
        exp = oracle.lba_line_rows(ocam, 1e-7, lm["T_kf_w"], lm["Lw"], ls["l_obs"], ls["ls_lm"], ls["ls_kf"], compat_iter_pass=compat)
        for g, e in zip(got, exp):
            assert _close(g, e, 1e-12)
    nkf, npt, nls = 69, 3000, 600
    plan = plslam_amd.LbaPlan(ctx, cam, 1e-7, 70, nkf, npt, nls, pt["pt_lm"], pt["pt_kf"], pt["pt_kf"] - 1, pt["obs_uv"],
                              ls["ls_lm"], ls["ls_kf"], ls["ls_kf"] - 1, ls["l_obs"])
    B = plan.iterate(lm["T_kf_w"], lm["Xw"], lm["Lw"])
    rp = oracle.lba_point_rows(ocam, 1e-7, lm["T_kf_w"], lm["Xw"], pt["obs_uv"], pt["pt_lm"], pt["pt_kf"])
    rl = oracle.lba_line_rows(ocam, 1e-7, lm["T_kf_w"], lm["Lw"], ls["l_obs"], ls["ls_lm"], ls["ls_kf"])
    H, g, e1 = oracle.lba_accumulate("points", nkf, npt, nls, pt["pt_lm"], pt["pt_kf"] - 1, *rp)
    H, g, e2 = oracle.lba_accumulate("lines", nkf, npt, nls, ls["ls_lm"], ls["ls_kf"] - 1, *rl, H=H, g=g)
    assert np.allclose(B["g"], g, rtol=0, atol=1e-9 * np.abs(g).max())
    assert abs(B["err"] - (e1 + e2)) <= 1e-9 * (e1 + e2)
    plan.close()


def test_device_pointer_rows_with_and_without_the_pose_count(ctx):
    """plslam_lba_point_rows_dev_n / _line_rows_dev_n (round 6): stating how many pose matrices T holds lets the kernels keep the
    first 32 in LDS; the rows are the same words as with the count left out (0), for 70 slots (38 of them beyond the cache) and 10."""
    import torch
    dev = torch.device("cuda", 0)
    cam, _ = _cams()
    for n_kf in (70, 10):
        lm = synth.local_map(n_kf=n_kf, n_pt=2000, n_ls=500, obs_per_lm=5, seed=n_kf)
        g = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in lm.items()}
        for kind in ("pt", "ls"):
            n = lm["pt_lm" if kind == "pt" else "ls_lm"].shape[0]
            outs = []
            for slots in (0, n_kf):
                Jp = torch.zeros((n, 6), dtype=torch.float64, device=dev)
                Jl = torch.zeros((n, 3 if kind == "pt" else 6), dtype=torch.float64, device=dev)
                rr, ww = torch.zeros(n, dtype=torch.float64, device=dev), torch.zeros(n, dtype=torch.float64, device=dev)
                if kind == "pt":
                    ctx.lba_point_rows_dev(cam, 1e-7, g["T_kf_w"].data_ptr(), g["Xw"].data_ptr(), g["obs_uv"].data_ptr(), g["pt_lm"].data_ptr(),
                                           g["pt_kf"].data_ptr(), n, Jp.data_ptr(), Jl.data_ptr(), rr.data_ptr(), ww.data_ptr(), 0, n_pose_slots=slots)
                else:
                    ctx.lba_line_rows_dev(cam, 1e-7, False, g["T_kf_w"].data_ptr(), g["Lw"].data_ptr(), g["l_obs"].data_ptr(), g["ls_lm"].data_ptr(),
                                          g["ls_kf"].data_ptr(), n, Jp.data_ptr(), Jl.data_ptr(), rr.data_ptr(), ww.data_ptr(), 0, n_pose_slots=slots)
                torch.cuda.synchronize()
                outs.append([t.cpu().numpy() for t in (Jp, Jl, rr, ww)])
            for a, b in zip(*outs):
                assert np.array_equal(a, b), (n_kf, kind)
            assert np.abs(outs[0][0]).max() > 0


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


def _assert_blocks_match_dense(B, H, g, nkf, npt, nls, pt_lm, pt_kf, ls_lm, ls_kf):
    """Landmark and cross blocks (and the landmark part of g) are bit-identical to the dense
    accumulation; the keyframe blocks / pose part of g are a fixed-shape two-level sum: equal to
    1e-13 of the block's scale."""
    Hd = _expand_blocks(B, nkf, npt, nls, pt_lm, pt_kf, ls_lm, ls_kf)
    p6 = 6 * nkf
    assert np.array_equal(Hd[p6:, :], H[p6:, :]) and np.array_equal(Hd[:, p6:], H[:, p6:])
    assert np.array_equal(B["g"][p6:], g[p6:])
    if nkf:
        scale = max(np.abs(H[:p6, :p6]).max(), 1e-300)
        assert np.abs(Hd[:p6, :p6] - H[:p6, :p6]).max() <= 1e-13 * scale
        assert np.abs(B["g"][:p6] - g[:p6]).max() <= 1e-13 * max(np.abs(g[:p6]).max(), 1e-300)


def _expand_blocks(B, nkf, npt, nls, pt_lm, pt_kf, ls_lm, ls_kf):
    """Scatter the block form back into the reference's dense H (src/mapHandler.cpp:1410-1429, :1519-1538)."""
    N = 6 * nkf + 3 * npt + 6 * nls
    H = np.zeros((N, N))
    for k in range(nkf):
        H[6 * k:6 * k + 6, 6 * k:6 * k + 6] = B["H_pose"][k]
    for l in range(npt):
        j = 6 * nkf + 3 * l
        H[j:j + 3, j:j + 3] = B["H_pt"][l]
    for l in range(nls):
        j = 6 * nkf + 3 * npt + 6 * l
        H[j:j + 6, j:j + 6] = B["H_ls"][l]
    for o in range(len(pt_lm)):
        if pt_kf[o] >= 0:
            j, i = 6 * nkf + 3 * pt_lm[o], 6 * pt_kf[o]
            H[j:j + 3, i:i + 6] += B["W_pt"][o]
            H[i:i + 6, j:j + 3] += B["W_pt"][o].T
    for o in range(len(ls_lm)):
        if ls_kf[o] >= 0:
            j, i = 6 * nkf + 3 * npt + 6 * ls_lm[o], 6 * ls_kf[o]
            H[j:j + 6, i:i + 6] += B["W_ls"][o]
            H[i:i + 6, j:j + 6] += B["W_ls"][o].T
    return H


@pytest.mark.parametrize("n_kf,n_pt,n_ls,obs", [(4, 60, 20, 3), (8, 300, 90, 5), (3, 10, 0, 2), (3, 0, 7, 3)])
def test_block_assembly_equals_dense_accumulation(ctx, oracle, n_kf, n_pt, n_ls, obs):
    """K7-K10: the block-form normal equations, expanded, equal the reference's dense H and g
    accumulation: landmark + cross blocks bit for bit (same summation order), keyframe blocks and err
    to rounding (fixed-shape two-level / tree sums)."""
    lm = synth.local_map(n_kf=n_kf, n_pt=n_pt, n_ls=n_ls, obs_per_lm=obs, seed=n_pt + 1)
    cam, ocam = _cams()
    nkf = n_kf - 1                                 # keyframe 0 is never optimised (:1231): kf_loc = slot - 1
    pt_kf_loc, ls_kf_loc = lm["pt_kf"] - 1, lm["ls_kf"] - 1
    rows_p = oracle.lba_point_rows(ocam, 1e-7, lm["T_kf_w"], lm["Xw"], lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
    rows_l = oracle.lba_line_rows(ocam, 1e-7, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"])
    H, g, e1 = oracle.lba_accumulate("points", nkf, n_pt, n_ls, lm["pt_lm"], pt_kf_loc, *rows_p)
    H, g, e2 = oracle.lba_accumulate("lines", nkf, n_pt, n_ls, lm["ls_lm"], ls_kf_loc, *rows_l, H=H, g=g)
    B = ctx.lba_assemble(nkf, n_pt, n_ls, lm["pt_lm"], pt_kf_loc, rows_p, lm["ls_lm"], ls_kf_loc, rows_l)
    _assert_blocks_match_dense(B, H, g, nkf, n_pt, n_ls, lm["pt_lm"], pt_kf_loc, lm["ls_lm"], ls_kf_loc)
    assert abs(B["err"] - (e1 + e2)) <= 1e-12 * abs(e1 + e2)
    if n_pt:
        with pytest.raises(plslam_amd.PlslamError):    # out-of-range keyframe slots are rejected, not read
            ctx.lba_assemble(nkf, n_pt, n_ls, lm["pt_lm"], pt_kf_loc + 100, rows_p, lm["ls_lm"], ls_kf_loc, rows_l)


def test_block_assembly_c3_size(ctx, oracle):
    """C3 scale (N = 42 054): the dense H of the reference would be 14 GB; check the blocks against
    per-block numpy sums instead."""
    lm = synth.local_map()
    cam, ocam = _cams()
    nkf, npt, nls = 9, 10000, 2000
    pkf, lkf = lm["pt_kf"] - 1, lm["ls_kf"] - 1
    rows_p = ctx.lba_point_rows(cam, 1e-7, lm["T_kf_w"], lm["Xw"], lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
    rows_l = ctx.lba_line_rows(cam, 1e-7, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"])
    B = ctx.lba_assemble(nkf, npt, nls, lm["pt_lm"], pkf, rows_p, lm["ls_lm"], lkf, rows_l)
    Jp, Jl, r, w = rows_p
    Hpt = np.zeros((npt, 3, 3))
    np.add.at(Hpt, lm["pt_lm"], Jl[:, :, None] * Jl[:, None, :] * w[:, None, None])
    assert np.allclose(B["H_pt"], Hpt, rtol=1e-12, atol=1e-300)
    Hp = np.zeros((nkf, 6, 6))
    sel = pkf >= 0
    np.add.at(Hp, pkf[sel], (Jp[:, :, None] * Jp[:, None, :] * w[:, None, None])[sel])
    Jp2, Jl2, r2, w2 = rows_l
    sel2 = lkf >= 0
    np.add.at(Hp, lkf[sel2], (Jp2[:, :, None] * Jp2[:, None, :] * w2[:, None, None])[sel2])
    assert np.allclose(B["H_pose"], Hp, rtol=1e-9)
    assert np.allclose(B["W_ls"][sel2], (Jl2[:, :, None] * Jp2[:, None, :] * w2[:, None, None])[sel2], rtol=1e-13)
    assert (B["W_ls"][~sel2] == 0).all()
    assert np.isclose(B["err"], (r * r * w).sum() + (r2 * r2 * w2).sum(), rtol=1e-12)
    gp = np.zeros(6 * nkf + 3 * npt + 6 * nls)
    np.add.at(gp, (6 * nkf + 3 * lm["pt_lm"][:, None] + np.arange(3)).ravel(), (Jl * (r * w)[:, None]).ravel())
    seg = slice(6 * nkf, 6 * nkf + 3 * npt)    # sums with cancellation: bound relative to the segment's scale
    assert np.max(np.abs(B["g"][seg] - gp[seg])) <= 1e-12 * np.max(np.abs(gp[seg]))


def test_committed_lba_goldens(ctx):
    """GPU vs the committed fixtures tests/golden/lba_golden.npz (no oracle involved at run time)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lba_golden.npz"))
    cam, _ = _cams()
    lm = {k[4:]: g[k] for k in g.files if k.startswith("map/")}
    got = ctx.lba_point_rows(cam, 1e-7, lm["T_kf_w"], lm["Xw"], lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
    for nm, a in zip(("J_pose", "J_lm", "r", "w"), got):
        assert _close(a, g[f"rows/pt/{nm}"], RTOL) and _close(a, g[f"rows/pt/{nm}"], 1e-12)
    got_l = ctx.lba_line_rows(cam, 1e-7, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"])
    for nm, a in zip(("J_pose", "J_lm", "r", "w"), got_l):
        assert _close(a, g[f"rows/ls/{nm}"], 1e-12)
    got_c = ctx.lba_line_rows(cam, 1e-3, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"], compat_iter_pass=True)
    for nm, a in zip(("J_pose", "J_lm", "r", "w"), got_c):
        assert _close(a, g[f"rows/ls_compat/{nm}"], 1e-12)
    B = ctx.lba_assemble(4, 120, 40, lm["pt_lm"], lm["pt_kf"] - 1, [g[f"rows/pt/{n}"] for n in ("J_pose", "J_lm", "r", "w")],
                         lm["ls_lm"], lm["ls_kf"] - 1, [g[f"rows/ls/{n}"] for n in ("J_pose", "J_lm", "r", "w")])
    _assert_blocks_match_dense(B, g["acc/H"], g["acc/g"], 4, 120, 40, lm["pt_lm"], lm["pt_kf"] - 1, lm["ls_lm"],
                               lm["ls_kf"] - 1)
    for kind in ("points", "lines"):
        s = {k.split("/")[2]: g[k] for k in g.files if k.startswith(f"drv/{kind}/")}
        m, n = ctx.map2kf_match(kind, cam, s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"],
                                0.9, True, 1.0, 10)
        assert np.array_equal(m, s["map_to_kf"]) and n == int(s["n"][0])
        vis = (ctx.map_line_visible if kind == "lines" else ctx.map_point_visible)(cam, s["Twf"], s["LM"])
        assert np.array_equal(vis, s["visible"])


def test_lba_plan_iterations_match_oracle(ctx, oracle):
    """plslam_lba_plan: lists uploaded once, iterate() per LM iteration with changed poses/landmarks;
    blocks and rows equal the oracle's rows + dense accumulation (bit-exact), first pass and
    iteration-pass (compat) line behaviour."""
    lm = synth.local_map(n_kf=6, n_pt=400, n_ls=120, obs_per_lm=4, seed=3)
    cam, ocam = _cams()
    nkf, npt, nls = 5, 400, 120
    pkf, lkf = lm["pt_kf"] - 1, lm["ls_kf"] - 1
    plan = plslam_amd.LbaPlan(ctx, cam, 1e-7, 6, nkf, npt, nls, lm["pt_lm"], lm["pt_kf"], pkf, lm["obs_uv"],
                              lm["ls_lm"], lm["ls_kf"], lkf, lm["l_obs"])
    r = np.random.Generator(np.random.PCG64(77))
    T, X, L = lm["T_kf_w"].copy(), lm["Xw"].copy(), lm["Lw"].copy()
    for it in range(3):
        compat = it > 0                                   # iterations after the first pass (:1668-1748)
        B = plan.iterate(T, X, L, compat_iter_pass=compat)
        rp = oracle.lba_point_rows(ocam, 1e-7, T, X, lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
        rl = oracle.lba_line_rows(ocam, 1e-7, T, L, lm["l_obs"], lm["ls_lm"], lm["ls_kf"], compat_iter_pass=compat)
        H, g, e1 = oracle.lba_accumulate("points", nkf, npt, nls, lm["pt_lm"], pkf, *rp)
        H, g, e2 = oracle.lba_accumulate("lines", nkf, npt, nls, lm["ls_lm"], lkf, *rl, H=H, g=g)
        gp, gl = plan.rows()
        for a, b in zip(gp + gl, list(rp) + list(rl)):
            assert _close(a, b, 1e-12)
        if all(np.array_equal(a, b) for a, b in zip(gp + gl, list(rp) + list(rl))):   # rows bit-equal => blocks too
            _assert_blocks_match_dense(B, H, g, nkf, npt, nls, lm["pt_lm"], pkf, lm["ls_lm"], lkf)
        else:
            assert np.allclose(B["g"], g, rtol=1e-9, atol=1e-9 * np.abs(g).max())
        assert abs(B["err"] - (e1 + e2)) <= 1e-12 * abs(e1 + e2)
        # an "LM step": perturb landmarks and the optimised poses
        X = X + 1e-3 * r.standard_normal(X.shape)
        L = L + 1e-3 * r.standard_normal(L.shape)
        for k in range(1, 6):
            T[k] = (T[k].reshape(4, 4) @ np.linalg.inv(synth.se3_exp(1e-3 * r.standard_normal(6)))).reshape(16)
    plan.close()
    with pytest.raises(plslam_amd.PlslamError):
        plslam_amd.LbaPlan(ctx, cam, 1e-7, 6, nkf, npt, nls, lm["pt_lm"], lm["pt_kf"] + 9, pkf, lm["obs_uv"],
                           lm["ls_lm"], lm["ls_kf"], lkf, lm["l_obs"])


def test_lba_plan_against_the_reference_source_text_outputs(ctx):
    """tests/golden/lba_ref_golden.npz = dense H, g, err produced by the reference's OWN observation loops
    (src/mapHandler.cpp:1358-1540 first pass, :1587-1772 iteration pass; compiled textually, see
    tests/golden/make_lba_ref_golden.py).  One device plan, used as INTEGRATION.md says: pose slots 0..n_kf-1 hold the
    stored key-frame poses, slots n_kf.. the current estimates; point observations of optimised key frames read the
    estimates, line observations ALWAYS the stored poses (:1680), iteration pass with compat (stride-3 end points,
    literal 1e-7).  Block-form output, expanded, vs the reference: 1e-10 of the matrix scale."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lba_ref_golden.npz"))
    n_kf, nkf, npt, nls = (int(x) for x in g["dims"])
    cam, _ = _cams()
    kp, kl = g["pt_kf"] - 1, g["ls_kf"] - 1
    slot_p = np.where(kp >= 0, n_kf + kp, g["pt_kf"]).astype(np.int32)
    plan = plslam_amd.LbaPlan(ctx, cam, float(g["th"][0]), n_kf + nkf, nkf, npt, nls, g["pt_lm"], slot_p, kp, g["obs_uv"],
                              g["ls_lm"], g["ls_kf"], kl, g["l_obs"])
    for name, T_est, compat in (("first", g["T_map"][1:], False), ("iter", g["T_slot"], True)):
        B = plan.iterate(np.concatenate([g["T_map"], T_est]), g["Xw"], g["Lw"], compat_iter_pass=compat)
        Hd = _expand_blocks(B, nkf, npt, nls, g["pt_lm"], kp, g["ls_lm"], kl)
        H, gg = g[f"{name}_H"], g[f"{name}_g"]
        assert np.abs(Hd - H).max() <= 1e-10 * np.abs(H).max(), name
        assert np.abs(B["g"] - gg).max() <= 1e-10 * np.abs(gg).max(), name
        assert abs(B["err"] - float(g[f"{name}_err"][0])) <= 1e-11 * float(g[f"{name}_err"][0]), name
    assert np.abs(g["iter_H"] - g["first_H"]).max() > 1e-6 * np.abs(g["first_H"]).max()     # the passes differ
    plan.close()


def test_lba_plan_device_resident_iteration_and_gba_blocks(ctx):
    """plslam_lba_plan_iterate_dev: the same iteration with the blocks left on the device -- what comes back (err, g) and
    what plslam_lba_plan_blocks fetches afterwards are bit-identical to plslam_lba_plan_iterate's; at C3 size the
    host-to-host iteration is several times faster because 11.5 MB of blocks stay put.  PLSLAM_LBA_COMPAT_GBA: the
    pose x line cross blocks come out TRANSPOSED (how levMarquardtOptimizationGBA writes them, src/mapHandler.cpp:2341-2352
    vs :1531-1532) and nothing else changes; the goldens of the reference's own loops hold for the _dev form too."""
    import os
    import time
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "lba_ref_golden.npz"))
    n_kf, nkf, npt, nls = (int(x) for x in g["dims"])
    cam, _ = _cams()
    kp, kl = g["pt_kf"] - 1, g["ls_kf"] - 1
    slot_p = np.where(kp >= 0, n_kf + kp, g["pt_kf"]).astype(np.int32)
    plan = plslam_amd.LbaPlan(ctx, cam, float(g["th"][0]), n_kf + nkf, nkf, npt, nls, g["pt_lm"], slot_p, kp, g["obs_uv"],
                              g["ls_lm"], g["ls_kf"], kl, g["l_obs"])
    T = np.concatenate([g["T_map"], g["T_slot"]])
    ref = plan.iterate(T, g["Xw"], g["Lw"], compat_iter_pass=1)
    err, gg = plan.iterate_dev(T, g["Xw"], g["Lw"], compat_flags=plan.COMPAT_ITER_PASS)
    got = plan.blocks()
    assert err == ref["err"] and np.array_equal(gg, ref["g"])
    for k in ("g", "H_pose", "H_pt", "H_ls", "W_pt", "W_ls"):
        assert np.array_equal(got[k], ref[k]), k
    Hd = _expand_blocks(got, nkf, npt, nls, g["pt_lm"], kp, g["ls_lm"], kl)
    assert np.abs(Hd - g["iter_H"]).max() <= 1e-10 * np.abs(g["iter_H"]).max()
    # GBA form: only W_ls changes, into the per-observation transposes
    err2, g2 = plan.iterate_dev(T, g["Xw"], g["Lw"], compat_flags=plan.COMPAT_ITER_PASS | plan.COMPAT_GBA)
    gba = plan.blocks()
    assert err2 == err and np.array_equal(g2, gg)
    for k in ("H_pose", "H_pt", "H_ls", "W_pt"):
        assert np.array_equal(gba[k], ref[k]), k
    assert np.array_equal(gba["W_ls"], np.transpose(ref["W_ls"], (0, 2, 1)))
    assert not np.array_equal(gba["W_ls"], ref["W_ls"])
    plan.close()
    # C3 size: host-to-host time of one iteration, blocks downloaded vs left on the device
    lm = synth.local_map()
    big = plslam_amd.LbaPlan(ctx, cam, 1e-7, 10, 9, 10000, 2000, lm["pt_lm"], lm["pt_kf"], lm["pt_kf"] - 1, lm["obs_uv"],
                             lm["ls_lm"], lm["ls_kf"], lm["ls_kf"] - 1, lm["l_obs"])
    for _ in range(3):
        big.iterate(lm["T_kf_w"], lm["Xw"], lm["Lw"])
        big.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"], want_g=False)
    t0 = time.perf_counter()
    for _ in range(10):
        big.iterate(lm["T_kf_w"], lm["Xw"], lm["Lw"])
    t_host = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(10):
        e_dev, _ = big.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"], want_g=False)
    t_dev = (time.perf_counter() - t0) / 10
    assert e_dev == big.iterate(lm["T_kf_w"], lm["Xw"], lm["Lw"])["err"]
    print(f"C3 LBA iteration host-to-host: blocks downloaded {1e3 * t_host:.2f} ms, device-resident {1e3 * t_dev:.2f} ms")
    assert t_dev < 0.6 * t_host and t_dev < 1.0e-3
    # the state resident too (round 4): no upload -- the same err and the same blocks; after an in-place update of the device
    # copy of Xw (what a device-side solver does) the iteration equals the uploaded one on the updated landmarks
    import torch
    ref_blocks = big.blocks()
    assert big.iterate_resident() == e_dev
    again = big.blocks()
    for k in ("g", "H_pose", "H_pt", "H_ls", "W_pt", "W_ls"):
        assert np.array_equal(again[k], ref_blocks[k]), k
    st = big.device_state()
    assert (st["n_pose_slots"], st["npt"], st["nls"]) == (10, 10000, 2000)
    X2 = lm["Xw"] + np.random.Generator(np.random.PCG64(3)).normal(0, 1e-3, lm["Xw"].shape)
    import ctypes
    torch.cuda.synchronize()
    dX = torch.from_numpy(X2).to(torch.device("cuda", 0))
    assert plslam_amd.capi._hip_memcpy_dtod(st["Xw"], dX.data_ptr(), X2.nbytes) == 0
    e_res = big.iterate_resident()
    b_res = big.blocks()
    e_up, _ = big.iterate_dev(lm["T_kf_w"], X2, lm["Lw"], want_g=False)
    b_up = big.blocks()
    assert e_res == e_up and e_res != e_dev
    for k in ("g", "H_pose", "H_pt", "H_ls", "W_pt", "W_ls"):
        assert np.array_equal(b_res[k], b_up[k]), k
    t0 = time.perf_counter()
    for _ in range(20):
        big.iterate_resident()
    print(f"state resident: {1e6 * (time.perf_counter() - t0) / 20:.0f} us per iteration")
    big.close()
    with pytest.raises(plslam_amd.PlslamError):           # no state on the device yet
        fresh = plslam_amd.LbaPlan(ctx, cam, 1e-7, 10, 9, 10000, 2000, lm["pt_lm"], lm["pt_kf"], lm["pt_kf"] - 1, lm["obs_uv"],
                                   lm["ls_lm"], lm["ls_kf"], lm["ls_kf"] - 1, lm["l_obs"])
        try:
            fresh.iterate_resident()
        finally:
            fresh.close()


def test_schur_step_on_the_resident_blocks_equals_the_dense_damped_solve(ctx, oracle):
    """plslam_lba_plan_diag_max / _schur / _backsub / _set_poses (round 5): the solve of src/mapHandler.cpp:1544-1575 with only
    the 6 nkf x 6 nkf reduced system leaving the device.  Checked against numpy on the DENSE system the oracle accumulates as the
    reference does (H(i,i) += lambda H(i,i), then a dense solve): the reduced system S, b against the dense Schur complement,
    the pose step and every landmark step against the dense solution, then the in-place update of the resident landmarks +
    uploaded poses against an iteration on the same state uploaded whole.  Observations by the fixed keyframe (kf_loc = -1), a
    landmark seen by the fixed keyframe only, a landmark without any observation (singular block: counted, zero step)."""
    lm = synth.local_map(n_kf=6, n_pt=400, n_ls=120, obs_per_lm=4, seed=11)
    cam, ocam = _cams()
    nkf, npt, nls = 5, 400, 120
    pt_lm, pt_kf, ls_lm, ls_kf = lm["pt_lm"].copy(), lm["pt_kf"].copy(), lm["ls_lm"].copy(), lm["ls_kf"].copy()
    pt_kf[pt_lm == 7] = 0                                   # landmark 7: seen by the fixed keyframe only
    pkf, lkf = pt_kf - 1, ls_kf - 1
    T, X, L = lm["T_kf_w"].copy(), lm["Xw"].copy(), lm["Lw"].copy()
    plan = plslam_amd.LbaPlan(ctx, cam, 1e-7, 6, nkf, npt, nls, pt_lm, pt_kf, pkf, lm["obs_uv"], ls_lm, ls_kf, lkf, lm["l_obs"])
    with pytest.raises(plslam_amd.PlslamError):
        plan.schur(1e-3)                                    # no iteration yet: no blocks
    B = plan.iterate(T, X, L)
    rp = oracle.lba_point_rows(ocam, 1e-7, T, X, lm["obs_uv"], pt_lm, pt_kf)
    rl = oracle.lba_line_rows(ocam, 1e-7, T, L, lm["l_obs"], ls_lm, ls_kf)
    H, g, _ = oracle.lba_accumulate("points", nkf, npt, nls, pt_lm, pkf, *rp)
    H, g, _ = oracle.lba_accumulate("lines", nkf, npt, nls, ls_lm, lkf, *rl, H=H, g=g)
    hmax = plan.diag_max()
    assert abs(hmax - np.abs(np.diag(H)).max()) <= 1e-12 * hmax
    lam = 1e-5 * hmax if hmax < 1e3 else 1e-3               # (the reference: lambda0 * Hmax; any positive value does)
    lam = 1e-3
    Hd = H.copy()
    Hd[np.diag_indices_from(Hd)] *= 1.0 + lam
    n6 = 6 * nkf
    Hpp, Hpl, Hll = Hd[:n6, :n6], Hd[:n6, n6:], Hd[n6:, n6:]
    Vi = np.linalg.inv(Hll)
    S_ref, b_ref = Hpp - Hpl @ Vi @ Hpl.T, g[:n6] - Hpl @ Vi @ g[n6:]
    S, b, ns = plan.schur(lam)
    assert ns == 0
    assert np.allclose(S, S_ref, rtol=0, atol=1e-9 * np.abs(S_ref).max()) and np.allclose(b, b_ref, rtol=0, atol=1e-9 * np.abs(b_ref).max())
    assert np.allclose(S, S.T, rtol=0, atol=1e-12 * np.abs(S).max())
    DX = np.linalg.solve(Hd, g)
    dp = np.linalg.solve(S, b)
    dxp, dxl = plan.backsub(dp)
    got = np.concatenate([dp, dxp.reshape(-1), dxl.reshape(-1)])
    assert np.allclose(got, DX, rtol=0, atol=1e-7 * np.abs(DX).max()), np.abs(got - DX).max() / np.abs(DX).max()
    # landmark 7 has no cross block: its step is its own block's solve
    V7 = B["H_pt"][7] * (1 + lam * np.eye(3))
    assert np.allclose(dxp[7], np.linalg.solve(V7, B["g"][n6 + 21:n6 + 24]), rtol=1e-9)
    # blocks written the GBA way (pose x line cross blocks transposed) are refused
    plan.iterate_dev(T, X, L, compat_flags=plslam_amd.LbaPlan.COMPAT_GBA, want_g=False)
    with pytest.raises(plslam_amd.PlslamError):
        plan.schur(lam)
    plan.iterate_dev(T, X, L, want_g=False)
    # determinism: the same call again gives the same words
    S2, b2, _ = plan.schur(lam)
    assert np.array_equal(S, S2) and np.array_equal(b, b2)
    # the update in place: landmarks on the device, poses by the host -> the resident iteration = the uploaded one
    plan.backsub(dp, apply=True, want=False)
    with pytest.raises(plslam_amd.PlslamError):
        plan.backsub(dp, apply=True, want=False)            # the step has been applied: a new schur() is needed
    T2 = T.copy()
    for k in range(nkf):
        T2[k + 1] = (T[k + 1].reshape(4, 4) @ np.linalg.inv(synth.se3_exp(dp[6 * k:6 * k + 6]))).reshape(16)
    plan.set_poses(T2)
    e_res = plan.iterate_resident()
    blocks_res = plan.blocks()
    e_up, _ = plan.iterate_dev(T2, X + dxp, L + dxl, want_g=False)
    blocks_up = plan.blocks()
    assert e_res == e_up
    for k in ("g", "H_pose", "H_pt", "H_ls", "W_pt", "W_ls"):
        assert np.array_equal(blocks_res[k], blocks_up[k]), k
    plan.close()
    # a landmark nobody observes: its block is zero -- counted, no contribution, zero step
    plan = plslam_amd.LbaPlan(ctx, cam, 1e-7, 6, nkf, npt + 1, nls, pt_lm, pt_kf, pkf, lm["obs_uv"], ls_lm, ls_kf, lkf, lm["l_obs"])
    plan.iterate(T, np.vstack([X, [[0.0, 0.0, 5.0]]]), L)
    S3, b3, ns3 = plan.schur(lam)
    assert ns3 == 1 and np.allclose(S3, S, rtol=0, atol=1e-12 * np.abs(S).max()) and np.allclose(b3, b, rtol=0, atol=1e-12 * np.abs(b).max())
    # (the counter of singular blocks alternates between two words, each call clearing the next one's: every call counts afresh)
    for _ in range(3):
        S4, b4, ns4 = plan.schur(lam)
        assert ns4 == 1 and np.array_equal(S4, S3) and np.array_equal(b4, b3)
    dxp3, _ = plan.backsub(dp)
    assert np.array_equal(dxp3[npt], np.zeros(3)) and np.allclose(dxp3[:npt], dxp, rtol=0, atol=1e-12 * np.abs(dxp).max())
    plan.close()


def test_schur_step_at_c3_size(ctx):
    """The C3 map (9 optimised keyframes, 10 000 points, 2 000 lines, 60 000 observations): the reduced system against a
    float64 numpy Schur complement built from the downloaded blocks, and what one LM iteration costs host to host with the
    blocks staying on the device (iterate + diag_max once + schur + the host's 54 x 54 solve + backsub with the update)."""
    import time
    lm = synth.local_map()
    cam, _ = _cams()
    nkf, npt, nls = 9, 10000, 2000
    pkf, lkf = lm["pt_kf"] - 1, lm["ls_kf"] - 1
    plan = plslam_amd.LbaPlan(ctx, cam, 1e-7, 10, nkf, npt, nls, lm["pt_lm"], lm["pt_kf"], pkf, lm["obs_uv"], lm["ls_lm"], lm["ls_kf"],
                              lkf, lm["l_obs"])
    plan.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"], want_g=False)
    Bk = plan.blocks()
    lam = 1e-3
    n6 = 6 * nkf
    S_ref = np.zeros((n6, n6))
    b_ref = Bk["g"][:n6].copy()
    for k in range(nkf):
        S_ref[6 * k:6 * k + 6, 6 * k:6 * k + 6] = Bk["H_pose"][k] * (1 + lam * np.eye(6))
    for lm_of, kf_of, Hb, Wb, goff, dl in ((lm["pt_lm"], pkf, Bk["H_pt"], Bk["W_pt"], n6, 3),
                                           (lm["ls_lm"], lkf, Bk["H_ls"], Bk["W_ls"], n6 + 3 * npt, 6)):
        Vi = np.linalg.inv(Hb * (1 + lam * np.eye(dl))[None])
        order = np.argsort(lm_of, kind="stable")
        start = np.searchsorted(lm_of[order], np.arange(Hb.shape[0] + 1))
        for j in range(Hb.shape[0]):
            obs = [o for o in order[start[j]:start[j + 1]] if kf_of[o] >= 0]
            gj = Bk["g"][goff + dl * j:goff + dl * j + dl]
            for o1 in obs:
                k1 = kf_of[o1]
                Y = Wb[o1].T @ Vi[j]                       # 6 x dl
                b_ref[6 * k1:6 * k1 + 6] -= Y @ gj
                for o2 in obs:
                    k2 = kf_of[o2]
                    S_ref[6 * k1:6 * k1 + 6, 6 * k2:6 * k2 + 6] -= Y @ Wb[o2]
    S, b, ns = plan.schur(lam)
    assert ns == 0
    assert np.allclose(S, S_ref, rtol=0, atol=1e-9 * np.abs(S_ref).max()) and np.allclose(b, b_ref, rtol=0, atol=1e-9 * np.abs(b_ref).max())
    dp = np.linalg.solve(S, b)
    for _ in range(3):
        plan.iterate_resident(); plan.schur(lam); plan.backsub(dp, apply=False, want=False)
    t0 = time.perf_counter()
    for _ in range(20):
        plan.iterate_resident()
        S, b, _ = plan.schur(lam)
        dp = np.linalg.solve(S, b)
        plan.backsub(dp, apply=False, want=False)
    print(f"C3: one LM iteration with the Schur step, blocks resident: {1e6 * (time.perf_counter() - t0) / 20:.0f} us host to host")
    plan.close()


def test_visibility_gates_and_median_descriptor_against_reference_source_text_outputs(ctx):
    """tests/golden/map2kf_ref_golden.npz = what the reference's OWN loops produce (matchMap2KFPoints / Lines visibility
    pre-filter and gates, src/mapHandler.cpp:545-558, :601-629, :647-663, :716-749, compiled textually; and
    updateAverageDescDir, src/mapFeatures.cpp:51-93 compiled as is -- tests/golden/make_map2kf_ref_golden.py).  The
    device entry points must reproduce every mask, count and index."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "map2kf_ref_golden.npz"))
    cam, _ = _cams()
    vp, vl = ctx.map_point_visible(cam, g["Twf"], g["X"]), ctx.map_line_visible(cam, g["Twf"], g["Lw"])
    assert np.array_equal(vp, g["vis_p"]) and np.array_equal(vl, g["vis_l"])
    Xv, Lv = g["X"][vp.astype(bool)], g["Lw"][vl.astype(bool)]
    for k, th in enumerate(g["th_p"]):
        mask, cnt = ctx.map2kf_point_gate(cam, g["Twf"], Xv, g["m12p"], g["pl"], float(th))
        assert np.array_equal(mask, g[f"gate_p{k}"]) and cnt == int(g[f"count_p{k}"][0])
    for k, th in enumerate(g["th_l"]):
        mask, cnt = ctx.map2kf_line_gate(cam, g["Twf"], Lv, g["m12l"], g["le"], float(th))
        assert np.array_equal(mask, g[f"gate_l{k}"]) and cnt == int(g[f"count_l{k}"][0])
    idx, md = ctx.median_desc_batched(g["med_desc_lists"], g["med_offsets"])
    assert np.array_equal(idx, g["med_idx"])
    assert np.array_equal(md, g["med_desc_lists"][g["med_offsets"][:-1] + g["med_idx"]])


def test_iteration_on_the_plans_page_locked_images_equals_the_staged_call(ctx):
    """plslam_lba_plan_host_state: a caller that keeps T / Xw / Lw in the plan's page-locked images and reads g there gets the
    bits of the staged call (which copies the caller's arrays into those images first); pointers of any other origin keep
    working array by array, and an update written into the images is what the next call sees."""
    lm = synth.local_map(n_kf=6, n_pt=700, n_ls=150, seed=23)
    cam, _ = _cams()
    nkf = 5
    pkf, lkf = lm["pt_kf"] - 1, lm["ls_kf"] - 1
    plan = plslam_amd.LbaPlan(ctx, cam, 1e-7, 6, nkf, 700, 150, lm["pt_lm"], lm["pt_kf"], pkf, lm["obs_uv"], lm["ls_lm"], lm["ls_kf"],
                              lkf, lm["l_obs"])
    err, g = plan.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"])
    hs = plan.host_state()
    assert hs["T_kf_w"].shape == (6, 16) and hs["Xw"].shape == (700, 3) and hs["Lw"].shape == (150, 6) and hs["g"].shape == g.shape
    # (the staged call has just left the caller's state in the images)
    assert np.array_equal(hs["Xw"], lm["Xw"]) and np.array_equal(hs["T_kf_w"].ravel(), np.asarray(lm["T_kf_w"]).ravel())
    hs["g"][:] = 0
    err2, g2 = plan.iterate_dev(hs["T_kf_w"], hs["Xw"], hs["Lw"], g_out=hs["g"])
    assert g2 is hs["g"] and err2 == err and np.array_equal(g2, g)
    # mixed origins: the landmarks from the image, the poses and g from ordinary memory
    err3, g3 = plan.iterate_dev(lm["T_kf_w"], hs["Xw"], hs["Lw"])
    assert err3 == err and np.array_equal(g3, g)
    # an update in place
    rng = np.random.default_rng(5)
    dX = 1e-3 * rng.standard_normal((700, 3))
    hs["Xw"] += dX
    err4, g4 = plan.iterate_dev(hs["T_kf_w"], hs["Xw"], hs["Lw"], g_out=hs["g"])
    g4 = g4.copy()
    err5, g5 = plan.iterate_dev(lm["T_kf_w"], lm["Xw"] + dX, lm["Lw"])
    assert err4 == err5 and err4 != err and np.array_equal(g4, g5)
    # an err-only iteration (a rejected LM trial step) leaves the gradient of the image alone (it landed on g[0])
    err6, none6 = plan.iterate_dev(hs["T_kf_w"], hs["Xw"], hs["Lw"], want_g=False)
    assert none6 is None and err6 == err4 and np.array_equal(hs["g"], g4)
    assert plan.iterate_resident() == err4 and np.array_equal(hs["g"], g4)
    plan.close()


@pytest.mark.parametrize("n_kf,n_pt,n_ls,obs", [
    (6, 300, 0, 4),        # points only: no line workgroups in any of the fused launches
    (6, 0, 90, 4),         # lines only
    (2, 200, 40, 2),       # ONE optimised keyframe: a single block of S, one pair per landmark
    (4, 1500, 300, 4),     # every block's pair list spans several chunks of 64 (the two-level sums)
    (3, 1, 1, 3),          # a workgroup with two live lanes
])
def test_schur_step_edge_shapes(ctx, oracle, n_kf, n_pt, n_ls, obs):
    """The Schur step against the dense damped solve over shapes that leave parts of its fused launches empty or make its
    chunked sums long (plslam_lba_plan_schur / _backsub; the reference: src/mapHandler.cpp:1552-1575)."""
    lm = synth.local_map(n_kf=n_kf, n_pt=n_pt, n_ls=n_ls, obs_per_lm=min(obs, n_kf), seed=100 + n_pt + n_ls)
    cam, ocam = _cams()
    nkf = n_kf - 1
    pkf, lkf = lm["pt_kf"] - 1, lm["ls_kf"] - 1
    T, X, L = lm["T_kf_w"], lm["Xw"], lm["Lw"]
    plan = plslam_amd.LbaPlan(ctx, cam, 1e-7, n_kf, nkf, n_pt, n_ls, lm["pt_lm"], lm["pt_kf"], pkf, lm["obs_uv"], lm["ls_lm"], lm["ls_kf"],
                              lkf, lm["l_obs"])
    plan.iterate_dev(T, X, L, want_g=False)
    N = 6 * nkf + 3 * n_pt + 6 * n_ls
    H, g = np.zeros((N, N)), np.zeros(N)
    if n_pt:
        rp = oracle.lba_point_rows(ocam, 1e-7, T, X, lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])
        H, g, _ = oracle.lba_accumulate("points", nkf, n_pt, n_ls, lm["pt_lm"], pkf, *rp, H=H, g=g)
    if n_ls:
        rl = oracle.lba_line_rows(ocam, 1e-7, T, L, lm["l_obs"], lm["ls_lm"], lm["ls_kf"])
        H, g, _ = oracle.lba_accumulate("lines", nkf, n_pt, n_ls, lm["ls_lm"], lkf, *rl, H=H, g=g)
    lam = 1e-3
    Hd = H.copy()
    Hd[np.diag_indices_from(Hd)] *= 1.0 + lam
    n6 = 6 * nkf
    Vi = np.linalg.inv(Hd[n6:, n6:])
    S_ref, b_ref = Hd[:n6, :n6] - Hd[:n6, n6:] @ Vi @ Hd[n6:, :n6], g[:n6] - Hd[:n6, n6:] @ Vi @ g[n6:]
    for rep in range(2):                                   # (twice: the two counters of singular blocks take turns)
        S, b, ns = plan.schur(lam)
        assert ns == 0
        assert np.allclose(S, S_ref, rtol=0, atol=1e-9 * np.abs(S_ref).max()) and np.allclose(b, b_ref, rtol=0, atol=1e-9 * np.abs(b_ref).max())
    DX = np.linalg.solve(Hd, g)
    dp = np.linalg.solve(S, b)
    dxp, dxl = plan.backsub(dp)
    got = np.concatenate([dp, dxp.reshape(-1), dxl.reshape(-1)])
    assert np.allclose(got, DX, rtol=0, atol=1e-7 * np.abs(DX).max()), np.abs(got - DX).max() / np.abs(DX).max()
    plan.close()


def test_fused_iteration_calls_equal_the_separate_ones(ctx):
    """plslam_lba_plan_iterate_schur / _apply_step (round 6): an LM iteration in two calls and two synchronisations.  Two plans
    on the same map run three iterations, one through iterate_resident + schur + backsub + set_poses, the other through the
    fused calls: the same words for err, S, b, the resident landmarks after each step, and the sum of squares of the landmark
    steps equal to numpy's on the downloaded steps (to rounding: the device sums in a tree).  A rejected step (apply = False,
    poses left) changes nothing resident; a step applied twice and blocks written the GBA way are refused."""
    lm = synth.local_map(n_kf=6, n_pt=700, n_ls=150, obs_per_lm=4, seed=23)
    cam, _ = _cams()
    nkf, npt, nls = 5, 700, 150
    pkf, lkf = lm["pt_kf"] - 1, lm["ls_kf"] - 1
    T, X, L = lm["T_kf_w"].copy(), lm["Xw"].copy(), lm["Lw"].copy()
    mk = lambda: plslam_amd.LbaPlan(ctx, cam, 1e-7, 6, nkf, npt, nls, lm["pt_lm"], lm["pt_kf"], pkf, lm["obs_uv"],
                                    lm["ls_lm"], lm["ls_kf"], lkf, lm["l_obs"])
    a, f = mk(), mk()
    with pytest.raises(plslam_amd.PlslamError):
        f.iterate_schur(1e-3)                               # no resident state yet
    a.iterate_dev(T, X, L, want_g=False)
    f.iterate_dev(T, X, L, want_g=False)
    lam, Tcur = 1e-3, T.copy()
    for it in range(3):
        flags = plslam_amd.LbaPlan.COMPAT_ITER_PASS if it else 0
        e_a = a.iterate_resident(flags)
        S_a, b_a, ns_a = a.schur(lam)
        e_f, S_f, b_f, ns_f = f.iterate_schur(lam, flags)
        assert e_a == e_f and ns_a == ns_f and np.array_equal(S_a, S_f) and np.array_equal(b_a, b_f), it
        dp = np.linalg.solve(S_a, b_a)
        apply = it != 1                                     # the middle step is "rejected"
        dxp, dxl = a.backsub(dp, apply=apply)
        Tn = None
        if apply:
            Tn = Tcur.copy()
            for k in range(nkf):
                Tn[k + 1] = (Tcur[k + 1].reshape(4, 4) @ np.linalg.inv(synth.se3_exp(dp[6 * k:6 * k + 6]))).reshape(16)
            a.set_poses(Tn)
            Tcur = Tn
        ss = f.apply_step(dp, Tn, apply=apply)
        want = float((dxp ** 2).sum() + (dxl ** 2).sum())
        assert abs(ss - want) <= 1e-13 * want, (ss, want)
        Xa, La = a.get_landmarks()
        Xf, Lf = f.get_landmarks()
        assert np.array_equal(Xa, Xf) and np.array_equal(La, Lf), it
        if apply:
            with pytest.raises(plslam_amd.PlslamError):
                f.apply_step(dp, Tn)                        # applied: a new Schur step is needed
        lam *= 10.0
    assert a.iterate_resident(plslam_amd.LbaPlan.COMPAT_ITER_PASS) == f.iterate_resident(plslam_amd.LbaPlan.COMPAT_ITER_PASS)
    with pytest.raises(plslam_amd.PlslamError):
        f.iterate_schur(lam, plslam_amd.LbaPlan.COMPAT_GBA)
    # determinism of the sum: the same step again -> the same word
    e1, S1, b1, _ = f.iterate_schur(lam, 0)
    dp = np.linalg.solve(S1, b1)
    s1 = f.apply_step(dp, None, apply=False)
    s2 = f.apply_step(dp, None, apply=False)
    assert s1 == s2
    a.close(); f.close()


@pytest.mark.parametrize("n_kf,n_pt,n_ls,obs", [
    (6, 300, 0, 4),        # points only: no line workgroups, no line chunk of the pair lists
    (6, 0, 90, 4),         # lines only: every chunk of the pair lists is a line chunk
    (2, 200, 40, 2),       # ONE optimised keyframe
    (4, 1500, 300, 4),     # point chunks padded by null pairs in front of the line chunks of every block
    (3, 1, 1, 3),          # a workgroup with two live lanes
    (31, 900, 200, 6),     # 30 optimised key frames: a 180 x 180 reduced system written in place, 465 blocks
])
def test_fused_iteration_edge_shapes(ctx, n_kf, n_pt, n_ls, obs):
    """plslam_lba_plan_iterate_schur / _apply_step over the shapes of test_schur_step_edge_shapes (and a wide one): the same words
    as iterate_resident + schur + backsub, twice in a row (the singular-block counters alternate), with and without the update."""
    lm = synth.local_map(n_kf=n_kf, n_pt=n_pt, n_ls=n_ls, obs_per_lm=obs, seed=77 + n_kf)
    cam, _ = _cams()
    nkf = n_kf - 1
    mk = lambda: plslam_amd.LbaPlan(ctx, cam, 1e-7, n_kf, nkf, n_pt, n_ls, lm["pt_lm"], lm["pt_kf"], lm["pt_kf"] - 1, lm["obs_uv"],
                                    lm["ls_lm"], lm["ls_kf"], lm["ls_kf"] - 1, lm["l_obs"])
    a, f = mk(), mk()
    for p in (a, f):
        p.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"], want_g=False)
    lam = 1e-3
    for it in range(3):
        e_a = a.iterate_resident(plslam_amd.LbaPlan.COMPAT_ITER_PASS)
        S_a, b_a, ns_a = a.schur(lam)
        e_f, S_f, b_f, ns_f = f.iterate_schur(lam, plslam_amd.LbaPlan.COMPAT_ITER_PASS)
        assert e_a == e_f and ns_a == ns_f and np.array_equal(S_a, S_f) and np.array_equal(b_a, b_f), it
        dp = np.linalg.solve(S_a, b_a) * 0.5
        dxp, dxl = a.backsub(dp, apply=(it != 1))
        ss = f.apply_step(dp, None, apply=(it != 1))
        want = float((dxp ** 2).sum() + (dxl ** 2).sum())
        assert abs(ss - want) <= 1e-12 * max(want, 1e-300)
        Xa, La = a.get_landmarks()
        Xf, Lf = f.get_landmarks()
        assert np.array_equal(Xa, Xf) and np.array_equal(La, Lf), it
        lam *= 3.0
    a.close(); f.close()
