"""GPU parity tests proper (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle
on identical seeded inputs -- bit-exact for indices, distances, match tables and counts -- plus the
committed golden vectors and size-independent properties at BASELINE.json's full sizes."""
import os

import numpy as np
import pytest

from plslam_amd import frontend, synth
from conftest import set_mfma_form

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "match_golden.npz")


def _rng(seed=0):
    return np.random.Generator(np.random.PCG64(seed))


def test_library_is_the_hip_one(ctx):
    info = ctx.device_info()
    assert "gfx950" in info["name"], info
    assert info["cu_count"] >= 200


def test_golden_vectors(ctx):
    g = np.load(GOLD)
    for n in sorted({k.split("/")[0] for k in g.files}):
        q, t = g[f"{n}/q"], g[f"{n}/t"]
        idx, dist = ctx.knn2(q, t)
        assert np.array_equal(idx, g[f"{n}/knn_idx"]), n
        assert np.array_equal(dist, g[f"{n}/knn_dist"]), n
        assert np.array_equal(np.take_along_axis(g[f"{n}/dist_ref_bitops"], np.clip(idx, 0, None), 1)[idx >= 0],
                              dist[idx >= 0]), n   # distances == the reference's own popcount code
        for nnr in (0.6, 0.75, 0.9):
            for mut in (0, 1):
                m, cnt = ctx.match(q, t, nnr, bool(mut))
                exp = g[f"{n}/m12_nnr{nnr}_mut{mut}"]
                assert np.array_equal(m, exp), (n, nnr, mut)
                assert cnt == int((exp >= 0).sum())


@pytest.mark.parametrize("nq,nt", [(0, 5), (5, 0), (3, 1), (3, 2), (1, 1), (64, 64), (65, 129), (255, 3),
                                   (256, 256), (257, 511), (1000, 4), (7, 1025)])
def test_knn2_and_match_edge_sizes(vctx, oracle, nq, nt):
    ctx = vctx
    r = _rng(nq * 4099 + nt)
    q, t = synth.random_desc(r, nq), synth.random_desc(r, nt)
    idx, dist = ctx.knn2(q, t)
    eidx, edist = oracle.knn2(q, t)
    assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
    for mutual in (False, True):
        m, n = ctx.match(q, t, 0.9, mutual)
        em, en = oracle.match(q, t, 0.9, mutual)
        assert np.array_equal(m, em) and n == en


@pytest.mark.parametrize("seed", range(6))
def test_tie_stress_bit_exact(ctx, oracle, seed):
    r = _rng(100 + seed)
    q, t = synth.tie_stress_desc(r, 300 + 17 * seed), synth.tie_stress_desc(r, 280 + 31 * seed)
    idx, dist = ctx.knn2(q, t)
    eidx, edist = oracle.knn2(q, t)
    assert np.array_equal(dist, edist)
    assert np.array_equal(idx, eidx)          # lowest trainIdx wins every tie
    for nnr in (0.6, 0.75, 0.9):
        m, n = ctx.match(q, t, nnr, True)
        em, en = oracle.match(q, t, nnr, True)
        assert np.array_equal(m, em) and n == en


@pytest.mark.parametrize("n1,n2", [(1, 2), (2, 1), (63, 65), (64, 64), (65, 63), (129, 1), (200, 200), (1500, 1500),
                                   (333, 1027), (1027, 333), (2, 2), (70, 3), (255, 17), (256, 16), (257, 15), (513, 31),
                                   (1, 1), (300, 5), (40, 1537), (1600, 3100), (17, 4000), (5000, 1999), (3000, 2048),
                                   (2048, 31), (257, 2047)])
def test_symmetric_and_directed_variants_agree(ctx, oracle, n1, n2):
    """Mutual problems default to the symmetric scan (one distance feeds both directions); forcing
    the directed lane-per-query scan must give the same tables, and both must equal the oracle."""
    import plslam_amd
    r = _rng(n1 * 7919 + n2)
    for gen in (synth.random_desc, synth.tie_stress_desc):
        d1, d2 = gen(r, n1), gen(r, n2)
        if gen is synth.random_desc and min(n1, n2) > 8:
            k = min(n1, n2) // 2
            d2[:k] = d1[:k] ^ np.packbits(r.random((k, 256)) < 0.06, axis=1)
        em, en = oracle.match(d1, d2, 0.9, True)
        try:
            # (variant, sym_rows, mfma_form): mfma_form 2 = K1f (group minima, the default), 1 = K1e (push per tile)
            for variant, sym_rows, form in ((plslam_amd.SCAN_MFMA, 0, 2), (plslam_amd.SCAN_MFMA, 0, 1), (plslam_amd.SCAN_MFMA, 0, 3),
                                            (plslam_amd.SCAN_MFMA, 0, 4), (plslam_amd.SCAN_MFMA, 0, 5), (plslam_amd.SCAN_MFMA, 0, 6),
                                            (plslam_amd.SCAN_SYMMETRIC, 4, 0), (plslam_amd.SCAN_SYMMETRIC, 1, 0),
                                            (plslam_amd.SCAN_LANE_PER_QUERY, 1, 0), (plslam_amd.SCAN_WAVE_PER_QUERY, 1, 0),
                                            (plslam_amd.SCAN_AUTO, 1, 0)):
                ctx.set_option("scan_variant", variant)
                ctx.set_option("sym_rows", sym_rows)
                # 4 = K1g (two directed scans per mutual problem), 5 = K1h, 6 = K1i; the earlier generations only where built
                if not set_mfma_form(ctx, {4: 3, 5: 4, 6: 5}.get(form, min(form, 2))):
                    continue
                ctx.set_option("fuse", 2 if form == 3 else 0)
                m, n = ctx.match(d1, d2, 0.9, True)
                assert np.array_equal(m, em) and n == en, (variant, sym_rows, form, gen.__name__)
        finally:
            ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)
            ctx.set_option("sym_rows", 0)
            ctx.set_option("mfma_form", 0)
            ctx.set_option("fuse", 0)


def test_all_scan_block_sizes(ctx, oracle):
    r = _rng(77)
    q = synth.random_desc(r, 1500)
    t, _, _ = synth.noisy_copy(r, q)
    em, en = oracle.match(q, t, 0.75, True)
    import plslam_amd
    try:
        ctx.set_option("scan_variant", plslam_amd.SCAN_LANE_PER_QUERY)
        for blk in (256, 512, 1024):
            ctx.set_option("scan_block", blk)
            m, n = ctx.match(q, t, 0.75, True)
            assert np.array_equal(m, em) and n == en, blk
    finally:
        ctx.set_option("scan_block", 0)
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)


@pytest.fixture(params=["auto", "lane_per_query", "wave_per_query", "symmetric", "mfma", "mfma_k1e", "mfma_fused", "mfma_k1g", "mfma_k1h", "mfma_k1f"])
def vctx(ctx, request):
    """The context with each scan variant forced in turn (AUTO picks wave-per-query for plans too
    small to fill the chip, the symmetric scan for mutual problems otherwise)."""
    import plslam_amd
    v = {"auto": plslam_amd.SCAN_AUTO, "lane_per_query": plslam_amd.SCAN_LANE_PER_QUERY,
         "wave_per_query": plslam_amd.SCAN_WAVE_PER_QUERY, "symmetric": plslam_amd.SCAN_SYMMETRIC,
         "mfma": plslam_amd.SCAN_MFMA, "mfma_k1e": plslam_amd.SCAN_MFMA, "mfma_fused": plslam_amd.SCAN_MFMA,
         "mfma_k1g": plslam_amd.SCAN_MFMA, "mfma_k1h": plslam_amd.SCAN_MFMA, "mfma_k1f": plslam_amd.SCAN_MFMA}[request.param]
    # "mfma" / "mfma_fused": form 0 = auto = K1i (the default scan; fused plans run K1f)
    if not set_mfma_form(ctx, {"mfma_k1e": 1, "mfma_k1g": 3, "mfma_k1h": 4, "mfma_k1f": 2}.get(request.param, 0)):
        pytest.skip("an earlier generation of the matrix-core scan: not in this build (PLSLAM_BUILD_LEGACY_SCANS=1)")
    ctx.set_option("scan_variant", v)
    ctx.set_option("fuse", 2 if request.param == "mfma_fused" else 0)      # one workgroup per problem incl. finalize
    yield ctx
    ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)
    ctx.set_option("mfma_form", 0)
    ctx.set_option("fuse", 0)


@pytest.fixture(params=[2, 1, 3, 4, 5, 6], ids=["k1f", "k1e", "k1f_fused", "k1g_directed", "k1h", "k1i"])
def mform(ctx, request):
    """The forms of the matrix-core scan: 2 = K1f (group minima + recomputed second best, the default), 1 = K1e (best-2
    push per tile), 3 = K1f with one workgroup per problem that also merges the columns and applies ratio + mutual
    (what AUTO picks for large plans; forced here so that small plans exercise it)."""
    if not set_mfma_form(ctx, {4: 3, 5: 4, 6: 5}.get(request.param, min(request.param, 2))):
        pytest.skip("an earlier generation of the matrix-core scan: not in this build (PLSLAM_BUILD_LEGACY_SCANS=1)")
    ctx.set_option("fuse", 2 if request.param == 3 else 1)
    yield request.param
    ctx.set_option("mfma_form", 0)
    ctx.set_option("fuse", 0)


def test_c2_full_size_pair_bit_exact(vctx, oracle):
    """BASELINE config 2: 1500 ORB + 200 LBD, L<->R and prev<->curr, mutual + ratio."""
    ctx = vctx
    s = synth.stereo_stream(2, 1500, 200, seed=synth.SEED0)
    for i in range(2):
        for name, d1, d2 in frontend.pair_problems(s["orb_l"], s["orb_r"], s["lbd_l"], s["lbd_r"], i):
            nnr = 0.75
            m, n = ctx.match(d1, d2, nnr, True)
            em, en = oracle.match(d1, d2, nnr, True)
            assert np.array_equal(m, em) and n == en, (i, name)
            if name.endswith("_lr") or i > 0:   # (pair 0's prev frame is the unrelated halo)
                assert n > 0.4 * len(m)         # the planted true matches are found


def test_c3_map_to_frame_sizes_bit_exact(vctx, oracle):
    """BASELINE config 3 matching half: 10k map points x 1500 frame rows, 2k lines x 200."""
    ctx = vctx
    r = _rng(31)
    frame_p = synth.random_desc(r, 1500)
    map_p = np.concatenate([synth.noisy_copy(r, frame_p)[0], synth.random_desc(r, 8500)])
    m, n = ctx.match(map_p, frame_p, 0.75, True)
    em, en = oracle.match(map_p, frame_p, 0.75, True)
    assert np.array_equal(m, em) and n == en and n > 500
    frame_l = synth.random_desc(r, 200)
    map_l = np.concatenate([synth.noisy_copy(r, frame_l)[0], synth.random_desc(r, 1800)])
    m, n = ctx.match(map_l, frame_l, 0.9, True)
    em, en = oracle.match(map_l, frame_l, 0.9, True)
    assert np.array_equal(m, em) and n == en


@pytest.mark.parametrize("shapes,mutual", [(((10000, 1500), (2000, 200)), True), (((3001, 2049), (700, 129)), True),
                                            (((517, 4400),), True), (((9000, 1500), (64, 33)), False)])
def test_column_split_of_few_large_problems(ctx, oracle, shapes, mutual):
    """C3 (one local map against one frame) in ONE plan: too few 256-row blocks to fill the chip, so the matrix-core
    scan cuts the columns into ranges (one workgroup per row block and range), merges the per-range row results
    (k_merge_row_splits) and the column partials of the ranges, then finalizes.  AUTO must pick it for the C3 shape; the
    tables, counts and every intermediate key word (plan dump: (best, second) incl. the second's index, both directions)
    must equal the unsplit scan's and the oracle's."""
    import torch
    import plslam_amd
    r = _rng(4711 + shapes[0][0])
    dev = torch.device("cuda", ctx.device)
    host, probs, outs = [], [], []
    cnt = torch.zeros(len(shapes), dtype=torch.int32, device=dev)
    for k, (n1, n2) in enumerate(shapes):
        b = synth.tie_stress_desc(r, n2) if k == 1 else synth.random_desc(r, n2)
        a = np.concatenate([synth.noisy_copy(r, b)[0], synth.random_desc(r, n1)])[:n1]
        ta, tb = torch.from_numpy(np.ascontiguousarray(a)).to(dev), torch.from_numpy(b).to(dev)
        m = torch.empty(n1, dtype=torch.int32, device=dev)
        host.append((a, b, ta, tb))
        outs.append(m)
        probs.append((ta.data_ptr(), n1, tb.data_ptr(), n2, 0.8, mutual, m.data_ptr(), cnt.data_ptr() + 4 * k))
    got = {}
    try:
        for tag, variant, split in (("auto", plslam_amd.SCAN_AUTO, 0), ("split", plslam_amd.SCAN_MFMA, 2),
                                    ("unsplit", plslam_amd.SCAN_MFMA, 1), ("graph", plslam_amd.SCAN_AUTO, 0)):
            ctx.set_option("scan_variant", variant)
            ctx.set_option("col_split", split)
            ctx.set_option("exact_second", 1)          # the key tables are compared word for word below
            ctx.set_option("graph", 2 if tag == "graph" else 1)     # the run captured once, replayed as a HIP graph
            plan = ctx.plan(probs)
            info = plan.info()
            for m in outs:
                m.fill_(-7)
            for _ in range(3 if tag == "graph" else 1):            # (capture + two replays)
                plan.run(0)
            torch.cuda.synchronize()
            keys, _ = plan.dump()
            got[tag] = (keys.copy(), [m.cpu().numpy().copy() for m in outs], cnt.cpu().numpy().copy(), info)
            plan.close()
    finally:
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)
        ctx.set_option("col_split", 0)
        ctx.set_option("exact_second", 0)
        ctx.set_option("graph", 1)
    big = sum(n1 * n2 for n1, n2 in shapes) >= 6 << 20
    assert got["auto"][3]["scan_variant"] == (plslam_amd.SCAN_MFMA if big else plslam_amd.SCAN_WAVE_PER_QUERY)
    assert got["split"][3]["scan_blocks"] > got["unsplit"][3]["scan_blocks"]          # more workgroups, same work
    nkeys = 2 * sum(n1 + (n2 if mutual else 0) for n1, n2 in shapes)
    assert np.array_equal(got["split"][0][:nkeys], got["unsplit"][0][:nkeys])
    for k, (a, b, _, _) in enumerate(host):
        em, en = oracle.match(a, b, 0.8, mutual)
        for tag in ("auto", "split", "unsplit", "graph"):
            assert np.array_equal(got[tag][1][k], em), (tag, k)
            assert got[tag][2][k] == en, (tag, k)


def test_host_to_host_pipeline(ctx, oracle):
    """plslam_match_pipeline: batches from pinned host memory, three in flight (upload / kernels / download on their own
    streams), problems that SHARE rows inside the arena (prev<->curr reuses the left image of the pair before).  Every
    table of every batch -- different data per batch -- equals the oracle's; slots are re-used several times."""
    B, n_orb, n_lbd = 6, 300, 70
    s = synth.stereo_stream(5 * B, n_orb, n_lbd, seed=606)
    hp = frontend.HostStereoPipeline(ctx, B, n_orb, n_lbd, nnr_p=0.8, nnr_l=0.9, mutual=True, depth=3)
    sl = frontend.table_slices(n_orb, n_lbd)
    got = {}
    for k in range(5):                                   # host buffers k % 4; device slot k % 3.  Host buffer k % 4 is free:
        slot = k % (hp.depth + 1)                        # the submit of batch k - 1 waited for batch k - 4's device slot
        hp.fill(slot, s, first=k * B)
        hp.submit(slot)
    hp.wait()
    for k in range(1, 5):                                # the last four batches are still in the host tables
        got[k] = hp.tables[k % (hp.depth + 1)].array.copy()
    for k, tab in got.items():
        for i in range(B):
            for name, d1, d2 in frontend.pair_problems(s["orb_l"], s["orb_r"], s["lbd_l"], s["lbd_r"], k * B + i):
                em, _ = oracle.match(d1, d2, 0.8 if name.startswith("orb") else 0.9, True)
                assert np.array_equal(tab[i, sl[name]], em), (k, i, name)
    hp.close()


def test_c5_dense_size_properties(ctx, oracle):
    """BASELINE config 5 size (4000 ORB): size-independent properties + oracle spot check."""
    r = _rng(55)
    a = synth.random_desc(r, 4000)
    # identity: distinct rows matched against themselves -> every row matches itself (d0=0 < d1*nnr)
    m, n = ctx.match(a, a, 0.75, True)
    assert n == 4000 and np.array_equal(m, np.arange(4000))
    # permutation equivariance + involution of mutual matches
    b, perm, fresh = synth.noisy_copy(r, a)
    m12, n12 = ctx.match(a, b, 0.75, True)
    m21, n21 = ctx.match(b, a, 0.75, True)
    assert n12 == n21
    ok = m12 >= 0
    assert np.array_equal(m21[m12[ok]], np.nonzero(ok)[0])       # mutual => involutive
    inv = np.empty(4000, np.int64)
    inv[perm] = np.arange(4000)
    planted = ok & ~fresh[np.clip(m12, 0, None)]
    assert (m12[planted] == inv[planted]).mean() > 0.999         # planted matches recovered
    # checksum of the table equals the oracle's (full bit-exact compare is cheap enough here too)
    em, en = oracle.match(a, b, 0.75, True)
    assert np.array_equal(m12, em) and n12 == en


def test_batched_ragged_problems(vctx, oracle):
    ctx = vctx
    r = _rng(9)
    sizes1 = [30, 0, 17, 64, 300, 1, 2, 513]
    sizes2 = [25, 10, 0, 70, 299, 5, 1, 255]
    off1 = np.concatenate([[0], np.cumsum(sizes1)]).astype(np.int32)
    off2 = np.concatenate([[0], np.cumsum(sizes2)]).astype(np.int32)
    d1, d2 = synth.random_desc(r, off1[-1]), synth.random_desc(r, off2[-1])
    for mutual in (False, True):
        m, nm = ctx.match_batched(d1, off1, d2, off2, 0.9, mutual)
        em, enm = oracle.match_batched(d1, off1, d2, off2, 0.9, mutual)
        assert np.array_equal(m, em) and np.array_equal(nm, enm)


def test_error_codes(ctx):
    import plslam_amd
    with pytest.raises(plslam_amd.PlslamError) as e:
        ctx.set_option("scan_block", 100)
    assert e.value.code == plslam_amd.capi.EINVAL
    with pytest.raises(plslam_amd.PlslamError) as e:
        ctx.set_option("no_such_option", 1)
    assert e.value.code == plslam_amd.capi.EINVAL
    off = np.array([0, 5, 3], np.int32)      # decreasing offsets
    with pytest.raises(plslam_amd.PlslamError):
        ctx.match_batched(np.zeros((5, 32), np.uint8), off, np.zeros((5, 32), np.uint8), off, 0.9, True)


def test_device_resident_plan_matches_oracle(vctx, oracle):
    """The throughput path: StereoBatchMatcher (device-resident, one plan, torch's stream)."""
    import torch
    import plslam_amd
    ctx = vctx
    s = synth.stereo_stream(3, 320, 70, seed=5)
    bm = frontend.StereoBatchMatcher(ctx, s, nnr_p=0.75, nnr_l=0.9, mutual=True)
    info = bm.plan.info()
    assert info["n_scans"] == 3 * 8 and info["directed_evals"] == 3 * 4 * (320 * 320 + 70 * 70)
    if info["scan_variant"] in (plslam_amd.SCAN_SYMMETRIC, plslam_amd.SCAN_MFMA) and ctx.get_option("mfma_form") != 3:
        assert info["distance_evals"] == info["directed_evals"] // 2   # half the distances (K1g executes both directions)
    else:
        assert info["distance_evals"] == info["directed_evals"]
    for _ in range(2):                        # re-running a plan is idempotent
        tab = bm.run()
        torch.cuda.synchronize()
        tab = tab.cpu().numpy()
        cnt = bm.counts.cpu().numpy()
        sl = frontend.table_slices(320, 70)
        for i in range(3):
            for k, (name, d1, d2) in enumerate(frontend.pair_problems(s["orb_l"], s["orb_r"], s["lbd_l"],
                                                                      s["lbd_r"], i)):
                em, en = oracle.match(d1, d2, 0.75 if name.startswith("orb") else 0.9, True)
                assert np.array_equal(tab[i, sl[name]], em), (i, name)
                assert cnt[i, k] == en
    bm.plan.set_profiling(True)
    bm.run()
    scan_ms, fin_ms, runs = bm.plan.elapsed()
    assert runs == 1 and scan_ms > 0 and fin_ms > 0
    bm.close()


@pytest.mark.parametrize("n_orb,n_lbd,expect", [(2048, 33, "mfma"), (2049, 33, "mfma"), (1999, 1, "mfma"),
                                                  (96, 2048, "mfma"), (31, 32, "mfma"), (4130, 65, "mfma")])
def test_matrix_core_scan_limits(ctx, oracle, mform, n_orb, n_lbd, expect):
    """K1e keeps 16-bit (distance, tile) row keys, so it scans in windows of 64 tiles (2048 columns) and merges
    the row results of successive windows.  At n2 = 2048 every tile number of one window is used; 2049 starts a
    second window with a single column; 4130 needs three.  Tie-stress data (many equal distances, so that
    equal keys meet across window borders), all four problems of every pair against the oracle."""
    import torch
    import plslam_amd
    s = synth.stereo_stream(2, n_orb, n_lbd, seed=77, tie_stress=True)
    ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)      # (AUTO would pick wave-per-query for 2 pairs)
    try:
        bm = frontend.StereoBatchMatcher(ctx, s, nnr_p=0.8, nnr_l=0.9, mutual=True)
    finally:
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)
    used = bm.plan.info()["scan_variant"]
    assert used == (plslam_amd.SCAN_MFMA if expect == "mfma" else plslam_amd.SCAN_SYMMETRIC)
    tab = bm.run()
    torch.cuda.synchronize()
    tab = tab.cpu().numpy()
    sl = frontend.table_slices(n_orb, n_lbd)
    for i in range(2):
        for name, d1, d2 in frontend.pair_problems(s["orb_l"], s["orb_r"], s["lbd_l"], s["lbd_r"], i):
            em, _ = oracle.match(d1, d2, 0.8 if name.startswith("orb") else 0.9, True)
            assert np.array_equal(tab[i, sl[name]], em), (i, name)
    bm.close()


def test_matrix_core_scan_extreme_distances(ctx, oracle, mform):
    """Distances 0 and 256 (exact complements) and rows of all zeros / all ones: the +-1 byte contraction
    must give exactly 2 d in [0, 512] and the keys must not wrap."""
    import plslam_amd
    r = _rng(5)
    a = synth.random_desc(r, 300)
    b = a.copy()[::-1].copy()
    b[:40] = ~a[:40]                        # complements: distance 256 to their source row
    a[7] = 0
    a[8] = 255
    b[100] = 0
    b[101] = 255
    try:
        ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
        for nnr in (0.6, 1.0, 1.5):
            m, n = ctx.match(a, b, nnr, True)
            em, en = oracle.match(a, b, nnr, True)
            assert np.array_equal(m, em) and n == en
        ctx.set_option("scan_variant", plslam_amd.SCAN_LANE_PER_QUERY)
        assert np.array_equal(ctx.match(a, b, 1.0, True)[0], oracle.match(a, b, 1.0, True)[0])
    finally:
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)


def test_directed_matrix_core_paths(ctx, oracle, mform):
    """Non-mutual problems and plain knnMatch(k=2) on the directed form of K1e: a large query set under AUTO
    (66 000 queries -> past the latency-kernel threshold), a multi-window train set, ties, and a batched
    non-mutual plan next to mutual problems."""
    import plslam_amd
    r = _rng(31)
    q = synth.tie_stress_desc(r, 66000)
    t = synth.tie_stress_desc(r, 300)
    idx, dist = ctx.knn2(q, t)                               # AUTO
    eidx, edist = oracle.knn2(q, t)
    assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
    try:
        ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
        q2, t2 = synth.random_desc(r, 700), synth.tie_stress_desc(r, 5000)      # 157 tiles: 3 windows
        idx, dist = ctx.knn2(q2, t2)
        eidx, edist = oracle.knn2(q2, t2)
        assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
        for n1, n2 in ((1, 1), (1, 2), (300, 1), (257, 33), (1000, 2049)):
            a, b = synth.random_desc(r, n1), synth.tie_stress_desc(r, n2)
            m, n = ctx.match(a, b, 0.9, False)
            em, en = oracle.match(a, b, 0.9, False)
            assert np.array_equal(m, em) and n == en, (n1, n2)
        # mixed batch: problems 0, 2 mutual (symmetric K1e), 1, 3 would be separate calls in the reference;
        # match_batched applies one mutual flag per call, so run both flags over the same ragged batch
        sizes1, sizes2 = [300, 0, 513, 64], [280, 7, 100, 2500]
        d1 = synth.random_desc(r, sum(sizes1)); d2 = synth.random_desc(r, sum(sizes2))
        off1 = np.concatenate([[0], np.cumsum(sizes1)]).astype(np.int32)
        off2 = np.concatenate([[0], np.cumsum(sizes2)]).astype(np.int32)
        for mutual in (False, True):
            m, cnt = ctx.match_batched(d1, off1, d2, off2, 0.8, mutual)
            for b_ in range(4):
                em, en = oracle.match(d1[off1[b_]:off1[b_ + 1]], d2[off2[b_]:off2[b_ + 1]], 0.8, mutual)
                assert np.array_equal(m[off1[b_]:off1[b_ + 1]], em) and cnt[b_] == en, (mutual, b_)
    finally:
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)


def test_large_train_set_index_bits(vctx, oracle):
    """Train indices beyond 16 bits (the composite key keeps 23 index bits): 70 000 train rows,
    planted best/second-best at the far end, plus exact duplicates to force index tie-breaks."""
    ctx = vctx
    r = _rng(123)
    q = synth.random_desc(r, 130)
    t = synth.random_desc(r, 70000)
    t[69990] = q[0]                       # exact hit at a > 16-bit index
    t[69999] = q[0] ^ np.packbits(np.arange(256) < 3)
    t[66000] = q[1]
    t[66001] = q[1]                       # duplicate: lower index must win
    idx, dist = ctx.knn2(q, t)
    eidx, edist = oracle.knn2(q, t)
    assert np.array_equal(idx, eidx) and np.array_equal(dist, edist)
    assert list(idx[0]) == [69990, 69999] and list(idx[1]) == [66000, 66001]
    m, n = ctx.match(q, t, 0.75, True)
    em, en = oracle.match(q, t, 0.75, True)
    assert np.array_equal(m, em) and n == en


def test_fuzz_all_variants_vs_oracle(ctx, oracle):
    """Randomised sizes / ties / thresholds / variants against the oracle (fixed seeds)."""
    import plslam_amd
    r = _rng(2024)
    variants = (plslam_amd.SCAN_AUTO, plslam_amd.SCAN_LANE_PER_QUERY, plslam_amd.SCAN_WAVE_PER_QUERY,
                plslam_amd.SCAN_SYMMETRIC, plslam_amd.SCAN_MFMA)
    try:
        for case in range(160):
            n1, n2 = int(r.integers(0, 420)), int(r.integers(0, 420))
            gen = synth.tie_stress_desc if case % 3 == 0 else synth.random_desc
            d1, d2 = gen(r, n1), gen(r, n2)
            if gen is synth.random_desc and min(n1, n2) > 4:
                k = int(r.integers(1, min(n1, n2)))
                d2[:k] = d1[:k] ^ np.packbits(r.random((k, 256)) < 0.07, axis=1)
            nnr = float(r.choice([0.6, 0.75, 0.8, 0.9]))
            mutual = bool(case % 2)
            ctx.set_option("scan_variant", variants[case % 4])
            ctx.set_option("sym_rows", [0, 1, 4][case % 3])
            m, n = ctx.match(d1, d2, nnr, mutual)
            em, en = oracle.match(d1, d2, nnr, mutual)
            assert np.array_equal(m, em) and n == en, (case, n1, n2, nnr, mutual)
    finally:
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)
        ctx.set_option("sym_rows", 0)


@pytest.mark.parametrize("n_orb,n_lbd,pairs,mutual", [(256, 256, 128, True), (800, 100, 64, True), (256, 256, 128, False)])
def test_batched_scan_is_deterministic_under_load(ctx, mform, n_orb, n_lbd, pairs, mutual):
    """Two overlapped plans (several workgroups per CU), repeated: every intermediate key word, column partial
    and table entry must repeat exactly.  Regression for the K1e symmetric scan's exec-masked column store, which
    made ~1 % of the key words differ from run to run while every single-problem parity test
    stayed green."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "determinism_check", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools", "determinism_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.check(ctx, n_orb, n_lbd, pairs, rounds=8, nnr=0.9, mutual=mutual)
    assert r["key_words"] > 0
    assert (r["key_diffs"], r["partial_diffs"], r["table_diffs"]) == (0, 0, 0), r


@pytest.mark.parametrize("n_orb,n_lbd,pairs,mutual", [(1500, 200, 24, True), (300, 70, 96, True), (2100, 33, 8, True),
                                                        (257, 4130, 4, True), (777, 130, 32, False), (40, 2500, 16, False),
                                                        (2060, 513, 3, True)])   # 65 tiles: a second window of ONE tile; 17 tiles
def test_both_matrix_core_forms_produce_identical_keys(ctx, n_orb, n_lbd, pairs, mutual):
    """K1f (group minima; the second best of the winner's group recomputed from the raw rows) against K1e (every key
    pushed): not only the match tables but every intermediate word -- keys12 = (best, second best) per row with the
    second best's INDEX, and keys21, the merged column results -- must be identical, on tie-heavy data, ragged sizes,
    several windows (n2 > 2048) and the directed form."""
    import torch
    import plslam_amd
    s = synth.stereo_stream(pairs, n_orb, n_lbd, seed=4242 + n_orb, tie_stress=(n_orb % 2 == 1))
    got = {}
    try:
        ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
        # K1e, K1f, K1f fused (one workgroup per problem incl. merge + finalize), K1g (two directed scans per mutual
        # problem), K1h with exact key tables ("exact_second": by default K1h leaves the second neighbour's INDEX and the
        # columns' second key to the stages that need them -- the match tables below are compared in that default too)
        # 7 / 8: K1i (the default scan) with exact key tables / in its default
        for form in (1, 2, 3, 4, 5, 6, 7, 8):
            if not set_mfma_form(ctx, {4: 3, 5: 4, 6: 4, 7: 5, 8: 5}.get(form, min(form, 2))):
                continue                           # (K1e / K1g / K1h: only in a build with PLSLAM_BUILD_LEGACY_SCANS=1)
            ctx.set_option("exact_second", 1 if form in (4, 5, 7) else 0)
            ctx.set_option("fuse", 2 if form == 3 else 1)
            ctx.set_option("post_fuse", 1)         # the merged column keys are compared below: they exist in memory only with the separate kernels
            bm = frontend.StereoBatchMatcher(ctx, s, nnr_p=0.85, nnr_l=0.9, mutual=mutual)
            tab = bm.run()
            torch.cuda.synchronize()
            keys, part = bm.plan.dump()
            got[form] = (keys.copy(), part.copy(), tab.cpu().numpy().copy(), bm.counts.cpu().numpy().copy())
            bm.close()
    finally:
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)
        ctx.set_option("mfma_form", 0)
        ctx.set_option("fuse", 0)
        ctx.set_option("exact_second", 0)
        ctx.set_option("post_fuse", 0)
    # the dump returns the buffers with their 25 % growth slack: compare the words the kernels write.  (The column
    # PARTIALS are laid out differently -- K1e: 32-bit keys per 256-row block, K1f: 16-bit keys per 64-row block -- so
    # the column direction is compared after the merge: keys21 is part of the key table.)
    rows = pairs * 2 * ((n_orb + n_lbd) * (2 if mutual else 1))
    base = 1 if 1 in got else 2                    # K1e where it is built, else K1f: both push / recompute EXACT keys
    assert {2, 3, 7, 8} <= set(got) and got[base][0].size >= 2 * rows
    k1 = got[base][0][:2 * rows]
    for form in sorted(set(got) - {base}):
        k2 = got[form][0][:2 * rows]
        if form not in (6, 8):
            assert np.array_equal(k1, k2), (form, int((k1 != k2).sum()))
        else:
            # default K1h: every row's best key is exact, and so is its second-best DISTANCE; of the merged column keys the
            # best one (the second one is completed lazily by the finalize kernel: ProblemDesc::lazy21)
            a, b = k1.reshape(-1, 2), k2.reshape(-1, 2)
            assert np.array_equal(a[:, 0], b[:, 0]), (form, int((a[:, 0] != b[:, 0]).sum()))
            is12 = np.zeros(rows, dtype=bool)                # per problem: n1 rows of keys12, then (mutual) n2 rows of keys21
            at = 0
            for _ in range(pairs):
                for n in (n_orb, n_orb, n_lbd, n_lbd):
                    is12[at:at + n] = True
                    at += n * (2 if mutual else 1)
            assert at == rows
            assert np.array_equal(a[is12, 0], b[is12, 0]), (form, int((a[is12, 0] != b[is12, 0]).sum()))
            assert np.array_equal(a[is12, 1] >> 23, b[is12, 1] >> 23), form
        assert np.array_equal(got[base][2], got[form][2]), form
        assert np.array_equal(got[base][3], got[form][3]), form


def test_batched_tables_equal_oracle_at_loose_ratio(ctx, oracle):
    """The overlapped batch at nnr 0.9 (where a one-bit distance error flips the most ratio tests) against the
    oracle, pair by pair, after several back-to-back runs on two streams."""
    import torch
    n_orb, n_lbd, pairs = 300, 120, 48
    stream = synth.stereo_stream(pairs, n_orb, n_lbd, seed=synth.SEED0 + 5, first_pair=0)
    bm = frontend.StereoBatchMatcher(ctx, stream, nnr_p=0.9, nnr_l=0.9, mutual=True, device=torch.device("cuda", 0),
                                     n_buffers=2)
    for k in range(6):
        bm.run_overlapped(k)
    bm.synchronize_all()
    sl = frontend.table_slices(n_orb, n_lbd)
    exp = np.full((pairs, bm.stride), -2, np.int32)
    for i in range(pairs):
        for name, d1, d2 in frontend.pair_problems(stream["orb_l"], stream["orb_r"], stream["lbd_l"], stream["lbd_r"], i):
            exp[i, sl[name]] = oracle.match(d1, d2, 0.9, True)[0]
    for t in bm.tables:
        got = t.cpu().numpy()
        for name in sl:
            assert np.array_equal(got[:, sl[name]], exp[:, sl[name]]), name
    bm.close()


def test_one_context_shared_by_three_threads(ctx, oracle):
    """The reference calls match() from the VO thread, the local-mapping thread and the loop-closure thread
    (src/mapHandler.cpp:1103, :1164 -> :3223; app/plslam_dataset.cpp:127) with no shared matcher state.
    Here all three share ONE context: the host-pointer entry points serialise on its mutex (ctypes drops the GIL
    for the call), every result must still be the oracle's."""
    import threading
    from test_match_grid_cpu import point_case
    r = _rng(4242)
    jobs = []
    for k in range(6):
        q, t = synth.random_desc(r, 200 + 61 * k), synth.random_desc(r, 180 + 47 * k)
        jobs.append((q, t, oracle.match(q, t, 0.9, True), oracle.knn2(q, t)))
    grid_case = point_case(11, 600, 640, 64, 48)
    grid_ref = oracle.match_grid(window=(3, 3, 3, 3), nnr=0.8, mutual=True, **grid_case)
    errors = []

    def matcher(tid):
        try:
            for rep in range(5):
                for q, t, (em, en), _ in jobs[tid::2]:
                    m, n = ctx.match(q, t, 0.9, True)
                    assert np.array_equal(m, em) and n == en
        except Exception as e:                                        # noqa: BLE001 -- reported below
            errors.append(("match", tid, repr(e)))

    def scanner():
        try:
            for rep in range(5):
                for q, t, _, (ei, ed) in jobs:
                    i, d = ctx.knn2(q, t)
                    assert np.array_equal(i, ei) and np.array_equal(d, ed)
        except Exception as e:                                        # noqa: BLE001
            errors.append(("knn2", repr(e)))

    def windowed():
        try:
            for rep in range(10):
                m, n = ctx.match_grid(window=(3, 3, 3, 3), nnr=0.8, mutual=True, **grid_case)
                assert np.array_equal(m, grid_ref[0]) and n == grid_ref[1]
        except Exception as e:                                        # noqa: BLE001
            errors.append(("grid", repr(e)))

    th = [threading.Thread(target=matcher, args=(0,)), threading.Thread(target=matcher, args=(1,)),
          threading.Thread(target=scanner), threading.Thread(target=windowed)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors


def test_knn2_against_the_reference_kNN_outputs(ctx):
    """tests/golden/ref_knn_golden.npz holds outputs of the reference's own exact kNN search
    (BinaryDescriptorMatcher::knnMatch, binary_descriptor_matcher.cpp:258-335; generated by
    tests/golden/make_ref_knn_golden.py): the device's two nearest distances must equal them for every query, the
    indices wherever distance alone decides."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_knn_golden.npz"))
    for name in g["names"]:
        q, t = g[f"{name}/q"], g[f"{name}/t"]
        idx, dist = ctx.knn2(q, t)
        assert np.array_equal(dist, g[f"{name}/k2_dist"]), name
        u = np.all(g[f"{name}/k2_idx"] >= 0, axis=1)      # queries whose three nearest distances are distinct
        assert u.sum() > 0 or name == "ties"
        assert np.array_equal(idx[u], g[f"{name}/k2_idx"][u]), name


@pytest.mark.parametrize("mutual", [False, True])
@pytest.mark.parametrize("n1,n2", [(1500, 1400), (200, 37), (3000, 2100), (5, 1)])
def test_match_on_a_vector_that_already_holds_entries(vctx, oracle, n1, n2, mutual):
    """plslam_match_prior / plslam_match_problem.keep_prior: the fall-back of src/mapHandler.cpp:274-278 (and :421-425,
    :594-598, :709-713) runs StVO::match on the vector matchGrid filled -- rejected rows keep their entry, the
    consistency loop covers them, the count follows the reference's arithmetic.  Every scan form, vs the oracle."""
    r = np.random.Generator(np.random.PCG64(n1 * 7 + n2))
    d1 = synth.random_desc(r, n1)
    d2 = synth.random_desc(r, n2)
    k = min(n1, n2) // 2
    d2[:k] = d1[:k] ^ np.packbits(r.random((k, 256)) < 0.04, axis=1)
    prior = np.where(r.random(n1) < 0.5, r.integers(0, n2, n1), -1).astype(np.int32)
    got, n = vctx.match_prior(d1, d2, 0.75, mutual, prior)
    ref, nref = oracle.match_prior(d1, d2, 0.75, mutual, prior)
    np.testing.assert_array_equal(got, ref)
    assert n == nref
    got0, n0 = vctx.match_prior(d1, d2, 0.75, mutual, np.full(n1, -1, np.int32))
    ref0, nref0 = oracle.match(d1, d2, 0.75, mutual)
    np.testing.assert_array_equal(got0, ref0)
    assert n0 == nref0


def test_entry_points_run_on_the_context_device_not_the_current_one():
    """Needs two GPUs (skipped on the 1-GPU test boxes): a context on device 1 driven from a thread whose current device
    is 0 -- the map<->keyframe drivers, the LBA plan and the matcher allocate and launch on device 1."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible GPU")
    from oracle import oracle as O
    torch.cuda.set_device(0)
    ctx = plslam_amd.Context(1)
    r = np.random.Generator(np.random.PCG64(5))
    d1, d2 = synth.random_desc(r, 700), synth.random_desc(r, 650)
    d2[:300] = d1[:300] ^ np.packbits(r.random((300, 256)) < 0.04, axis=1)
    got = ctx.match(d1, d2, 0.75, True)
    ref = O.match(d1, d2, 0.75, True)
    np.testing.assert_array_equal(got[0], ref[0])
    import test_map2kf as T
    cam, ocam = plslam_amd.make_cam(**synth.EUROC), O.make_cam(**synth.EUROC)
    s = T.scene(1200, 500)
    a = (s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"])
    g = ctx.map2kf_match_fast("points", cam, *a, 0.9, True, 1.5, 10, T.fast_cfg())
    e = O.map2kf_match_fast("points", ocam, *a, 0.9, True, 1.5, 10, T.fast_cfg())
    np.testing.assert_array_equal(g[0], e[0])
    assert torch.cuda.current_device() == 0


def test_split_runs_order_themselves(ctx, oracle):
    """plslam_match_plan_run_split: scan on one stream, merge / finalize / gates on another.  The same plan run again
    (split, then plain, on a third stream) waits for the stages of its previous run; the tables equal the oracle's."""
    import torch
    dev = torch.device("cuda", 0)
    n_orb, n_lbd, pairs = 700, 150, 96
    stream = synth.stereo_stream(pairs, n_orb, n_lbd, seed=synth.SEED0 + 9, first_pair=0)
    bm = frontend.StereoBatchMatcher(ctx, stream, nnr_p=0.8, nnr_l=0.8, mutual=True, device=dev, n_buffers=1)
    s = [torch.cuda.Stream(device=dev) for _ in range(3)]
    plan = bm.plans[0]
    for rep in range(4):
        plan.run_split(s[0].cuda_stream, s[1].cuda_stream)
        plan.run_split(s[1].cuda_stream, s[0].cuda_stream)        # roles swapped: still ordered by the plan's events
        plan.run(s[2].cuda_stream)
    for x in s:
        x.synchronize()
    sl = frontend.table_slices(n_orb, n_lbd)
    got = bm.tables[0].cpu().numpy()
    for i in range(0, pairs, 7):
        for name, d1, d2 in frontend.pair_problems(stream["orb_l"], stream["orb_r"], stream["lbd_l"], stream["lbd_r"], i):
            assert np.array_equal(got[i, sl[name]], oracle.match(d1, d2, 0.8, True)[0]), (i, name)
    bm.close()


@pytest.mark.parametrize("n_orb,n_lbd,pairs", [(1500, 200, 16), (300, 70, 64), (257, 4000, 6), (2100, 33, 6), (40, 17, 200), (513, 1, 24)])
def test_fused_stage_behind_the_scan_equals_the_separate_kernels(ctx, oracle, n_orb, n_lbd, pairs):
    """k_post_fused (one workgroup per problem: column merge into LDS, finalize, gates) against k_merge_fix16 + k_finalize on
    the same plans -- tables, counts, gated associations, disparities -- and against the oracle; tie-heavy data, ragged sizes,
    several row blocks and column windows, keep_prior-free batch plans.  Forced ("post_fuse" 2: AUTO wants two problems per
    CU) and refused (1)."""
    import torch
    import plslam_amd
    from plslam_amd import frontend, synth
    s = synth.stereo_stream(pairs, n_orb, n_lbd, seed=77 + n_orb, tie_stress=(n_orb % 2 == 1))
    geo = synth.stereo_geometry(s, seed=5)
    out = {}
    try:
        ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
        for pf in (1, 2):
            ctx.set_option("post_fuse", pf)
            bm = frontend.StereoBatchMatcher(ctx, s, nnr_p=0.8, nnr_l=0.9, mutual=True, geometry=geo, gates=dict(synth.KITTI_GATES), n_buffers=2)
            for k in range(3):
                bm.run_overlapped(k)
            bm.synchronize_all()
            out[pf] = [x.cpu().numpy().copy() for x in (bm.tables[0], bm.count_bufs[0], bm.stereo_tabs[0], bm.stereo_disps[0].view(torch.int64),
                                                       bm.stereo_cnts[0], bm.tables[1], bm.stereo_tabs[1])]
            bm.close()
    finally:
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)
        ctx.set_option("post_fuse", 0)
    for a, b in zip(out[1], out[2]):
        assert np.array_equal(a, b)
    sl = frontend.table_slices(n_orb, n_lbd)
    for i in range(0, pairs, max(1, pairs // 8)):
        for name, d1, d2 in frontend.pair_problems(s["orb_l"], s["orb_r"], s["lbd_l"], s["lbd_r"], i):
            em, _ = oracle.match(d1, d2, 0.8 if name.startswith("orb") else 0.9, True)
            assert np.array_equal(out[2][0][i, sl[name]], em), (i, name)


@pytest.mark.parametrize("form", [4, 5], ids=["k1h", "k1i"])
def test_lazy_second_keys_with_prior_entries_column_split_and_a_capped_post_grid(ctx, oracle, form):
    """The minimum-only scans complete a column's second key lazily in the finalize kernel, from the neighbouring rows' keys
    (DPP over 16-row groups: finalize blocks start at multiples of 256 rows and every lane runs the sequence).  The option
    combinations the default tests do not reach: a vector that already holds entries (keep_prior), a forced column split,
    a capped grid behind the scan (post_workgroups) -- tables and counts against the oracle, exact_second at its default 0."""
    import plslam_amd
    r = np.random.Generator(np.random.PCG64(900 + form))
    try:
        ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
        if not set_mfma_form(ctx, form):
            pytest.skip("K1h's scan kernel: not in this build (PLSLAM_BUILD_LEGACY_SCANS=1)")
        for n1, n2 in ((1500, 1400), (3000, 2100), (700, 2600)):
            d1, d2 = synth.random_desc(r, n1), synth.random_desc(r, n2)
            k = min(n1, n2) // 2
            d2[:k] = d1[:k] ^ np.packbits(r.random((k, 256)) < 0.04, axis=1)
            prior = np.where(r.random(n1) < 0.5, r.integers(0, n2, n1), -1).astype(np.int32)
            ref, nref = oracle.match_prior(d1, d2, 0.75, True, prior)
            ref0, nref0 = oracle.match(d1, d2, 0.75, True)
            for split, post in ((0, 0), (2, 0), (0, 3), (2, 5)):
                ctx.set_option("col_split", split)
                ctx.set_option("post_workgroups", post)
                got, n = ctx.match_prior(d1, d2, 0.75, True, prior)
                np.testing.assert_array_equal(got, ref)
                assert n == nref
                got0, n0 = ctx.match(d1, d2, 0.75, True)
                np.testing.assert_array_equal(got0, ref0)
                assert n0 == nref0
    finally:
        for k in ("col_split", "post_workgroups", "mfma_form"):
            ctx.set_option(k, 0)
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)


@pytest.mark.parametrize("shapes", [((10000, 1500), (2000, 200)), ((3001, 2049), (700, 129)), ((9000, 1500), (64, 33)),
                                    ((517, 4400),), ((6000, 300), (257, 2)), ((4100, 130), (5000, 1))],
                         ids=["c3", "multi_window", "with_a_one_range_problem", "wide", "two_columns", "one_column"])
def test_column_split_plan_in_two_launches(ctx, oracle, shapes):
    """A column-split plan of mutual problems runs as TWO launches: the scan, then k_split_post, which merges the column
    partials and decides the matches from the column side (rows without a match keep the -1 the scan's first column range
    writes).  Tables (pre-filled with garbage), counts and every key word of the dump must equal the three-launch form
    (merge kernel + finalize kernel, option split_post = 1) and the oracle; repeated runs must not accumulate counts."""
    import torch
    import plslam_amd
    r = _rng(9100 + shapes[0][0])
    dev = torch.device("cuda", ctx.device)
    host, probs, outs = [], [], []
    cnt = torch.zeros(len(shapes), dtype=torch.int32, device=dev)
    for k, (n1, n2) in enumerate(shapes):
        b = synth.tie_stress_desc(r, n2) if (k == 1 and n2 > 8) else synth.random_desc(r, n2)
        a = np.concatenate([synth.noisy_copy(r, b)[0], synth.random_desc(r, n1)])[:n1]
        a = np.ascontiguousarray(a[r.permutation(n1)])       # the matching rows anywhere in the map, not in its first blocks
        ta, tb = torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)
        m = torch.empty(n1, dtype=torch.int32, device=dev)
        host.append((a, b, ta, tb))
        outs.append(m)
        probs.append((ta.data_ptr(), n1, tb.data_ptr(), n2, 0.8, True, m.data_ptr(), cnt.data_ptr() + 4 * k))
    got = {}
    try:
        ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
        ctx.set_option("col_split", 2)
        for sp in (0, 1):
            ctx.set_option("split_post", sp)
            plan = ctx.plan(probs)
            for rep in range(3):
                for m in outs:
                    m.fill_(-7 - rep)
                plan.run(0)
            torch.cuda.synchronize()
            keys, _ = plan.dump()
            got[sp] = (keys.copy(), [m.cpu().numpy().copy() for m in outs], cnt.cpu().numpy().copy(), plan.info())
            plan.close()
    finally:
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)
        ctx.set_option("col_split", 0)
        ctx.set_option("split_post", 0)
    nkeys = 2 * sum(n1 + n2 for n1, n2 in shapes)
    assert np.array_equal(got[0][0][:nkeys], got[1][0][:nkeys])
    for k, (a, b, _, _) in enumerate(host):
        em, en = oracle.match(a, b, 0.8, True)
        for sp in (0, 1):
            assert np.array_equal(got[sp][1][k], em), (sp, k)
            assert got[sp][2][k] == en, (sp, k)


def test_column_split_plan_with_a_gate_or_prior_entries_keeps_the_finalize_kernel(ctx, oracle):
    """k_split_post decides matches from the column side: it cannot apply a stereo gate to the row it decides nor keep a
    rejected row's earlier entry.  A gate added to such a plan switches it back to merge + finalize (and removing the gate
    switches it forth); keep_prior problems never take it.  Tables, gate outputs and counts against the oracle."""
    import torch
    import plslam_amd
    r = _rng(9191)
    dev = torch.device("cuda", ctx.device)
    n1, n2 = 5000, 1400
    b = synth.random_desc(r, n2)
    a = np.concatenate([synth.noisy_copy(r, b)[0], synth.random_desc(r, n1)])[:n1]
    kl = np.ascontiguousarray(r.random((n1, 2)) * [752.0, 480.0], np.float32)
    kr = np.ascontiguousarray(r.random((n2, 2)) * [752.0, 480.0], np.float32)
    em, en = oracle.match(a, b, 0.8, True)
    kr[em[em >= 0]] = kl[em >= 0] - np.array([7.0, 0.25], np.float32)      # most matches pass the gate
    es, ed, ens = oracle.stereo_point_gate(em, kl, kr, 1.0, 1.0)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in dict(a=a, b=b, kl=kl, kr=kr).items()}
    m = torch.empty(n1, dtype=torch.int32, device=dev)
    st = torch.empty(n1, dtype=torch.int32, device=dev)
    disp = torch.empty(n1, dtype=torch.float64, device=dev)
    cnt = torch.zeros(2, dtype=torch.int32, device=dev)
    try:
        ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
        ctx.set_option("col_split", 2)
        plan = ctx.plan([(t["a"].data_ptr(), n1, t["b"].data_ptr(), n2, 0.8, True, m.data_ptr(), cnt.data_ptr())])
        gate = dict(matches_12=m.data_ptr(), f_l=t["kl"].data_ptr(), f_r=t["kr"].data_ptr(), n_l=n1, n_r=n2, lines=False,
                    max_dist_epip=1.0, min_disp=1.0, stereo_12=st.data_ptr(), disp=disp.data_ptr(), n_stereo=cnt.data_ptr() + 4)
        for stage in ("plain", "gated", "plain again"):
            plan.add_stereo_gates([gate] if stage == "gated" else [])
            m.fill_(-9); st.fill_(-9)
            plan.run(0)
            torch.cuda.synchronize()
            assert np.array_equal(m.cpu().numpy(), em), stage
            assert int(cnt[0]) == en, stage
            if stage == "gated":
                assert np.array_equal(st.cpu().numpy(), es) and int(cnt[1]) == ens
                keep = es >= 0
                assert np.array_equal(disp.cpu().numpy()[keep], ed[keep])
        plan.close()
        prior = np.where(r.random(n1) < 0.5, r.integers(0, n2, n1), -1).astype(np.int32)
        ref, nref = oracle.match_prior(a, b, 0.8, True, prior)
        got, n = ctx.match_prior(a, b, 0.8, True, prior)
        assert np.array_equal(got, ref) and n == nref
    finally:
        ctx.set_option("scan_variant", plslam_amd.SCAN_AUTO)
        ctx.set_option("col_split", 0)


def test_finalize_kernel_options(ctx, oracle):
    """Options of the finalize kernel, schedule-only: post_xcd = 1 hands every XCD a contiguous run of the block table, 2 deals
    the table to the XCDs problem by problem with padding entries (either way the row blocks of one problem gather its column
    keys through one L2; with a capped grid -- post_workgroups -- the walking workgroups stay on their XCD's entries).  Same
    tables, counts, gated associations and disparities as the defaults and as the oracle, on a plan whose table has ragged chunks."""
    import torch
    pairs, n_orb, n_lbd = 37, 300, 70                        # 37 x (2 x 2 + 2 x 1) = 222 blocks: chunks of 28, the last one short
    s = synth.stereo_stream(pairs, n_orb, n_lbd, seed=4242)
    geo = synth.stereo_geometry(s, seed=6)
    out = {}
    try:
        for xcd, cap in ((0, 0), (1, 0), (1, 21), (1, 5), (0, 9), (2, 0), (2, 13)):
            ctx.set_option("post_xcd", xcd)
            ctx.set_option("post_workgroups", cap)
            bm = frontend.StereoBatchMatcher(ctx, s, nnr_p=0.8, nnr_l=0.9, mutual=True, geometry=geo, gates=dict(synth.KITTI_GATES), n_buffers=2)
            for k in range(3):                               # (split runs: the capped grid applies to runs beside a scan)
                bm.run_overlapped(k)
            bm.synchronize_all()
            out[(xcd, cap)] = tuple(x.cpu().numpy().copy() for x in (
                bm.tables[0], bm.count_bufs[0], bm.tables[1], bm.count_bufs[1], bm.stereo_tabs[0], bm.stereo_disps[0].view(torch.int64),
                bm.stereo_cnts[0], bm.stereo_tabs[1], bm.stereo_disps[1].view(torch.int64), bm.stereo_cnts[1]))
            bm.close()
    finally:
        ctx.set_option("post_xcd", 0)
        ctx.set_option("post_workgroups", 0)
    base = out[(0, 0)]
    for k, v in out.items():
        for x, y in zip(v, base):
            assert np.array_equal(x, y), k
    sl = frontend.table_slices(n_orb, n_lbd)
    for i in range(0, pairs, 6):
        for name, d1, d2 in frontend.pair_problems(s["orb_l"], s["orb_r"], s["lbd_l"], s["lbd_r"], i):
            em, _ = oracle.match(d1, d2, 0.8 if name.startswith("orb") else 0.9, True)
            assert np.array_equal(base[0][i, sl[name]], em), (i, name)
    th = synth.KITTI_GATES
    for i in range(0, pairs, 9):                             # the gated L<->R associations of the default setting vs the oracle
        es, _, en = oracle.stereo_point_gate(base[0][i, sl["orb_lr"]], geo["kp_l"][i + 1], geo["kp_r"][i + 1], th["max_dist_epip"], th["min_disp"])
        assert np.array_equal(base[4][i, :n_orb], es) and base[6][i, 0] == en, i


def test_earlier_scan_generations_cross_check_in_a_legacy_build():
    """The product library leaves the earlier generations of the matrix-core scan out (K1e, K1g, K1h's scan kernel: `mfma_form`
    1 / 3 / 4 -> PLSLAM_ENOTSUP; their cases above skip in this process).  The cross-checks against them still run: this file
    once more, in a subprocess, against plslam_amd/lib/libplslam_hip_legacy.so (the same sources + PLSLAM_BUILD_LEGACY_SCANS=1,
    built by __graft_entry__.build(); built here if a compiler is at hand) -- nothing of it may skip but the two-GPU case."""
    import re
    import subprocess
    import sys
    if os.environ.get("PLSLAM_LEGACY_SUBPROCESS"):
        pytest.skip("this IS the subprocess")
    from plslam_amd import build as B
    if not os.path.exists(B.LEGACY_OUT):
        try:
            B.build_hip(legacy=True)
        except Exception as e:                             # noqa: BLE001 -- no compiler on this box: say so
            pytest.skip(f"no legacy build and no way to make one here: {e}")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PLSLAM_HIP_LIB_EXPERIMENT=B.LEGACY_OUT, PLSLAM_LEGACY_SUBPROCESS="1")
    res = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_match.py"), "-x", "-q", "-m", "gpu"],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=root)
    tail = res.stdout[-1500:]
    assert res.returncode == 0, tail + res.stderr[-1500:]
    m = re.search(r"(\d+) passed(?:, (\d+) skipped)?", tail)
    assert m, tail
    passed, skipped = int(m.group(1)), int(m.group(2) or 0)
    assert passed >= 300 and skipped <= 2, tail             # (skipped: this wrapper itself, and the two-GPU case where it lives here)
