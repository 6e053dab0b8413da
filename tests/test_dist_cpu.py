"""world_size-2 gloo test of the multi-GPU path's host logic: contiguous sharding
of stereo pairs with a one-pair halo, per-rank match tables, gather to rank 0 in rank order.
On CPU the per-rank tables come from the oracle (the checker); the thing under test is the
partition + table layout + gather, which is what runs unchanged over RCCL on the GPUs."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from plslam_amd import frontend, synth

N_PAIRS, N_ORB, N_LBD = 5, 48, 12   # odd count: the last shard is ragged


def _oracle_tables(stream):
    from oracle import oracle as O
    B = stream["orb_l"].shape[0] - 1
    sl = frontend.table_slices(N_ORB, N_LBD)
    tab = np.full((B, frontend.table_stride(N_ORB, N_LBD)), -2, np.int32)
    for i in range(B):
        for name, d1, d2 in frontend.pair_problems(stream["orb_l"], stream["orb_r"], stream["lbd_l"],
                                                   stream["lbd_r"], i):
            tab[i, sl[name]] = O.match(d1, d2, 0.75 if name.startswith("orb") else 0.9, True)[0]
    return tab


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = frontend.shard_range(N_PAIRS, world, rank)
    per = -(-N_PAIRS // world)
    stream = synth.stereo_stream(hi - lo, N_ORB, N_LBD, seed=99, first_pair=lo)
    tab = _oracle_tables(stream)
    pad = np.full((per, tab.shape[1]), -3, np.int32)   # fixed-stride payload; ragged tail padded
    pad[: tab.shape[0]] = tab
    got = frontend.gather_tables(torch.from_numpy(pad), world, rank, root=0)
    if rank == 0:
        np.save(out, got.numpy())
    dist.destroy_process_group()


def test_shard_range_partitions():
    for n in (0, 1, 5, 8, 4096):
        for w in (1, 2, 3, 8):
            rs = [frontend.shard_range(n, w, r) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(b - a for a, b in rs) <= -(-n // w) if n else True


def test_gloo_world2_gather_matches_single_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    full = _oracle_tables(synth.stereo_stream(N_PAIRS, N_ORB, N_LBD, seed=99, first_pair=0))
    per = -(-N_PAIRS // 2)
    assert got.shape == (2 * per, full.shape[1])
    assert np.array_equal(got[:N_PAIRS], full)          # rank order == pair order, halo handled
    assert (got[N_PAIRS:] == -3).all()
    assert (full != -2).all()


# ---- the pipelined gather of bench.py's N > 1 path, driven over gloo ------------------------------------------------------
def _pipeline_worker(rank, world, port, out, compact):
    """Each rank 'computes' step k's table as its oracle table + k (so that a buffer delivered for the wrong step is
    visible), through the SAME TableGatherPipeline the RCCL path uses: int16 wire format through uint8 views,
    widening on the root, two buffers reused by steps k and k + 2."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = frontend.shard_range(N_PAIRS, world, rank)
    per = -(-N_PAIRS // world)
    stream = synth.stereo_stream(hi - lo, N_ORB, N_LBD, seed=99, first_pair=lo)
    tab = _oracle_tables(stream)
    pad = np.full((per, tab.shape[1]), -3, np.int32)              # ragged tail: filler rows survive the int16 trip
    pad[: tab.shape[0]] = tab
    base = torch.from_numpy(pad)
    pipe = frontend.TableGatherPipeline(per, tab.shape[1], max(N_ORB, N_LBD), world, rank, root=0, nbuf=2, compact=compact)
    assert pipe.wire == (torch.int16 if compact else torch.int32)
    tables = [torch.empty_like(base) for _ in range(2)]
    seen = {}
    for k in range(5):
        b = k % 2
        pipe.before_overwrite(b)                                   # step k - 2's gather has consumed tables[b]
        tables[b].copy_(torch.where(base >= 0, base + k, base))    # "compute" step k
        pipe.submit(b, tables[b])
        if k >= 1:                                                 # while step k is in flight, step k - 1 is complete
            pipe.before_overwrite((k - 1) % 2)
            g = pipe.gathered((k - 1) % 2)
            if rank == 0:
                seen[k - 1] = g.numpy().copy()
    pipe.finish()
    if rank == 0:
        seen[4] = pipe.gathered(0).numpy().copy()
        np.savez(out, **{f"step{k}": v for k, v in seen.items()})
    else:
        assert pipe.gathered(0) is None
    dist.destroy_process_group()


@pytest.mark.parametrize("compact", [True, False])
def test_gloo_world2_pipelined_gather(tmp_path, compact):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "pipe.npz")
    mp.spawn(_pipeline_worker, args=(2, port, out, compact), nprocs=2, join=True)
    got = np.load(out)
    full = _oracle_tables(synth.stereo_stream(N_PAIRS, N_ORB, N_LBD, seed=99, first_pair=0))
    per = -(-N_PAIRS // 2)
    for k in range(5):
        g = got[f"step{k}"]
        assert g.dtype == np.int32 and g.shape == (2 * per, full.shape[1])
        assert np.array_equal(g[:N_PAIRS], np.where(full >= 0, full + k, full)), k      # the table of step k, rank order
        assert (g[N_PAIRS:] == -3).all()


def _verify_worker(rank, world, port, out):
    """bench.py's root-side check of a gathered table (frontend.verify_gathered_tables): every rank's shard is regenerated
    from (seed, first_pair = rank * B) on the root.  One rank's table is corrupted in one entry: exactly that (rank, pair,
    problem) must be reported."""
    from oracle import oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = 3
    stream = synth.stereo_stream(B, N_ORB, N_LBD, seed=synth.SEED0, first_pair=rank * B)
    tab = torch.from_numpy(_oracle_tables(stream))
    if rank == 1:
        tab[2, N_ORB + 5] = (tab[2, N_ORB + 5] + 1) % N_ORB        # pair 2 of rank 1, problem orb_pc
    pipe = frontend.TableGatherPipeline(B, tab.shape[1], max(N_ORB, N_LBD), world, rank, root=0, nbuf=1)
    pipe.submit(0, tab)
    pipe.finish()
    if rank == 0:
        full = pipe.gathered(0).numpy()
        match = lambda d1, d2, nnr: O.match(d1, d2, nnr, True)[0]   # noqa: E731
        bad = frontend.verify_gathered_tables(full, world, B, N_ORB, N_LBD, 0.75, 0.9, [0, 2], match, local_stream=stream)
        ok = frontend.verify_gathered_tables(full, world, B, N_ORB, N_LBD, 0.75, 0.9, [0, 1], match, local_stream=stream)
        with open(out, "w") as f:
            f.write(repr((bad, ok)))
    dist.destroy_process_group()


def test_gloo_world2_root_side_verification(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "verify.txt")
    mp.spawn(_verify_worker, args=(2, port, out), nprocs=2, join=True)
    bad, ok = eval(open(out).read())
    assert bad == [(1, 2, "orb_pc")] and ok == []


# ---- bench.py --scaling strong: ONE batch per step, contiguous shards of total / N pairs (BASELINE config 4) ----------------
def _strong_worker(rank, world, port, out, total):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = frontend.shard_range(total, world, rank)              # bench.py's rule: first_pair = lo, B = hi - lo
    B = hi - lo
    stream = synth.stereo_stream(B, N_ORB, N_LBD, seed=synth.SEED0, first_pair=lo)   # carries the halo: pair lo - 1's left rows
    tab = torch.from_numpy(_oracle_tables(stream))
    pipe = frontend.TableGatherPipeline(B, tab.shape[1], max(N_ORB, N_LBD), world, rank, root=0, nbuf=2)
    for k in range(3):                                             # three steps through two buffers
        pipe.before_overwrite(k % 2)
        pipe.submit(k % 2, tab)
    pipe.finish()
    if rank == 0:
        np.save(out, pipe.gathered(0).numpy())
    dist.destroy_process_group()


def test_gloo_world2_strong_scaling_shards(tmp_path):
    """The strong-scaling partition of bench.py (--scaling strong): 6 pairs over 2 ranks = shards [0, 3) and [3, 6).  Pair 3's
    prev <-> curr problems need pair 2's LEFT descriptors -- the halo rank 1 regenerates locally -- so the gathered table must
    equal the unsharded stream's, and bench.py's root-side verification (which regenerates rank r's shard from first_pair =
    r * B) must accept it."""
    from oracle import oracle as O
    total = 6
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "strong.npy")
    mp.spawn(_strong_worker, args=(2, port, out, total), nprocs=2, join=True)
    got = np.load(out)
    full = _oracle_tables(synth.stereo_stream(total, N_ORB, N_LBD, seed=synth.SEED0, first_pair=0))
    assert got.dtype == np.int32 and np.array_equal(got, full)
    bad = frontend.verify_gathered_tables(got, 2, total // 2, N_ORB, N_LBD, 0.75, 0.9, [0, 1, 2],
                                          lambda d1, d2, nnr: O.match(d1, d2, nnr, True)[0])
    assert bad == []
    got[3, 5] ^= 1                                                 # the first pair of rank 1's shard (the halo consumer)
    bad = frontend.verify_gathered_tables(got, 2, total // 2, N_ORB, N_LBD, 0.75, 0.9, [0, 1, 2],
                                          lambda d1, d2, nnr: O.match(d1, d2, nnr, True)[0])
    assert [b[:2] for b in bad] == [(1, 0)]


# ---- BASELINE config 4's exact partition at world 8: 4096 pairs -> 8 x 512, a halo at every shard boundary ---------------------
def _world8_worker(rank, world, port, out, total):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = frontend.shard_range(total, world, rank)
    per = -(-total // world)                                       # fixed-stride payload: a ragged last shard is padded
    stream = synth.stereo_stream(hi - lo, N_ORB, N_LBD, seed=synth.SEED0, first_pair=lo)
    tab = np.full((per, frontend.table_stride(N_ORB, N_LBD)), -3, np.int32)
    tab[: hi - lo] = _oracle_tables(stream)
    pipe = frontend.TableGatherPipeline(per, tab.shape[1], max(N_ORB, N_LBD), world, rank, root=0, nbuf=2)
    for k in range(2):
        pipe.before_overwrite(k % 2)
        pipe.submit(k % 2, torch.from_numpy(tab))
    pipe.finish()
    if rank == 0:
        np.save(out, pipe.gathered(1).numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4096, 4099])
def test_gloo_world8_config4_partition(tmp_path, total):
    """BASELINE config 4 as written, on eight gloo ranks: 4096 pairs in contiguous shards of 512 (and 4099: seven shards of 513
    and a ragged one of 508, padded on the wire).  Every shard but the first starts with a pair whose prev <-> curr problems need
    the LEFT descriptors of the pair before it -- the one-pair halo each rank regenerates locally -- so rank 0's
    check of 64 pairs of EVERY rank (the shard's ends and four runs in between: bench.py's root-side verification, with the
    per-rank first pairs of the contiguous partition) covers all seven boundaries; the gathered table is also compared whole with
    the unsharded stream's at the boundaries."""
    from oracle import oracle as O
    world = 8
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "w8.npy")
    mp.spawn(_world8_worker, args=(world, port, out, total), nprocs=world, join=True)
    got = np.load(out)
    per = -(-total // world)
    assert got.shape == (world * per, frontend.table_stride(N_ORB, N_LBD)) and got.dtype == np.int32
    firsts = [frontend.shard_range(total, world, r)[0] for r in range(world)]
    sizes = [frontend.shard_range(total, world, r)[1] - firsts[r] for r in range(world)]
    assert sum(sizes) == total and firsts == [r * per for r in range(world)] and (total % world == 0) == (len(set(sizes)) == 1)
    fn = lambda d1, d2, nnr: O.match(d1, d2, nnr, True)[0]          # noqa: E731
    # 64 pairs of every rank (a ragged shard: of the pairs it holds), as bench.py's root does
    for r in range(world):
        sample = frontend.spread_sample(sizes[r], 64)
        assert len(sample) == 64 and sample[0] == 0 and sample[-1] == sizes[r] - 1
        one = got[r * per:(r + 1) * per]
        bad = frontend.verify_gathered_tables(one, 1, per, N_ORB, N_LBD, 0.75, 0.9, sample, fn, first_pairs=[firsts[r]])
        assert bad == [], (r, bad[:4])
    # the padding of the ragged shard arrived as padding
    for r in range(world):
        assert (got[r * per + sizes[r]:(r + 1) * per] == -3).all()
    # every boundary pair against the UNSHARDED stream (the halo consumer is the first pair of ranks 1..7)
    for r in range(1, world):
        st = synth.stereo_stream(2, N_ORB, N_LBD, seed=synth.SEED0, first_pair=firsts[r] - 1)      # pairs firsts[r] - 1 and firsts[r]
        ref = _oracle_tables(st)
        assert np.array_equal(got[(r - 1) * per + sizes[r - 1] - 1], ref[0]) and np.array_equal(got[r * per], ref[1]), r
    # a corrupted boundary entry is found and named
    got[3 * per, 7] ^= 1
    bad = frontend.verify_gathered_tables(got[3 * per:4 * per], 1, per, N_ORB, N_LBD, 0.75, 0.9, [0, 1], fn, first_pairs=[firsts[3]])
    assert [b[:2] for b in bad] == [(0, 0)]


def test_bench_launches_itself_at_eight_ranks():
    """`python bench.py --gpus 8 --launch-check`: the road the driver's SCALE run takes (one rank per GPU under
    torch.distributed.run on this node) -- eight ranks rendezvous over gloo, rank 0 prints the one line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "5", "--launch-check"],
                         capture_output=True, text=True, timeout=900, env=dict(env, OMP_NUM_THREADS="1"))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d == {"launch_check": True, "world_size": 8, "rank_sum": 36, "argv": ["--gpus", "8", "--steps", "5", "--launch-check"]}
    assert "--nproc-per-node=8" in res.stderr


# ---- bench.py launches itself at N > 1 (`python bench.py --gpus 8` exited with a usage message) ----------------
def test_bench_launches_itself_at_two_ranks():
    """`python bench.py --gpus 2 ...` with no launcher around it re-executes under torch.distributed.run; --launch-check
    stops after the rendezvous (gloo: no GPU here) -- one JSON line on stdout, from rank 0, the arguments intact."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "7", "--launch-check"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d == {"launch_check": True, "world_size": 2, "rank_sum": 3, "argv": ["--gpus", "2", "--steps", "7", "--launch-check"]}
    assert "torch.distributed.run" in res.stderr and "--nproc-per-node=2" in res.stderr


def test_root_side_check_of_strong_scaling_shards():
    """verify_gathered_tables with per-rank first pairs (config 4's contiguous shards) and a sample of runs: clean tables pass,
    one wrong entry in the second rank's shard is found and named."""
    from oracle import oracle as O
    world, total = 2, 12
    full = _oracle_tables(synth.stereo_stream(total, N_ORB, N_LBD, seed=synth.SEED0, first_pair=0))
    # (_oracle_tables uses nnr 0.75 / 0.9)
    per = total // world
    firsts = [frontend.shard_range(total, world, r)[0] for r in range(world)]
    sample = frontend.spread_sample(per, 4)
    assert sample == sorted(set(sample)) and {0, per - 1} <= set(sample) and max(sample) < per
    fn = lambda d1, d2, nnr: O.match(d1, d2, nnr, True)[0]     # noqa: E731
    assert frontend.verify_gathered_tables(full, world, per, N_ORB, N_LBD, 0.75, 0.9, sample, fn, first_pairs=firsts) == []
    bad = full.copy()
    bad[per + sample[-1], 3] += 1
    got = frontend.verify_gathered_tables(bad, world, per, N_ORB, N_LBD, 0.75, 0.9, sample, fn, first_pairs=firsts)
    assert got == [(1, sample[-1], "orb_lr")]
    assert len(frontend.spread_sample(4096, 64)) == 64 and len(frontend.spread_sample(40, 64)) == 40
