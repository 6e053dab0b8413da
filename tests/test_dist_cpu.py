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
