"""plslam_gather_match_tables through the C ABI with a real RCCL communicator (one rank: the only
size a 1-GPU box offers; the N-rank data path is the ncclSend/ncclRecv group inside the entry point,
the sharding/ordering logic is covered on CPU by tests/test_dist_cpu.py)."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load_rccl():
    import torch
    cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*")) + \
        ["/opt/rocm/lib/librccl.so", "librccl.so"]
    for c in cands:
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
    pytest.skip("librccl.so not loadable")


def test_gather_one_rank_roundtrip(ctx):
    import torch
    import plslam_amd
    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")          # bring HIP up through torch before RCCL touches it
    rccl = _load_rccl()

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    try:
        local = torch.arange(5000, dtype=torch.int32, device="cuda") * 3 - 7
        out = torch.full((5000,), -1, dtype=torch.int32, device="cuda")
        s = torch.cuda.Stream()
        L = plslam_amd.load()
        rc = L.plslam_gather_match_tables(ctx.handle, comm, 1, 0, 0, local.data_ptr(), 5000, out.data_ptr(), s.cuda_stream)
        assert rc == 0, L.plslam_last_error()
        s.synchronize()
        assert torch.equal(out, local)
        # argument validation
        assert L.plslam_gather_match_tables(ctx.handle, comm, 1, 3, 0, local.data_ptr(), 5000, out.data_ptr(), None) == plslam_amd.capi.EINVAL
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


# ---- plslam_match_plan_set_wire16: the int16 mirror of a plan's match tables (the gather's wire format) -------------------
@pytest.mark.parametrize("pairs,n_orb,n_lbd", [(3, 96, 20), (160, 512, 64)])      # the latency kernel's plan / the matrix-core scan's
def test_finalize_kernel_writes_the_int16_wire_table(ctx, oracle, pairs, n_orb, n_lbd):
    import torch
    from plslam_amd import frontend, synth
    st = synth.stereo_stream(pairs, n_orb, n_lbd, seed=77)
    bm = frontend.StereoBatchMatcher(ctx, st, nnr_p=0.75, nnr_l=0.9, mutual=True, n_buffers=2,
                                     geometry=synth.stereo_geometry(st), gates=dict(synth.KITTI_GATES))
    try:
        assert bm.enable_wire16(True)
        for k in range(4):
            bm.run_overlapped(k)
        bm.synchronize_all()
        sl = frontend.table_slices(n_orb, n_lbd)
        for b in range(2):
            t32 = bm.tables[b].cpu().numpy()
            t16 = bm.wire16[b].cpu().numpy()
            assert t16.dtype == np.int16 and np.array_equal(t16.astype(np.int32), t32)          # the same entries, narrowed
            for i in (0, pairs - 1):
                for name, d1, d2 in frontend.pair_problems(st["orb_l"], st["orb_r"], st["lbd_l"], st["lbd_r"], i):
                    em = oracle.match(d1, d2, 0.75 if name.startswith("orb") else 0.9, True)[0]
                    assert np.array_equal(t32[i, sl[name]], em)
        # removing the mirror: the int16 tables are no longer written
        keep = [w.clone() for w in bm.wire16]
        w16 = bm.wire16
        for w in w16:
            w.fill_(-5)
        bm.enable_wire16(False)
        bm.run_overlapped(0)
        bm.synchronize_all()
        assert all(bool((w == -5).all()) for w in w16) and keep[0].shape == w16[0].shape
    finally:
        bm.close()


def test_wire16_is_refused_where_another_kernel_writes_the_table(ctx):
    """A column-split plan (a few LARGE problems: k_split_post decides the entries) has no finalize kernel to mirror from."""
    import torch
    import plslam_amd
    from plslam_amd import synth
    r = np.random.Generator(np.random.PCG64(3))
    a = torch.from_numpy(synth.random_desc(r, 6000)).cuda()
    b = torch.from_numpy(synth.random_desc(r, 1500)).cuda()
    m = torch.empty(6000, dtype=torch.int32, device="cuda")
    w = torch.empty(6000, dtype=torch.int16, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    plan = ctx.plan([(a.data_ptr(), 6000, b.data_ptr(), 1500, 0.75, True, m.data_ptr(), cnt.data_ptr())])
    try:
        if plan.info()["scan_variant"] == plslam_amd.SCAN_MFMA:
            with pytest.raises(Exception):
                plan.set_wire16(m.data_ptr(), w.data_ptr(), 6000)
        plan.set_wire16(m.data_ptr(), 0, 6000)              # removing a mirror that is not there is fine
    finally:
        plan.close()


def test_gather_step_narrows_by_copy_where_the_plan_has_no_finalize_kernel(ctx):
    """40 pairs of 512 + 64 features: few enough waves for AUTO to take the column-split form (k_split_post writes the tables):
    the matcher reports that no mirror can be had, and the table pipeline narrows with its copy."""
    from plslam_amd import frontend, synth
    st = synth.stereo_stream(40, 512, 64, seed=5)
    bm = frontend.StereoBatchMatcher(ctx, st, nnr_p=0.75, nnr_l=0.9, mutual=True, n_buffers=2)
    try:
        got = bm.enable_wire16(True)
        assert got == (bm.wire16 is not None)
        if not got:
            bm.run_overlapped(0)
            bm.synchronize_all()
            pipe = frontend.TableGatherPipeline(bm.B, bm.stride, 512, 1, 0, nbuf=2, device=bm.dev, compact=True)
            assert pipe.send is not None and pipe.send[0].dtype.itemsize == 2
    finally:
        bm.close()


# ---- plslam_match_plan_step_gather: the N > 1 step as ONE C-ABI call (round 6) -----------------------------------------------
def _make_comm(rccl):
    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    return comm


@pytest.mark.parametrize("wire", ["int32", "int16"])
@pytest.mark.parametrize("with_comm", [False, True])
@pytest.mark.parametrize("pairs,n_orb,n_lbd", [(3, 96, 20), (160, 512, 64)])
def test_native_step_gathers_what_the_torch_step_gathers(ctx, oracle, wire, with_comm, pairs, n_orb, n_lbd):
    """PipelinedGather with native=True (plan run + send / receive group + widening in one call into the C ABI) against the torch
    path on the same matcher, both wire formats; with a real one-rank RCCL communicator the rank's table travels through
    ncclSend / ncclRecv to itself, without one it is copied.  Six steps through two buffers: every step waits for the gather of
    the buffer it rewrites.  Gathered tables = the matcher's own tables = the oracle's."""
    import torch
    import plslam_amd
    from plslam_amd import frontend, synth
    torch.cuda.set_device(0)
    st = synth.stereo_stream(pairs, n_orb, n_lbd, seed=91)
    bm = frontend.StereoBatchMatcher(ctx, st, nnr_p=0.75, nnr_l=0.9, mutual=True, n_buffers=2,
                                     geometry=synth.stereo_geometry(st), gates=dict(synth.KITTI_GATES))
    comm, rccl = None, None
    try:
        pg = frontend.PipelinedGather(bm, 1, 0, root=0, compact=(wire == "int16"), native=True)
        if wire == "int16" and not pg.kernel_wire16:
            pytest.skip("this plan has no finalize kernel to write the int16 mirror (column-split form)")
        assert pg.native
        if with_comm:
            rccl = _load_rccl()
            path = next((l.split()[-1] for l in open("/proc/self/maps") if "librccl.so" in l), None)
            if path:
                plslam_amd.load().plslam_rccl_use(path.encode())      # (EINVAL if an earlier test already made the library load it)
            comm = _make_comm(rccl)
            pg._comm = comm.value                        # (the process-group lookup is bench.py's; here a communicator of our own)
        for k in range(6):
            pg.step(k)
        pg.finish()
        bm.synchronize_all()
        assert pg.host_ms_per_step() > 0
        sl = frontend.table_slices(n_orb, n_lbd)
        for b in range(2):
            got = pg.gathered(b).cpu().numpy()
            assert got.dtype == np.int32 and np.array_equal(got, bm.tables[b].cpu().numpy())
            for i in (0, pairs - 1):
                for name, d1, d2 in frontend.pair_problems(st["orb_l"], st["orb_r"], st["lbd_l"], st["lbd_r"], i):
                    assert np.array_equal(got[i, sl[name]], oracle.match(d1, d2, 0.75 if name.startswith("orb") else 0.9, True)[0])
        pg.close()
        # the torch path on the same matcher gathers the same tables
        pt = frontend.PipelinedGather(bm, 1, 0, root=0, compact=(wire == "int16"), native=False)
        assert not pt.native
        import torch.distributed as dist
        if not dist.is_initialized():
            import socket
            s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        for k in range(2):
            pt.step(k)
        pt.finish()
        bm.synchronize_all()
        for b in range(2):
            assert np.array_equal(pt.gathered(b).cpu().numpy(), bm.tables[b].cpu().numpy())
        pt.close()
    finally:
        bm.close()
        if comm is not None:
            rccl.ncclCommDestroy.argtypes = [C.c_void_p]
            rccl.ncclCommDestroy(comm)


def test_native_step_argument_validation(ctx):
    import torch
    from plslam_amd import frontend, synth
    st = synth.stereo_stream(3, 96, 20, seed=2)
    bm = frontend.StereoBatchMatcher(ctx, st, nnr_p=0.75, nnr_l=0.9, mutual=True, n_buffers=2)
    try:
        p, t = bm.plans[0], bm.tables[0]
        recv = torch.empty_like(t)
        s0, s1 = bm.streams[0].cuda_stream, bm.streams[1].cuda_stream
        ok = p.make_gather_step(0, 1, 0, 0, 4, t.data_ptr(), t.numel(), recv.data_ptr(), 0, s0, s1)
        p.step_gather(ok)
        p.gather_sync()
        assert torch.equal(recv, t)
        for bad in (p.make_gather_step(0, 1, 0, 0, 3, t.data_ptr(), t.numel(), recv.data_ptr(), 0, s0, s1),       # wire_bytes
                    p.make_gather_step(0, 1, 1, 0, 4, t.data_ptr(), t.numel(), recv.data_ptr(), 0, s0, s1),       # rank
                    p.make_gather_step(0, 2, 0, 0, 4, t.data_ptr(), t.numel(), recv.data_ptr(), 0, s0, s1),       # two ranks, no communicator
                    p.make_gather_step(0, 1, 0, 0, 4, t.data_ptr(), t.numel(), 0, 0, s0, s1),                     # root without a receive buffer
                    p.make_gather_step(0, 1, 0, 0, 4, t.data_ptr(), t.numel(), recv.data_ptr(), 0, s0, s0)):      # one stream for both
            with pytest.raises(Exception):
                p.step_gather(bad)
    finally:
        bm.close()
