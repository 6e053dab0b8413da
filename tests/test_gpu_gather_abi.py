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
