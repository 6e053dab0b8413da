import ctypes as C, glob, os, time, sys
import torch
torch.cuda.set_device(0); torch.zeros(1, device="cuda")
cands = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))
rccl = C.CDLL(cands[0], mode=C.RTLD_GLOBAL)
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UniqueId()
rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
t0 = time.time(); assert rccl.ncclGetUniqueId(C.byref(uid)) == 0; t1 = time.time()
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0; t2 = time.time()
rccl.ncclCommDestroy.argtypes = [C.c_void_p]; rccl.ncclCommDestroy(comm); t3 = time.time()
print(sys.argv[1:], "uid %.2f init %.2f destroy %.2f" % (t1 - t0, t2 - t1, t3 - t2))
