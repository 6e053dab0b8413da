#!/usr/bin/env python3
"""Where K1i's wave time goes (an experiment build with -DPLSLAM_MI_PROF: tools/build_exp.py hamming_mfma_i.hip prof:-DPLSLAM_MI_PROF):
shader cycles per wave in the prologue, the tile loops and behind them, summed over all waves of a few launches of the C2 batch.
usage: PLSLAM_HIP_LIB_EXPERIMENT=build/exp/prof.so python tools/k1i_profile.py [pairs]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import plslam_amd  # noqa: E402
from plslam_amd import frontend, synth  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = plslam_amd.Context(0)
L = plslam_amd.load()
fn = L.plslam_debug_k1i_profile
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
s = synth.stereo_stream(64, 1500, 200, seed=synth.SEED0)
reps = pairs // 64
big = {k: np.concatenate([v[:1]] + [v[1:]] * reps) for k, v in s.items()}
bm = frontend.StereoBatchMatcher(ctx, big, nnr_p=0.75, nnr_l=0.75, mutual=True)
bm.plan.set_profiling(True)
st = torch.cuda.Stream()
for _ in range(2):
    bm.plan.run(st.cuda_stream)
st.synchronize()
bm.plan.elapsed()
nwg = bm.plan.info()["scan_blocks"]
buf = (ctypes.c_ulonglong * (4 * 65536))()
N = 3
for _ in range(N):
    bm.plan.run(st.cuda_stream)
st.synchronize()
a, b, n = bm.plan.elapsed()
got = fn(buf, min(nwg, 65536))
assert got > 0
v = np.frombuffer(buf, dtype=np.uint64)[:4 * got].reshape(got, 4).astype(np.float64)
raw3 = np.frombuffer(buf, dtype=np.uint64)[:4 * got].reshape(got, 4)[:, 3]
tiles = (raw3 >> np.uint64(40)).astype(np.float64)
total = (raw3 & np.uint64((1 << 40) - 1)).astype(np.float64)
ok = tiles > 0                      # (padding entries of the block table never write)
pro, loop, fin = v[ok, 0], v[ok, 1], v[ok, 2]
tiles, total = tiles[ok], total[ok]
for name, sel in (("all", np.ones(len(tiles), bool)), ("ORB (47 tiles)", tiles == 47), ("LBD (7 tiles)", tiles == 7)):
    if sel.sum() == 0:
        continue
    print(f"{name}: {int(sel.sum())} workgroups; scan {a / n:.3f} ms/launch; wave 0 per workgroup: total {total[sel].mean():.0f} cycles = prologue "
          f"{pro[sel].mean():.0f} + tile loops {loop[sel].mean():.0f} + behind them {fin[sel].mean():.0f}; {loop[sel].sum() / tiles[sel].sum():.0f} cycles of wave "
          f"time per tile (three waves share a SIMD); shares of the summed wave time: prologue {100 * pro[sel].sum() / total.sum():.1f} %, loops "
          f"{100 * loop[sel].sum() / total.sum():.1f} %, finish {100 * fin[sel].sum() / total.sum():.1f} %")
