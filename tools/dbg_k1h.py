#!/usr/bin/env python3
"""debug: time plan runs of small batched plans with K1h (mfma_form 4)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import plslam_amd
from plslam_amd import frontend, synth
ctx = plslam_amd.Context(0)
ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
ctx.set_option("mfma_form", int(sys.argv[1]) if len(sys.argv) > 1 else 4)
for (pairs, no, nl) in [(1, 320, 70), (3, 320, 70), (3, 1500, 200), (3, 320, 320)]:
    s = synth.stereo_stream(pairs, no, nl, seed=5)
    t0 = time.time()
    bm = frontend.StereoBatchMatcher(ctx, s, nnr_p=0.75, nnr_l=0.9, mutual=True)
    torch.cuda.synchronize(); t1 = time.time()
    for it in range(3):
        ta = time.time(); tab = bm.run(); torch.cuda.synchronize(); tb = time.time()
        print(f"pairs {pairs} orb {no} lbd {nl}: create {t1 - t0:.3f}s run[{it}] {tb - ta:.4f}s", flush=True)
    bm.close()
