#!/usr/bin/env python3
"""One or two scan streams for a stream of small steps (tools: A/B on one box).  usage: scan_streams_ab.py [pairs] [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import plslam_amd
from plslam_amd import frontend, synth
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = plslam_amd.Context(0)
st = synth.stereo_stream(pairs, 1500, 200, seed=synth.SEED0)
geo = synth.stereo_geometry(st, first_pair=0)
bms = {}
for ss in (1, 2):
    for nb in (2, 3):
        bms[(ss, nb)] = frontend.StereoBatchMatcher(ctx, st, nnr_p=0.75, nnr_l=0.75, mutual=True, n_buffers=nb, geometry=geo,
                                                    gates=dict(synth.KITTI_GATES), scan_streams=ss)
for rep in range(3):
    for key, bm in bms.items():
        for k in range(6):
            bm.run_overlapped(k)
        bm.synchronize_all()
        t0 = time.perf_counter()
        for k in range(steps):
            bm.run_overlapped(6 + k)
        th = time.perf_counter() - t0
        bm.synchronize_all()
        dt = time.perf_counter() - t0
        print(f"pairs {pairs} scan_streams {key[0]} buffers {key[1]}: {pairs * steps / dt:.0f} pairs/s, step {1e3 * dt / steps:.3f} ms (host enqueue {1e3 * th / steps:.3f} ms/step)", flush=True)
