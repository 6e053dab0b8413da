#!/usr/bin/env python3
"""Runs the host-to-host pipeline for a few batches (for a rocprofv3 --kernel-trace --memory-copy-trace timeline)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import plslam_amd
from plslam_amd import frontend, synth

B = 256
ctx = plslam_amd.Context(0)
st = synth.stereo_stream(B, 1500, 200)
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
hp = frontend.HostStereoPipeline(ctx, B, 1500, 200, depth=depth)
for s in range(depth + 1):
    hp.fill(s, st)
for k in range(4):
    hp.submit(k % (depth + 1))
hp.wait()
t0 = time.perf_counter()
n = 12
for k in range(n):
    hp.submit(k % (depth + 1))
hp.wait()
dt = time.perf_counter() - t0
print(f"depth {depth}: {1e3 * dt / n:.3f} ms per batch of {B} pairs = {B * n / dt:.0f} pairs/s")
hp.close()
