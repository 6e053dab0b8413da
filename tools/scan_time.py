#!/usr/bin/env python3
"""Times the scan kernel alone on the C2 batch (no verification): a scratch tool for kernel experiments.
usage: scan_time.py [scan_variant] [pairs] [mutual 0|1]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import plslam_amd  # noqa: E402
from plslam_amd import frontend, synth  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
mutual = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
ctx = plslam_amd.Context(0)
ctx.set_option("scan_variant", variant)
s = synth.stereo_stream(64, 1500, 200, seed=synth.SEED0)
s = {k: (v[:1].repeat(1, 0) if False else v) for k, v in s.items()}
import numpy as np  # noqa: E402
reps = pairs // 64
big = {k: np.concatenate([v[:1]] + [v[1:]] * reps) for k, v in s.items()}
bm = frontend.StereoBatchMatcher(ctx, big, nnr_p=0.75, nnr_l=0.75, mutual=mutual)
bm.plan.set_profiling(True)
st = torch.cuda.Stream()
for _ in range(3):
    bm.plan.run(st.cuda_stream)
st.synchronize()
bm.plan.elapsed()
for _ in range(5):
    bm.plan.run(st.cuda_stream)
st.synchronize()
print("variant", variant, "pairs", pairs, "mutual", mutual, "elapsed(scan_ms, merge+finalize_ms, runs):", bm.plan.elapsed(), bm.plan.info()["scan_variant"])
