#!/usr/bin/env python3
"""Times the scan kernel alone on the C2 batch (no verification): a scratch tool for kernel experiments.
usage: scan_time.py [scan_variant] [pairs] [mutual 0|1] [mfma_form] [fuse]        (PLSLAM_HIP_LIB_EXPERIMENT selects a build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import plslam_amd  # noqa: E402
from plslam_amd import frontend, synth  # noqa: E402

variant = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
mutual = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
form = int(sys.argv[4]) if len(sys.argv) > 4 else 0
fuse = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ctx = plslam_amd.Context(0)
ctx.set_option("scan_variant", variant)
ctx.set_option("mfma_form", form)
ctx.set_option("fuse", fuse)
for kv in filter(None, os.environ.get("PLSLAM_OPTS", "").split(",")):      # any other option: PLSLAM_OPTS=stripe_mix=4,group_cap=2
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
s = synth.stereo_stream(64, 1500, 200, seed=synth.SEED0)
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
a, b, n = bm.plan.elapsed()
print(f"[{os.environ.get('PLSLAM_OPTS', '')}] {os.path.basename(os.environ.get('PLSLAM_HIP_LIB_EXPERIMENT', 'shipped')):16s} variant {variant} form {form} fuse {fuse} pairs {pairs} "
      f"mutual {int(mutual)}: scan {a / n:.3f} ms  merge+finalize {b / n:.3f} ms")
