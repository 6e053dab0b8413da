#!/usr/bin/env python3
"""C3 (one local map against one frame: 10 000 x 1500 ORB + 2 000 x 200 LBD, mutual) as ONE plan: time per run and the
scan / post-scan split from the plan's own events.  usage: c3_time.py [col_split 0|1|2] [scan_variant] [mfma_form] [graph 0|1|2] [split_post 0|1]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import plslam_amd
from plslam_amd import synth

split = int(sys.argv[1]) if len(sys.argv) > 1 else 0
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
form = int(sys.argv[3]) if len(sys.argv) > 3 else 0
graph = int(sys.argv[4]) if len(sys.argv) > 4 else 1
split_post = int(sys.argv[5]) if len(sys.argv) > 5 else 0
ctx = plslam_amd.Context(0)
ctx.set_option("col_split", split)
ctx.set_option("scan_variant", variant)
ctx.set_option("mfma_form", form)
ctx.set_option("graph", graph)
ctx.set_option("split_post", split_post)
for kv in filter(None, os.environ.get("PLSLAM_OPTS", "").split(",")):      # any other option: PLSLAM_OPTS=split_target=2,split_min_tiles=3
    ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
dev = torch.device("cuda", 0)
r = np.random.Generator(np.random.PCG64(31))
frame_p = synth.random_desc(r, 1500)
map_p = np.concatenate([synth.noisy_copy(r, frame_p)[0], synth.random_desc(r, 8500)])
frame_l = synth.random_desc(r, 200)
map_l = np.concatenate([synth.noisy_copy(r, frame_l)[0], synth.random_desc(r, 1800)])
t = {k: torch.from_numpy(v).to(dev) for k, v in dict(mp=map_p, fp=frame_p, ml=map_l, fl=frame_l).items()}
m_p = torch.empty(10000, dtype=torch.int32, device=dev)
m_l = torch.empty(2000, dtype=torch.int32, device=dev)
cnt = torch.zeros(2, dtype=torch.int32, device=dev)
plan = ctx.plan([(t["mp"].data_ptr(), 10000, t["fp"].data_ptr(), 1500, 0.75, True, m_p.data_ptr(), cnt.data_ptr()),
                 (t["ml"].data_ptr(), 2000, t["fl"].data_ptr(), 200, 0.75, True, m_l.data_ptr(), cnt.data_ptr() + 4)])
st = torch.cuda.Stream(device=dev)
for _ in range(20):
    plan.run(st.cuda_stream)
st.synchronize()
ref_p, ref_l = m_p.clone(), m_l.clone()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.cuda.stream(st):
    e0.record(st)
    for _ in range(200):
        plan.run(st.cuda_stream)
    e1.record(st)
st.synchronize()
plan.set_profiling(True)
plan.elapsed()
for _ in range(20):
    plan.run(st.cuda_stream)
    st.synchronize()
a, b, n = plan.elapsed()
plan.set_profiling(False)
import time
t0 = time.perf_counter()
for _ in range(200):
    plan.run(st.cuda_stream)
    st.synchronize()
wall = (time.perf_counter() - t0) / 200
assert torch.equal(ref_p, m_p) and torch.equal(ref_l, m_l)
print(f"graph {graph}: run + synchronize from the host {1e6 * wall:.1f} us; ", end="")
print(f"[{os.environ.get('PLSLAM_OPTS', '')}] split_post {split_post} col_split {split} variant {variant} form {form}: {1e3 * e0.elapsed_time(e1) / 200:.1f} us per back-to-back run; serial runs: scan "
      f"{1e3 * a / n:.1f} us, post-scan {1e3 * b / n:.1f} us; info {plan.info()}")
