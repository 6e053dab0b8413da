#!/usr/bin/env python3
"""The N > 1 step on a one-rank RCCL group: step time and host enqueue time of PipelinedGather around a 512-pair matcher, beside
the plain overlapped step; modes: gather (own communication stream, int16 wire), gather_int32, gather_stage (wait + widening on
the matcher's stage stream: PipelinedGather's default); PG_HIGH=1: a high-priority process-group stream.
usage: gather_step_probe.py [pairs] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
import plslam_amd
from plslam_amd import frontend, synth
pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
t0 = time.perf_counter()
pg_high = int(os.environ.get("PG_HIGH", "0"))      # PG_HIGH=1: the collective's own stream comes from the high-priority pool
opts = None
if pg_high:
    from torch.distributed import ProcessGroupNCCL
    opts = ProcessGroupNCCL.Options()
    opts.is_high_priority_stream = True
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, pg_options=opts)
print(f"process group up in {time.perf_counter() - t0:.2f} s (high-priority collective stream: {pg_high})", flush=True)
ctx = plslam_amd.Context(0)
st = synth.stereo_stream(pairs, 1500, 200, seed=synth.SEED0)
geo = synth.stereo_geometry(st, first_pair=0)
for nb in (2, 3):
    bm = frontend.StereoBatchMatcher(ctx, st, nnr_p=0.75, nnr_l=0.75, mutual=True, n_buffers=nb, geometry=geo, gates=dict(synth.KITTI_GATES))
    for mode in ("plain", "gather", "gather_int32", "gather_stage", "plain", "gather", "gather_stage", "gather", "gather_stage"):
        pg = None if mode == "plain" else frontend.PipelinedGather(bm, 1, 0, root=0, compact=(mode != "gather_int32"),
                                                                   comm_on_stage_stream=(mode == "gather_stage"))     # "gather" / "gather_int32": a communication stream of its own
        step = (lambda k: bm.run_overlapped(k)) if pg is None else pg.step
        for k in range(6):
            step(k)
        (pg.finish() if pg else None); bm.synchronize_all()
        t0 = time.perf_counter()
        for k in range(steps):
            step(6 + k)
        th = time.perf_counter() - t0
        (pg.finish() if pg else None); bm.synchronize_all()
        dt = time.perf_counter() - t0
        print(f"buffers {nb} {mode:13s}: {pairs * steps / dt:.0f} pairs/s, step {1e3 * dt / steps:.3f} ms, host enqueue {1e3 * th / steps:.3f} ms/step", flush=True)
    bm.close()
dist.destroy_process_group()
