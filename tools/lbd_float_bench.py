#!/usr/bin/env python3
"""K18 (BinaryDescriptor::computeLBD) timing on one MI355X: one frame's 200 lines and a batch of 65 536 lines on a
752 x 480 octave, device-resident, against the oracle on one host thread.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plslam_amd  # noqa: E402
from plslam_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    import torch
    ctx = plslam_amd.Context(0)
    dev = torch.device("cuda", 0)
    r = np.random.Generator(np.random.PCG64(4))
    W, H = 752, 480
    dx, dy = synth.gradient_images(r, W, H)
    tx, ty = torch.from_numpy(dx).to(dev), torch.from_numpy(dy).to(dev)
    out = {}
    st = torch.cuda.Stream(device=dev)
    for n in (200, 65536):
        lines = synth.lbd_lines(r, n, W, H, min_len=20, max_len=200, dtype=O.LBD_LINE_DTYPE)
        f = torch.empty((n, 72), dtype=torch.float32, device=dev)
        for _ in range(3):
            ctx.lbd_compute_dev(tx.data_ptr(), ty.data_ptr(), W, H, lines, f.data_ptr(), 7, st.cuda_stream)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if n <= 1000 else 5
        t0 = time.perf_counter()
        e0.record(st)
        for _ in range(reps):
            ctx.lbd_compute_dev(tx.data_ptr(), ty.data_ptr(), W, H, lines, f.data_ptr(), 7, st.cuda_stream)
        e1.record(st)
        st.synchronize()
        wall = (time.perf_counter() - t0) / reps
        m = min(n, 2000)
        t1 = time.perf_counter()
        ref = O.lbd_compute(dx, dy, lines[:m])
        cpu = (time.perf_counter() - t1) / m
        assert np.array_equal(f[:m].cpu().numpy().view(np.uint32), ref.view(np.uint32))
        px = float(lines["num_pixels"].astype(np.int64).sum()) * 63
        out[f"lines_{n}"] = {"call_us_incl_line_upload": wall * 1e6, "gpu_us_events": e0.elapsed_time(e1) * 1e3 / reps,
                             "lines_per_s": n / wall, "pixel_visits": px, "pixel_visits_per_s": px / wall,
                             "cpu_oracle_1thread_lines_per_s": 1.0 / cpu}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
