#!/usr/bin/env python3
"""The LBA row kernels' streaming launches of bench.py's secondary.c3 record on their own (256 point maps / 1024 line maps
per launch = ~1.5 GB moved, each map with its own landmark array; `lba_stream.py 64 256` = the ~0.4 GB footprint), for rocprofv3: `rocprofv3 --kernel-trace --stats -- python tools/lba_stream.py` and the FETCH_SIZE /
WRITE_SIZE passes (tools/pmc_passes.sh).  Prints the moved-byte model per launch next to the event-timed rate."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import plslam_amd  # noqa: E402
from plslam_amd import synth  # noqa: E402


def main():
    ctx = plslam_amd.Context(0)
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)
    lm = synth.local_map()
    cam = plslam_amd.make_cam(**synth.EUROC)
    g = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in lm.items()}
    reps_of = {"point": int(sys.argv[1]) if len(sys.argv) > 1 else 256, "line": int(sys.argv[2]) if len(sys.argv) > 2 else 1024}
    # (the number of pose matrices, stated to the kernels: 0 = not stated, as before round 6; env PLSLAM_STREAM_SLOTS overrides)
    NSLOTS = int(os.environ.get("PLSLAM_STREAM_SLOTS", lm["T_kf_w"].shape[0]))
    out = {}
    for kind, n, nlm, lmk, xk, keys, nl in (("point", lm["pt_lm"].shape[0], lm["Xw"].shape[0], "pt_lm", "Xw", ("obs_uv", "pt_kf"), 3),
                                             ("line", lm["ls_lm"].shape[0], lm["Lw"].shape[0], "ls_lm", "Lw", ("l_obs", "ls_kf"), 6)):
        reps = reps_of[kind]
        big = {k: torch.cat([g[k]] * reps) for k in keys}
        idx = torch.cat([g[lmk] + r * nlm for r in range(reps)])
        X = torch.cat([g[xk]] * reps)
        nb = n * reps
        Jp = torch.empty((nb, 6), dtype=torch.float64, device=dev)
        Jl = torch.empty((nb, nl), dtype=torch.float64, device=dev)
        rr = torch.empty(nb, dtype=torch.float64, device=dev)
        ww = torch.empty(nb, dtype=torch.float64, device=dev)
        if kind == "point":
            fn = lambda: ctx.lba_point_rows_dev(cam, 1e-7, g["T_kf_w"].data_ptr(), X.data_ptr(), big["obs_uv"].data_ptr(), idx.data_ptr(),  # noqa: E731
                                                big["pt_kf"].data_ptr(), nb, Jp.data_ptr(), Jl.data_ptr(), rr.data_ptr(), ww.data_ptr(), st.cuda_stream, n_pose_slots=NSLOTS)
            moved = 8 + 16 + 88 + 24.0 * nlm / n
        else:
            fn = lambda: ctx.lba_line_rows_dev(cam, 1e-7, False, g["T_kf_w"].data_ptr(), X.data_ptr(), big["l_obs"].data_ptr(), idx.data_ptr(),  # noqa: E731
                                               big["ls_kf"].data_ptr(), nb, Jp.data_ptr(), Jl.data_ptr(), rr.data_ptr(), ww.data_ptr(), st.cuda_stream, n_pose_slots=NSLOTS)
            moved = 8 + 24 + 112 + 48.0 * nlm / n
        for _ in range(3):
            fn()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(10):
            fn()
        e1.record(st)
        st.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out[kind] = {"maps_per_launch": reps, "rows_per_launch": nb, "bytes_per_row_moved": moved, "bytes_per_launch_moved": nb * moved, "launches": 13,
                     "ms_per_launch_events": ms, "GBps_moved": nb * moved / (ms * 1e-3) / 1e9}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
