import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import plslam_amd
from plslam_amd import frontend, synth
ctx = plslam_amd.Context(0)
for pairs, steps in ((512, 40), (4096, 12)):
    st = synth.stereo_stream(pairs if pairs <= 512 else 64, 1500, 200, seed=synth.SEED0)
    if pairs > 512:
        reps = pairs // 64
        st = {k: np.concatenate([v[:1]] + [v[1:]] * reps) for k, v in st.items()}
    geo = synth.stereo_geometry(st, first_pair=0)
    for ss in (1, 2, 1, 2):
        bm = frontend.StereoBatchMatcher(ctx, st, nnr_p=0.75, nnr_l=0.75, mutual=True, n_buffers=2, geometry=geo,
                                         gates=dict(synth.KITTI_GATES), scan_streams=ss)
        for k in range(4):
            bm.run_overlapped(k)
        bm.synchronize_all()
        t0 = time.perf_counter()
        for k in range(steps):
            bm.run_overlapped(4 + k)
        bm.synchronize_all()
        dt = time.perf_counter() - t0
        same = torch.equal(bm.tables[0], bm.tables[1])
        print(f"pairs {pairs} scan_streams {ss}: {pairs * steps / dt:.0f} pairs/s, step {1e3 * dt / steps:.3f} ms, buffers equal {same}", flush=True)
        bm.close()
