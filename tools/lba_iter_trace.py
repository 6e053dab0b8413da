#!/usr/bin/env python3
"""plslam_lba_plan_iterate_dev at C3 sizes, repeated, for rocprofv3 --kernel-trace (tools/kt.sh): where one LM iteration's time goes.
argv[2] = "schur": the whole LM iteration with state and blocks resident instead (iterate_resident + Schur step + the host's
dense solve + back-substitution with the update applied): the kernels K19-K24 of lba_assemble.hip."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import plslam_amd  # noqa: E402
from plslam_amd import synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
ctx = plslam_amd.Context(0)
lm = synth.local_map()
cam = plslam_amd.make_cam(**synth.EUROC)
plan = plslam_amd.LbaPlan(ctx, cam, 1e-7, 10, 9, 10000, 2000, lm["pt_lm"], lm["pt_kf"], lm["pt_kf"] - 1, lm["obs_uv"],
                          lm["ls_lm"], lm["ls_kf"], lm["ls_kf"] - 1, lm["l_obs"])
if len(sys.argv) > 2 and sys.argv[2] == "schur":
    import numpy as np
    plan.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"], want_g=False)
    lam = 1e-3

    def it():
        plan.iterate_resident()
        S, b, _ = plan.schur(lam)
        plan.backsub(np.linalg.solve(S, b), apply=False, want=False)
    def it2():
        _, S, b, _n = plan.iterate_schur(lam)
        plan.apply_step(np.linalg.solve(S, b), None, apply=False)
    if len(sys.argv) > 3 and sys.argv[3] == "two":
        it = it2
    for _ in range(5):
        it()
    t0 = time.perf_counter()
    for _ in range(reps):
        it()
    print(f"LM iteration, state and blocks resident ({'iterate_schur + 54 x 54 solve + apply_step' if it is it2 else 'iterate_resident + schur + 54 x 54 solve + backsub'}): "
          f"{1e6 * (time.perf_counter() - t0) / reps:.1f} us per iteration over {reps}")
    plan.close()
    sys.exit(0)
for _ in range(5):
    plan.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"])
t0 = time.perf_counter()
for _ in range(reps):
    plan.iterate_dev(lm["T_kf_w"], lm["Xw"], lm["Lw"])
print(f"lba_plan_iterate_dev: {1e6 * (time.perf_counter() - t0) / reps:.1f} us per call over {reps} calls")
plan.close()
