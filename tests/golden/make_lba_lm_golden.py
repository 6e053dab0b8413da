#!/usr/bin/env python3
"""tests/golden/make_lba_lm_golden.py -- writes tests/golden/lba_lm_golden.npz: one local bundle adjustment run by the
REFERENCE'S OWN Levenberg-Marquardt text (src/mapHandler.cpp:1334-1812 compiled where it lies: oracle/ref_wrap_lba_lm.cpp via
oracle.ref_lba_lm) on a seeded synthetic local map -- the inputs and, per solve, lambda and err; the final state; the loop's
last values.  Run in the container that holds /root/reference (python tests/golden/make_lba_lm_golden.py); the GPU box only
reads the fixture.  What the fixture pins for plslam_amd/host/lba_rows.hpp LbaPlanSolver::optimize: the first step applied
unconditionally, err / (Npt_obs + Nls_obs) in the first pass and err / (Npt + Nls) in the iterations, lambda GROWING on success,
the stop tests, the pose update through expmap / logmap."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from plslam_amd import synth  # noqa: E402

CFG = dict(homog_th=1e-7, lambda_lm=0.00001, lambda_k=10.0, max_iters=15, min_err_change=1e-7, min_err=1e-7)   # config/config/config.yaml:28-30, 101-106


def problem(n_kf_map, n_pt, n_ls, obs, seed, sigma_lm, sigma_pose):
    """Key frame 0 of the map is never optimised (src/mapHandler.cpp:1231); the others are the local ones, kf_loc = index - 1."""
    lm = synth.local_map(n_kf=n_kf_map, n_pt=n_pt, n_ls=n_ls, obs_per_lm=obs, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    x = np.stack([O.logmap_se3(T) for T in lm["T_kf_w"].reshape(-1, 4, 4)])
    x[1:] += sigma_pose * rng.standard_normal((n_kf_map - 1, 6)) * np.array([1, 1, 1, 0.2, 0.2, 0.2])
    T_map = np.stack([O.expmap_se3(v) for v in x]).reshape(-1, 16)        # the stored T_kf_w of every key frame
    Xw = lm["Xw"] + sigma_lm * rng.standard_normal(lm["Xw"].shape)
    Lw = lm["Lw"] + sigma_lm * rng.standard_normal(lm["Lw"].shape)
    return dict(n_kf_map=n_kf_map, nkf=n_kf_map - 1, T_map=T_map, x_kf=x[1:].reshape(-1), Xw=Xw, Lw=Lw,
                pt_lm=lm["pt_lm"], pt_kf_map=lm["pt_kf"], pt_kf_loc=lm["pt_kf"] - 1, pt_uv=lm["obs_uv"],
                ls_lm=lm["ls_lm"], ls_kf_map=lm["ls_kf"], ls_kf_loc=lm["ls_kf"] - 1, ls_l=lm["l_obs"])


def run_ref(p, cfg=CFG):
    cam = O.make_cam(synth.EUROC["fx"], synth.EUROC["fy"], synth.EUROC["cx"], synth.EUROC["cy"])
    return O.ref_lba_lm(cam, cfg["homog_th"], cfg["lambda_lm"], cfg["lambda_k"], cfg["max_iters"], cfg["min_err_change"], cfg["min_err"],
                        p["nkf"], p["T_map"], p["x_kf"], p["Xw"], p["Lw"], p["pt_lm"], p["pt_kf_map"], p["pt_kf_loc"], p["pt_uv"],
                        p["ls_lm"], p["ls_kf_map"], p["ls_kf_loc"], p["ls_l"])


CASES = {      # name -> problem arguments
    "main": dict(n_kf_map=7, n_pt=400, n_ls=120, obs=4, seed=31, sigma_lm=0.05, sigma_pose=0.01),      # 6 optimised key frames
    # rough starts (found by a seed search): the second / third iteration's err is ABOVE the one before -- step not applied,
    # lambda /= lambda_k, and the iteration after recomputes the same err and stops at the first test
    "reject": dict(n_kf_map=5, n_pt=120, n_ls=40, obs=3, seed=61, sigma_lm=0.8, sigma_pose=0.3),
    "reject_later": dict(n_kf_map=5, n_pt=120, n_ls=40, obs=3, seed=74, sigma_lm=1.5, sigma_pose=0.15),
    "points_only": dict(n_kf_map=4, n_pt=90, n_ls=0, obs=3, seed=5, sigma_lm=0.05, sigma_pose=0.01),
}


def main():
    out = {}
    for name, args in CASES.items():
        p = problem(**args)
        r = run_ref(p)
        if r is None:
            raise SystemExit("oracle/_ref/libplslam_ref.so (ref_lba_lm) is not available: build it where /root/reference is")
        print(f"{name}: N = {r['X'].size}, solves {len(r['lam'])}, iters {r['iters']}, err {r['err']}, lambda {r['lam']}, "
              f"last err {r['err_last']:.6g} prev {r['err_prev']:.6g}")
        for k, v in p.items():
            out[f"{name}_{k}"] = np.asarray(v)
        for k, v in r.items():
            out[f"{name}_ref_{k}"] = np.asarray(v)
    out["cfg"] = np.array([CFG[k] for k in ("homog_th", "lambda_lm", "lambda_k", "max_iters", "min_err_change", "min_err")], np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lba_lm_golden.npz"), **out)


if __name__ == "__main__":
    main()
