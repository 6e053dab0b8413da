"""Generates tests/golden/lba_ref_golden.npz.  Run from the repo root IN THE BUILD CONTAINER (needs /root/reference):
    python tests/golden/make_lba_ref_golden.py

Outputs of the REFERENCE'S OWN SOURCE TEXT: the four observation loops of MapHandler::levMarquardtOptimizationLBA
(src/mapHandler.cpp:1358-1431, :1436-1540 first pass; :1587-1666, :1668-1772 iteration pass), cut out of the file where
it lies and compiled textually by oracle/Makefile (oracle/ref_extract_lba.py, oracle/ref_wrap_lba.cpp), run on a seeded
local map: dense H, g and err after both loops of a pass.  In the iteration pass the optimised key frames' poses differ
from the stored ones (T_slot != T_map[1:]), so the fixture also pins WHICH pose each loop reads (points: expmap(X);
lines: the stored pose) together with the stride-3 / 1e-7 quirks.  The script refuses to write unless the C oracle
agrees to 1e-11.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from plslam_amd import synth  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lba_ref_golden.npz")
N_KF, NPT, NLS, TH = 5, 40, 14, 1e-7


def main():
    lm = synth.local_map(n_kf=N_KF, n_pt=NPT, n_ls=NLS, obs_per_lm=3, seed=91, noise_px=2.0)
    cam = O.make_cam(**synth.EUROC)
    nkf = N_KF - 1
    kp, kl = (lm["pt_kf"] - 1).astype(np.int32), (lm["ls_kf"] - 1).astype(np.int32)
    T_map = np.asarray(lm["T_kf_w"], np.float64).reshape(N_KF, 16)
    r = np.random.Generator(np.random.PCG64(4711))
    T_slot = np.stack([(synth.se3_exp(r.normal(0, 0.02, 6)) @ T_map[k + 1].reshape(4, 4)).reshape(16) for k in range(nkf)])
    args = (cam, TH, nkf, T_map, T_slot, lm["Xw"], lm["Lw"], lm["pt_lm"], lm["pt_kf"], kp, lm["obs_uv"], lm["ls_lm"],
            lm["ls_kf"], kl, lm["l_obs"])
    first = O.ref_lba_accumulate(False, *args)
    if first is None:
        raise SystemExit("oracle/_ref lacks ref_lba_accumulate: run `make -C oracle ref` with /root/reference present")
    it = O.ref_lba_accumulate(True, *args)

    def oracle_Hg(T, slot_p, slot_l, compat):
        rp = O.lba_point_rows(cam, TH, T, lm["Xw"], lm["obs_uv"], lm["pt_lm"], slot_p)
        rl = O.lba_line_rows(cam, TH, T, lm["Lw"], lm["l_obs"], lm["ls_lm"], slot_l, compat_iter_pass=compat)
        H, g, e1 = O.lba_accumulate("points", nkf, NPT, NLS, lm["pt_lm"], kp, *rp)
        H, g, e2 = O.lba_accumulate("lines", nkf, NPT, NLS, lm["ls_lm"], kl, *rl, H=H, g=g)
        return H, g, e1 + e2

    T_all = np.concatenate([T_map, T_slot])
    slot_p = np.where(kp >= 0, N_KF + kp, lm["pt_kf"]).astype(np.int32)
    for ref, (H, g, e) in ((first, oracle_Hg(T_map, lm["pt_kf"], lm["ls_kf"], False)),
                           (it, oracle_Hg(T_all, slot_p, lm["ls_kf"], True))):
        assert np.allclose(ref[0], H, rtol=1e-11, atol=1e-11 * np.abs(H).max()), "oracle != reference source text (H)"
        assert np.allclose(ref[1], g, rtol=1e-11, atol=1e-11 * np.abs(g).max()), "oracle != reference source text (g)"
        assert np.isclose(ref[2], e, rtol=1e-12)
    np.savez_compressed(OUT, T_map=T_map, T_slot=T_slot, Xw=lm["Xw"], Lw=lm["Lw"], obs_uv=lm["obs_uv"], l_obs=lm["l_obs"],
                        pt_lm=lm["pt_lm"], pt_kf=lm["pt_kf"], ls_lm=lm["ls_lm"], ls_kf=lm["ls_kf"],
                        dims=np.array([N_KF, nkf, NPT, NLS], np.int32), th=np.array([TH]),
                        first_H=first[0], first_g=first[1], first_err=np.array([first[2]]),
                        iter_H=it[0], iter_g=it[1], iter_err=np.array([it[2]]))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
