"""Generates tests/golden/lba_golden.npz: small inputs + the oracle's outputs for the LBA rows, the
gates, the block accumulation and the map<->keyframe drivers.  Run from the repo root:
       python tests/golden/make_lba_golden.py
The reference has no fixtures for this path; these pin OUR restatement (oracle/plslam_oracle.c, which
restates src/mapHandler.cpp:1358-1540, :601-613, :716-729, :532-752) so that the GPU tests can also
run against committed numbers rather than only against a freshly built oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from plslam_amd import synth  # noqa: E402
from test_map2kf import scene  # noqa: E402


def main():
    out = {}
    cam = O.make_cam(**synth.EUROC)
    lm = synth.local_map(n_kf=5, n_pt=120, n_ls=40, obs_per_lm=3, seed=21)
    for k, v in lm.items():
        out["map/" + k] = v
    for name, rows in (("pt", O.lba_point_rows(cam, 1e-7, lm["T_kf_w"], lm["Xw"], lm["obs_uv"], lm["pt_lm"], lm["pt_kf"])),
                       ("ls", O.lba_line_rows(cam, 1e-7, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"], lm["ls_kf"])),
                       ("ls_compat", O.lba_line_rows(cam, 1e-3, lm["T_kf_w"], lm["Lw"], lm["l_obs"], lm["ls_lm"],
                                                     lm["ls_kf"], compat_iter_pass=True))):
        for nm, a in zip(("J_pose", "J_lm", "r", "w"), rows):
            out[f"rows/{name}/{nm}"] = a
    nkf = 4
    pk, lk = lm["pt_kf"] - 1, lm["ls_kf"] - 1
    rp = [out[f"rows/pt/{n}"] for n in ("J_pose", "J_lm", "r", "w")]
    rl = [out[f"rows/ls/{n}"] for n in ("J_pose", "J_lm", "r", "w")]
    H, g, e1 = O.lba_accumulate("points", nkf, 120, 40, lm["pt_lm"], pk, *rp)
    H, g, e2 = O.lba_accumulate("lines", nkf, 120, 40, lm["ls_lm"], lk, *rl, H=H, g=g)
    out["acc/H"], out["acc/g"], out["acc/err"] = H, g, np.array([e1 + e2])
    for kind, lines in (("points", False), ("lines", True)):
        s = scene(400, 120, lines=lines, seed=5)
        for k, v in s.items():
            out[f"drv/{kind}/{k}"] = np.asarray(v)
        m, n = O.map2kf_match(kind, cam, s["Twf"], s["LM"], s["med"], s["cand"], s["kf_desc"], s["kf_feat"], s["kf_idx"],
                              0.9, True, 1.0, 10)
        out[f"drv/{kind}/map_to_kf"], out[f"drv/{kind}/n"] = m, np.array([n])
        vis = (O.map_line_visible if lines else O.map_point_visible)(cam, s["Twf"], s["LM"])
        out[f"drv/{kind}/visible"] = vis
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lba_golden.npz"), **out)
    print("wrote lba_golden.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
