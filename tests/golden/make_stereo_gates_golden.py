#!/usr/bin/env python3
"""Writes tests/golden/stereo_gates_golden.npz: inputs and expected outputs of the stereo L<->R gates
(StereoFrame::matchStereoPoints / matchStereoLines, stvo-pl [RECALL]; thresholds = config/config/config_kitti.yaml:25-36) for
a few seeded frames, computed with the CPU restatement (oracle/plslam_oracle.c).  The fixture is what the device is checked
against at run time on the GPU box (no oracle needed there) and what tools/pin_stvo replays through a real stvo-pl build.
   python tests/golden/make_stereo_gates_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from test_stereo_gates import stereo_lines, stereo_points  # noqa: E402

POINT_TH = ((1.0, 1.0), (0.0, 1.0), (2.5, 0.0))                       # (max_dist_epip, min_disp)
LINE_TH = ((1.0, 0.1, 0.75, 0.7), (0.0, 0.1, 0.2, 0.3), (1.0, 1.0, 0.75, 0.7))   # (min_disp, line_horiz_th, stereo_overlap_th, ls_min_disp_ratio)


def main():
    out = {}
    for c, (n_l, n_r) in enumerate(((300, 280), (37, 64), (1, 1))):
        m12, kp_l, kp_r = stereo_points(500 + c, n_l, n_r)
        out[f"p{c}_m12"], out[f"p{c}_kp_l"], out[f"p{c}_kp_r"] = m12, kp_l, kp_r
        for t, (th, md) in enumerate(POINT_TH):
            s12, disp, n = O.stereo_point_gate(m12, kp_l, kp_r, th, md)
            out[f"p{c}_t{t}_stereo"], out[f"p{c}_t{t}_disp"], out[f"p{c}_t{t}_n"] = s12, disp, np.int32(n)
    with np.errstate(all="ignore"):
        for c, (n_l, n_r) in enumerate(((120, 110), (19, 7), (1, 1))):
            m12, seg_l, seg_r = stereo_lines(700 + c, n_l, n_r)
            out[f"l{c}_m12"], out[f"l{c}_seg_l"], out[f"l{c}_seg_r"] = m12, seg_l, seg_r
            for t, (md, hz, ov, ratio) in enumerate(LINE_TH):
                s12, disp, n = O.stereo_line_gate(m12, seg_l, seg_r, md, hz, ov, ratio)
                out[f"l{c}_t{t}_stereo"], out[f"l{c}_t{t}_disp"], out[f"l{c}_t{t}_n"] = s12, disp, np.int32(n)
    out["point_thresholds"] = np.array(POINT_TH, np.float64)
    out["line_thresholds"] = np.array(LINE_TH, np.float64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "stereo_gates_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
