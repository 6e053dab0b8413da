#!/usr/bin/env python3
"""Golden vectors of the stvo-pl helpers the LBA rows and the drivers call and that are [RECALL] in the oracle
(DESIGN.md section 3): inverse_se3, expmap_se3, logmap_se3 (auxiliar.h), PinholeStereoCamera::projection, robustWeightCauchy.
Generated from the oracle's restatement (oracle/plslam_oracle.c:283-392); tools/pin_stvo replays them through a stvo-pl
checkout's own functions the day one exists.   usage: python tests/golden/make_se3_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402


def main():
    r = np.random.Generator(np.random.PCG64(20260924))
    # twists [t, w]: zero, below / at / above the small-angle switch (theta < 1e-6), ordinary, near pi
    xs = [np.zeros(6), np.r_[1.0, -2.0, 0.5, 0, 0, 0], np.r_[0.1, 0.2, 0.3, 5e-7, 0, 0], np.r_[0.1, 0.2, 0.3, 1e-6, 0, 0],
          np.r_[0.1, 0.2, 0.3, 0, 2e-6, 0], np.r_[3.0, -1.0, 2.0, 0.3, -0.2, 0.1], np.r_[0, 0, 0, 0, 0, 3.1],
          np.r_[-5.0, 4.0, 9.0, 1.8, 1.8, -1.8]]
    xs += [np.r_[r.normal(0, 2.0, 3), r.normal(0, 0.7, 3)] for _ in range(24)]
    xs = np.array(xs, dtype=np.float64)
    Ts = np.array([O.expmap_se3(x) for x in xs])
    out = {"twists": xs, "expmap": Ts, "inverse": np.array([O.inverse_se3(T) for T in Ts]),
           "logmap": np.array([O.logmap_se3(T) for T in Ts])}
    # projection u = cx + fx X / Z (the formula the oracle's rows use, oracle/plslam_oracle.c:385-390), incl. points behind the
    # camera, on the optical axis and with Z = 0 (inf / nan, as the C++ double division gives)
    cam = np.array([458.654, 457.296, 367.215, 248.375])            # fx, fy, cx, cy (EuRoC)
    P = np.concatenate([r.normal(0, 3.0, (40, 3)) + np.r_[0, 0, 6.0], np.array([[0, 0, 1.0], [1, 1, -2.0], [1.0, -1.0, 0.0], [0, 0, 0.0]])])
    with np.errstate(divide="ignore", invalid="ignore"):
        uv = np.stack([cam[2] + cam[0] * P[:, 0] / P[:, 2], cam[3] + cam[1] * P[:, 1] / P[:, 2]], axis=1)
    out.update(cam=cam, points=P, projection=uv)
    # robustWeightCauchy(r) = 1 / (1 + r^2)   (mapHandler.cpp:1407 calls it on the residual norm)
    rs = np.concatenate([np.array([0.0, 1e-300, 1.0, 1e8, 1e200, np.inf]), np.abs(r.normal(0, 3.0, 26))])
    with np.errstate(over="ignore"):
        out.update(cauchy_r=rs, cauchy_w=1.0 / (1.0 + rs * rs))
    np.savez(os.path.join(ROOT, "tests", "golden", "se3_helpers_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
