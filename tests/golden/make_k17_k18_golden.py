"""Generates tests/golden/lbd_float_golden.npz and pose_gn_golden.npz (run from the repo root:
python tests/golden/make_k17_k18_golden.py, in the build container: needs oracle/_ref built from /root/reference).
The LBD vectors are OUTPUTS OF THE REFERENCE ITSELF: BinaryDescriptor::computeLBD (binary_descriptor_custom.cpp:1026-1372)
compiled from where it lies (oracle/ref_wrap_lbd.cpp) and run on the seeded gradient images; the script refuses to
write them unless the C oracle reproduces them bit for bit and the independent float64 numpy derivation of
tests/test_lbd_float.py agrees to 2e-5.  The pose-GN vectors are likewise the reference's own: the point and line loops
of computeRelativePoseGN (src/mapHandler.cpp:3331-3426) compiled textually (oracle/ref_wrap_lba.cpp); written only if the
C oracle agrees to 1e-11 and the twin loops of computeRelativePoseRobustGN give the same numbers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from plslam_amd import synth  # noqa: E402
from test_lbd_float import _np_one_line  # noqa: E402
from test_pose_gn import scene  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    r = np.random.Generator(np.random.PCG64(77))
    dx, dy = synth.gradient_images(r, 128, 96)
    lines = synth.lbd_lines(r, 24, 128, 96, min_len=8, max_len=70, dtype=O.LBD_LINE_DTYPE)
    out = O.ref_lbd_compute(dx, dy, lines)
    if out is None:
        raise SystemExit("oracle/_ref lacks ref_lbd_compute: run `make -C oracle ref` with /root/reference present")
    assert np.array_equal(out.view(np.uint32), O.lbd_compute(dx, dy, lines).view(np.uint32)), "oracle != reference"
    inner = (np.minimum(lines["sx"], lines["ex"]) > 35) & (np.maximum(lines["sx"], lines["ex"]) < 93) & \
            (np.minimum(lines["sy"], lines["ey"]) > 35) & (np.maximum(lines["sy"], lines["ey"]) < 61)
    for i in np.nonzero(inner)[0]:
        np.testing.assert_allclose(out[i], _np_one_line(dx, dy, lines[i]), rtol=0, atol=5e-5)
    np.savez_compressed(os.path.join(OUT, "lbd_float_golden.npz"), dx=dx, dy=dy, lines=lines, lbd=out,
                        codes=O.lbd_binarise(out))
    s = scene(60, 20, seed=31)
    cam = O.make_cam(**synth.EUROC)
    a = (cam, 1e-7, s["T"], s["P"], s["pl_obs"], s["pt_in"], s["sPeP"], s["le_obs"], s["ls_in"])
    ref = O.ref_pose_gn_accumulate(False, *a)          # the reference's own loops, compiled textually (:3331-3426)
    if ref is None:
        raise SystemExit("oracle/_ref lacks ref_pose_gn_accumulate: run `make -C oracle ref` with /root/reference present")
    H, g, e, n = ref
    Ho, go, eo, no = O.pose_gn_accumulate(*a)
    assert no == n and np.allclose(Ho, H, rtol=1e-11, atol=1e-11 * np.abs(H).max()) and np.allclose(go, g, rtol=1e-11) \
        and np.isclose(eo, e, rtol=1e-12), "oracle != reference source text"
    rr = O.ref_pose_gn_accumulate(True, *a)            # computeRelativePoseRobustGN's twin loops (:3595-3689)
    assert rr[3] == n and np.allclose(rr[0], H, rtol=1e-13, atol=0) and np.allclose(rr[1], g, rtol=1e-13, atol=0)
    np.savez_compressed(os.path.join(OUT, "pose_gn_golden.npz"), H=H, g=g, e=e, n=np.array(n), **s)
    print("wrote lbd_float_golden.npz, pose_gn_golden.npz")


if __name__ == "__main__":
    main()
