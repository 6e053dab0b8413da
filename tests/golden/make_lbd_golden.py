"""Generates tests/golden/lbd_golden.npz: 77 LBD float rows (incl. ties, NaN, inf, -0.0) and the
oracle's 32-byte binary rows.  Run from the repo root:
       python tests/golden/make_lbd_golden.py
The outputs are written by the numpy mirror (oracle.np_lbd_binarise) and must equal the C oracle
(plo_lbd_binarise, restating 3rdparty/line_descriptor/src/binary_descriptor_custom.cpp:401-412,
:653-668) -- the script asserts that before saving.  The band-pair table both use is pinned to the
reference source text by tests/test_oracle_pin.py::test_lbd_pair_table_pinned_to_reference_source."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from plslam_amd import synth  # noqa: E402


def main():
    rng = np.random.Generator(np.random.PCG64(20260924))
    f = np.concatenate([synth.lbd_float(rng, 40), synth.lbd_float(rng, 33, levels=6)])
    f[3, 5] = np.nan
    f[3, 13] = np.nan
    f[4, :8] = np.inf
    f[5, 8:16] = -0.0
    f[5, 0:8] = 0.0
    f[6] = 0.25                                  # all equal -> all-zero code
    f[7] = np.arange(72, dtype=np.float32)[::-1] # strictly decreasing -> all-ones code
    d = O.np_lbd_binarise(f)
    assert (d == O.lbd_binarise(f)).all()
    assert (d[6] == 0).all() and (d[7] == 255).all()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lbd_golden.npz"), lbd_f32=f, desc_u8=d)
    print("wrote lbd_golden.npz", f.shape, d.shape)


if __name__ == "__main__":
    main()
