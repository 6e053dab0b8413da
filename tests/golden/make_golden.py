"""Generates tests/golden/*.npz.  Run from the repo root IN THE BUILD CONTAINER:
       python tests/golden/make_golden.py

The reference holds no tests or golden vectors for this path, and its matcher
arithmetic lives in OpenCV / stvo-pl which are not under /root/reference.  What CAN be taken from
the reference itself is the 256-bit Hamming distance: the `dist_ref_*` arrays below are produced by
the reference's own code compiled from where it lies (oracle/_ref: bitops_custom.hpp:83-96 and
DBoW2 FORB.cpp:78-101).  The kNN-2 / match tables are produced by the numpy mirror
(oracle.oracle.np_*: full distance matrix + stable sort), i.e. by a formulation independent of both
the C oracle and the HIP kernels, from those reference-pinned distances.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from plslam_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def ref_dist_matrix(fn, q, t):
    D = np.empty((q.shape[0], t.shape[0]), np.int32)
    for i in range(q.shape[0]):
        for j in range(t.shape[0]):
            D[i, j] = fn(q[i], t[j])
    return D


def main():
    ref = O.ref_lib()
    assert ref is not None, "oracle/_ref must be built (needs /root/reference)"
    bit = lambda a, b: ref.ref_ld_match(a.ctypes.data, b.ctypes.data, 32)
    forb = lambda a, b: ref.ref_forb_distance(a.ctypes.data, b.ctypes.data)

    cases = {}
    rng = np.random.Generator(np.random.PCG64(424242))
    # 1. random with planted matches
    q = synth.random_desc(rng, 96)
    t, _, _ = synth.noisy_copy(rng, q)
    t = np.ascontiguousarray(t[:80])
    cases["planted"] = (q, t)
    # 2. tie stress (duplicates, equal distances)
    cases["ties"] = (synth.tie_stress_desc(rng, 70), synth.tie_stress_desc(rng, 90))
    # 3. known-answer rows: all-zero, all-one, single bits, duplicates
    kat = np.zeros((12, 32), np.uint8)
    kat[1] = 0xFF
    kat[2, 0] = 1
    kat[3, 31] = 0x80
    kat[4] = kat[2]
    kat[5, :16] = 0xFF
    kat[6, 16:] = 0xFF
    kat[7] = 0xAA
    kat[8] = 0x55
    kat[9] = kat[1]
    kat[10, 5] = 0x18
    kat[11, 5] = 0x18
    cases["kat"] = (kat, kat[::-1].copy())
    # 4. ratio boundaries: query = zeros, train rows with exactly d bits set
    def row_with_bits(d):
        r = np.zeros(256, np.uint8)
        r[:d] = 1
        return np.packbits(r)
    qb = np.zeros((1, 32), np.uint8)
    for name, (d0, d1) in {"9_10": (9, 10), "3_4": (3, 4), "75_100": (75, 100), "0_0": (0, 0),
                           "5_5": (5, 5), "89_99": (89, 99), "90_100": (90, 100), "6_10": (6, 10),
                           "59_99": (59, 99), "60_100": (60, 100)}.items():
        cases["ratio_" + name] = (qb, np.stack([row_with_bits(d1), row_with_bits(d0), row_with_bits(200)]))

    out = {}
    for name, (q, t) in cases.items():
        D1 = ref_dist_matrix(bit, q, t)
        D2 = ref_dist_matrix(forb, q, t)
        assert np.array_equal(D1, D2), name
        assert np.array_equal(D1, O.np_dist_matrix(q, t)), name
        idx, dist = O.np_knn2(q, t)
        out[f"{name}/q"], out[f"{name}/t"] = q, t
        out[f"{name}/dist_ref_bitops"] = D1
        out[f"{name}/knn_idx"], out[f"{name}/knn_dist"] = idx, dist
        for nnr in (0.6, 0.75, 0.9):
            for mutual in (0, 1):
                m, n = O.np_match(q, t, nnr, bool(mutual))
                out[f"{name}/m12_nnr{nnr}_mut{mutual}"] = m
    np.savez_compressed(os.path.join(OUT, "match_golden.npz"), **out)
    print("wrote match_golden.npz with", len(out), "arrays,",
          os.path.getsize(os.path.join(OUT, "match_golden.npz")), "bytes")


if __name__ == "__main__":
    main()
