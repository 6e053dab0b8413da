"""Generates tests/golden/ref_knn_golden.npz.  Run from the repo root IN THE BUILD CONTAINER (needs /root/reference):
    python tests/golden/make_ref_knn_golden.py

Outputs of the REFERENCE ITSELF: cv::line_descriptor::BinaryDescriptorMatcher::knnMatch
(/root/reference/3rdparty/line_descriptor/src/binary_descriptor_matcher.cpp:258-335, the exact multi-index-hashing
search; compiled from where it lies into oracle/_ref by oracle/Makefile, called through oracle/ref_wrap_mih.cpp) on
seeded descriptor sets, k = 2 and k = 3.  It is the only kNN code inside the reference tree; the matcher pl-slam's
hot path calls (cv::BFMatcher through stvo-pl) is not vendored.  An exact search fixes the k nearest DISTANCES; which
of several equally distant rows it names is the engine's own business (MIH: bucket order; observed to vary between
runs), so the fixture stores the reference's indices only for queries whose nearest distances are all distinct.
Observed while building this: for train sets of fewer than ~6 rows the reference engine returns out-of-range
trainIdx values (its `results` array is allocated uninitialised and not fully written) while the distances stay
right -- so no spec below is that small.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from plslam_amd import synth  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_knn_golden.npz")
# (name, nq, nt, kind): plain = i.i.d. rows; planted = t holds noisy copies of q rows; ties = 5-bit-entropy rows
SPECS = [("plain", 96, 128, "plain"), ("planted", 120, 150, "planted"), ("ties", 80, 90, "ties"),
         ("tiny", 5, 9, "plain"), ("wide", 16, 600, "planted")]


def make(name, nq, nt, kind, seed):
    r = np.random.Generator(np.random.PCG64(seed))
    if kind == "ties":
        return synth.tie_stress_desc(r, nq), synth.tie_stress_desc(r, nt)
    q = synth.random_desc(r, nq)
    if kind == "planted":
        t = synth.random_desc(r, nt)
        m = min(nq, nt)
        t[:m] = synth.noisy_copy(r, q[:m])[0]
        return q, t
    return q, synth.random_desc(r, nt)


def main():
    if O.ref_mih_knn(np.zeros((1, 32), np.uint8), np.zeros((2, 32), np.uint8), 2) is None:
        raise SystemExit("oracle/_ref lacks ref_mih_knn: run `make -C oracle ref` with /root/reference present")
    out = {"names": np.array([s[0] for s in SPECS])}
    for k, (name, nq, nt, kind) in enumerate(SPECS):
        q, t = make(name, nq, nt, kind, 77000 + k)
        out[f"{name}/q"], out[f"{name}/t"] = q, t
        D = np.sort(np.bitwise_count(q[:, None, :] ^ t[None, :, :]).sum(-1), axis=1)
        for kk in (2, 3):
            idx, dist = O.ref_mih_knn(q, t, kk)
            # Among equally distant rows the engine's choice is not even repeatable from run to run: keep the indices
            # only where the kk + 1 smallest distances of the query are distinct, -1 elsewhere.
            tied = ~np.all(np.diff(D[:, :kk + 1], axis=1) > 0, axis=1)
            idx = np.where(tied[:, None], -1, idx)
            out[f"{name}/k{kk}_idx"], out[f"{name}/k{kk}_dist"] = idx, dist
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
