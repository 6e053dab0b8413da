"""Run-to-run determinism of the batched stereo matcher under load (gpurun / GPU box only).

Two plans over the same descriptors run back to back on two streams (the bench's overlapped mode: several
workgroups share every CU), repeatedly.  After each round the first plan's intermediate key table and column
partials (plslam_match_plan_dump) and its match table must equal the first round's, word for word.  The key words
are the sensitive instrument: a wrong distance changes a key always, a match-table entry only when it flips a
ratio test (~10^-4 of the cases at nnr 0.9).

    python tools/determinism_check.py [--n-orb 256 --n-lbd 256 --pairs 128 --rounds 12 --nnr 0.9 --directed]

Exit code 1 and a per-round report when anything differs.  History: this is the instrument that bisected the K1e
symmetric scan's exec-masked column store ("K1e determinism").
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import plslam_amd
from plslam_amd import frontend, synth


def check(ctx, n_orb, n_lbd, pairs, rounds, nnr, mutual=True, device=None):
    """-> dict(key_words, key_diffs, partial_diffs, table_diffs) summed over rounds 2..rounds vs round 1."""
    device = device or torch.device("cuda", 0)
    stream = synth.stereo_stream(pairs, n_orb, n_lbd, seed=synth.SEED0, first_pair=0)
    # the merged column keys are part of the dumped key table: they exist in memory only with the separate kernels behind the
    # scan (the fused stage keeps them in LDS)
    prev = ctx.get_option("post_fuse")
    ctx.set_option("post_fuse", 1)
    try:
        bm = frontend.StereoBatchMatcher(ctx, stream, nnr_p=nnr, nnr_l=nnr, mutual=mutual, device=device, n_buffers=2)
    finally:
        ctx.set_option("post_fuse", prev)
    ref = None
    out = {"key_words": 0, "key_diffs": 0, "partial_diffs": 0, "table_diffs": 0, "rounds": rounds}
    for _ in range(rounds):
        for k in range(4):
            bm.run_overlapped(k)
        bm.synchronize_all()
        keys, part = bm.plans[0].dump()
        table = bm.tables[0].cpu().numpy()
        if ref is None:
            ref = (keys.copy(), part.copy(), table.copy())
            out["key_words"] = int(keys.size)
            continue
        out["key_diffs"] += int((keys != ref[0]).sum())
        out["partial_diffs"] += int((part != ref[1]).sum())
        out["table_diffs"] += int((table != ref[2]).sum())
    bm.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-orb", type=int, default=256)
    ap.add_argument("--n-lbd", type=int, default=256)
    ap.add_argument("--pairs", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=12)
    ap.add_argument("--nnr", type=float, default=0.9)
    ap.add_argument("--directed", action="store_true", help="non-mutual problems (the directed scan)")
    ap.add_argument("--scan-variant", type=int, default=None)
    a = ap.parse_args()
    ctx = plslam_amd.Context(0)
    if a.scan_variant is not None:
        ctx.set_option("scan_variant", a.scan_variant)
    r = check(ctx, a.n_orb, a.n_lbd, a.pairs, a.rounds, a.nnr, mutual=not a.directed)
    r["config"] = vars(a)
    print(json.dumps(r))
    return 1 if (r["key_diffs"] or r["partial_diffs"] or r["table_diffs"]) else 0


if __name__ == "__main__":
    sys.exit(main())
