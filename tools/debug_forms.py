#!/usr/bin/env python3
"""Diagnostic: K1e vs K1f key tables of one batched plan, decoded per problem and compared with the oracle's knn2.
usage: debug_forms.py n_orb n_lbd pairs [mutual=1] [tie=0]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import plslam_amd
from oracle import oracle as O
from plslam_amd import frontend, synth

n_orb, n_lbd, pairs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mutual = bool(int(sys.argv[4])) if len(sys.argv) > 4 else True
tie = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
s = synth.stereo_stream(pairs, n_orb, n_lbd, seed=4242 + n_orb, tie_stress=tie)
ctx = plslam_amd.Context(0)
ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
got = {}
for form in (1, 2):
    ctx.set_option("mfma_form", form)
    bm = frontend.StereoBatchMatcher(ctx, s, nnr_p=0.85, nnr_l=0.9, mutual=mutual)
    bm.run()
    torch.cuda.synchronize()
    k, p = bm.plan.dump()
    got[form] = (k.copy(), p.copy(), bm.table.cpu().numpy().copy())
    bm.close()
print("key words", got[1][0].size, "diff", int((got[1][0] != got[2][0]).sum()), "partials", got[1][1].size, "diff",
      int((got[1][1] != got[2][1]).sum()), "tables diff", int((got[1][2] != got[2][2]).sum()))
off = 0
shown = 0
for i in range(pairs):
    for name, d1, d2 in frontend.pair_problems(s["orb_l"], s["orb_r"], s["lbd_l"], s["lbd_r"], i):
        n1, n2 = len(d1), len(d2)
        segs = [("k12", d1, d2, n1)] + ([("k21", d2, d1, n2)] if mutual else [])
        for tag, q, t, n in segs:
            eidx, edist = O.knn2(q, t)
            exp = np.where(eidx >= 0, (edist.astype(np.uint32) << 23) | eidx.astype(np.uint32), 0xFFFFFFFF).astype(np.uint32)
            for form in (1, 2):
                k = got[form][0][off:off + 2 * n].reshape(n, 2)
                bad = np.argwhere(k != exp)
                if len(bad) and shown < 12:
                    shown += 1
                    r, c = bad[0]
                    print(f"pair {i} {name} {tag} form {form}: {len(bad)} wrong words; first row {r} col {c}: got d={k[r, c] >> 23} "
                          f"j={k[r, c] & 0x7FFFFF}  expect d={exp[r, c] >> 23} j={exp[r, c] & 0x7FFFFF}; best got j={k[r, 0] & 0x7FFFFF} d={k[r, 0] >> 23}")
            off += 2 * n
print("done")
