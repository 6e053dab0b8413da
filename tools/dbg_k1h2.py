#!/usr/bin/env python3
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import plslam_amd
from plslam_amd import frontend, synth
from oracle import oracle as O
t00 = time.time()
ctx = plslam_amd.Context(0)
ctx.set_option("scan_variant", plslam_amd.SCAN_MFMA)
ctx.set_option("mfma_form", int(sys.argv[1]) if len(sys.argv) > 1 else 4)
s = synth.stereo_stream(3, 320, 70, seed=5)
bm = frontend.StereoBatchMatcher(ctx, s, nnr_p=0.75, nnr_l=0.9, mutual=True)
info = bm.plan.info()
print("info", info, "t", time.time() - t00, flush=True)
for it in range(2):
    tab = bm.run(); torch.cuda.synchronize()
    tab = tab.cpu().numpy(); cnt = bm.counts.cpu().numpy()
    sl = frontend.table_slices(320, 70)
    for i in range(3):
        for k, (name, d1, d2) in enumerate(frontend.pair_problems(s["orb_l"], s["orb_r"], s["lbd_l"], s["lbd_r"], i)):
            em, en = O.match(d1, d2, 0.75 if name.startswith("orb") else 0.9, True)
            same = np.array_equal(tab[i, sl[name]], em)
            if not same or cnt[i, k] != en:
                bad = np.nonzero(tab[i, sl[name]] != em)[0]
                print("MISMATCH", it, i, name, "rows", bad[:10], "got", tab[i, sl[name]][bad[:10]], "exp", em[bad[:10]], "cnt", cnt[i, k], en, flush=True)
print("compared t", time.time() - t00, flush=True)
bm.plan.set_profiling(True)
bm.run()
print(bm.plan.elapsed(), "t", time.time() - t00, flush=True)
bm.close()
