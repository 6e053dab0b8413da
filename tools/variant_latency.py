import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np
import plslam_amd
from plslam_amd import synth
r = np.random.Generator(np.random.PCG64(1))
ctx = plslam_amd.Context(0)
for (n1, n2) in ((10000, 1500), (2000, 200), (1500, 1500), (4000, 4000)):
    a = synth.random_desc(r, n1); b = synth.random_desc(r, n2)
    for name, v in (("auto", 0), ("wpq", 2), ("sym", 3), ("mfma", 4)):
        ctx.set_option("scan_variant", v)
        for _ in range(5): ctx.match(a, b, 0.75, True)
        ts = []
        for _ in range(30):
            t0 = time.perf_counter(); ctx.match(a, b, 0.75, True); ts.append(time.perf_counter() - t0)
        print(n1, n2, name, "median us %.1f" % (1e6 * float(np.median(ts))))
