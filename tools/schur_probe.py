import sys, json, time
sys.path.insert(0, '.')
import numpy as np
import plslam_amd
from oracle import oracle as O
import bench_rows as R
ctx = plslam_amd.Context(0)
t0=time.perf_counter()
rec = R.lba_iterate(ctx, O)
print(json.dumps({k: rec[k] for k in ("us_median", "err_only_us_median", "state_resident_us_median", "state_in_page_locked_images_us_median", "schur_step")}, indent=1))
print("record took", time.perf_counter()-t0)
