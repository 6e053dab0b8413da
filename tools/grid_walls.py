import sys, os, json
R = os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, R)
import numpy as np, torch
import plslam_amd, bench_rows
from oracle import oracle as O
ctx = plslam_amd.Context(0)
dev = torch.device("cuda", 0)
out = {}
g = bench_rows.grid(ctx, dev, torch, O, torch.cuda.Stream(device=dev))
out["grid"] = {k: v for k, v in g.items() if k != "workload"}
d = bench_rows.drivers(ctx, O, dev)
out["drivers"] = {k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if "us" in a}) for k, v in d.items() if k != "workload"}
print(json.dumps(out, indent=1))
