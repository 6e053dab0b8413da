#!/bin/bash
# scratch: the native LM timing program under a debugger
cd $GRAFT_REPO_ROOT
python - <<'P'
import sys, os
sys.path.insert(0, "tests")
import numpy as np
import test_gpu_lba_lm as t
from oracle import oracle as O
from plslam_amd import synth
lm = synth.local_map()
n = lm["T_kf_w"].shape[0]
x = np.stack([O.logmap_se3(T) for T in lm["T_kf_w"].reshape(-1, 4, 4)])
g = {"cfg": np.array([1e-7, 1e-5, 10.0, 15, 1e-7, 1e-7]), "c3_nkf": n - 1, "c3_n_kf_map": n,
     "c3_T_map": lm["T_kf_w"], "c3_x_kf": x[1:].reshape(-1), "c3_Xw": lm["Xw"], "c3_Lw": lm["Lw"],
     "c3_pt_lm": lm["pt_lm"], "c3_pt_kf_map": lm["pt_kf"], "c3_pt_kf_loc": lm["pt_kf"] - 1, "c3_pt_uv": lm["obs_uv"],
     "c3_ls_lm": lm["ls_lm"], "c3_ls_kf_map": lm["ls_kf"], "c3_ls_kf_loc": lm["ls_kf"] - 1, "c3_ls_l": lm["l_obs"]}
os.makedirs("/tmp/lmdbg", exist_ok=True)
t._compile("/tmp/lmdbg")
t._write_problem("/tmp/lmdbg/c3.bin", g, "c3")
P
if command -v rocgdb >/dev/null; then rocgdb -batch -ex run -ex bt --args /tmp/lmdbg/test_lm_loop /tmp/lmdbg/c3.bin --time 50 2>&1 | tail -30
else /tmp/lmdbg/test_lm_loop /tmp/lmdbg/c3.bin --time 50; echo rc=$?; fi
