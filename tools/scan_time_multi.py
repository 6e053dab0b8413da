#!/usr/bin/env python3
"""scan_time.py for several builds in ONE process start each (data generated once on disk).  usage:
scan_time_multi.py form[:fuse] lib1.so lib2.so ...   ('shipped' = the in-tree library)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
form, _, fuse = sys.argv[1].partition(":")
for lib in sys.argv[2:]:
    env = dict(os.environ)
    if lib != "shipped":
        env["PLSLAM_HIP_LIB_EXPERIMENT"] = lib
    else:
        env.pop("PLSLAM_HIP_LIB_EXPERIMENT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_time.py"), "4", "4096", "1", form, fuse or "0"],
                       env=env, capture_output=True, text=True)
    print((r.stdout.strip().split("\n") or [""])[-1] or r.stderr[-300:], flush=True)
