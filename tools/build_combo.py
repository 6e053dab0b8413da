#!/usr/bin/env python3
"""Experiment build of libplslam_hip.so with extra -D flags on SEVERAL sources (tools/build_exp.py studies one file):
usage: build_combo.py <name> <source.hip>:-DFOO=1,-DBAR <other.hip>:-DBAZ=2 ...   -> build/exp/<name>.so
(timing builds for tools/scan_time.py / tools/ab_lib.sh through PLSLAM_HIP_LIB_EXPERIMENT; objects are cached by flag set)"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plslam_amd import build as B  # noqa: E402

OBJ = os.path.join(ROOT, "build", "exp", "obj")
OUT = os.path.join(ROOT, "build", "exp")


def main():
    name = sys.argv[1]
    study = {}
    for spec in sys.argv[2:]:
        src, _, flags = spec.partition(":")
        study[src] = [f for f in flags.split(",") if f]
    os.makedirs(OBJ, exist_ok=True)
    deps = [os.path.join(B.CSRC, h) if not os.path.isabs(h) else h for h in B.HEADERS]
    objs = []
    for s in B.sources():
        extra = study.get(s, [])
        tag = hashlib.sha1(" ".join(extra).encode()).hexdigest()[:8] if extra else "plain"
        o = os.path.join(OBJ, f"{s}.{tag}.o")
        srcs = [os.path.join(B.CSRC, s)] + deps
        if not os.path.exists(o) or any(os.path.getmtime(x) > os.path.getmtime(o) for x in srcs if os.path.exists(x)):
            cmd = [B.hipcc_path()] + B._flags_for(s, B.legacy_scans()) + extra + ["-c", os.path.join(B.CSRC, s), "-o", o]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise SystemExit(r.stdout + r.stderr)
        objs.append(o)
    so = os.path.join(OUT, name + ".so")
    r = subprocess.run([B.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", so, "-ldl"], capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stdout + r.stderr)
    print(so)


if __name__ == "__main__":
    main()
