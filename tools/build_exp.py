#!/usr/bin/env python3
"""Builds experiment variants of libplslam_hip.so into build/exp/<name>.so: every source is compiled to an object once
(build/exp/obj), the file under study is recompiled per variant with extra -D flags, and the objects are linked.
Timing-only builds (results may be wrong with an experiment macro on) for tools/scan_time.py, loaded through
PLSLAM_HIP_LIB_EXPERIMENT; the shipped library is built by plslam_amd/build.py alone.

usage: build_exp.py <source.hip> name1:-DFOO=1 name2:-DFOO=2,-DBAR ...      (name 'base' with no flags = the source as is)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plslam_amd import build as B  # noqa: E402

OBJ = os.path.join(ROOT, "build", "exp", "obj")
OUT = os.path.join(ROOT, "build", "exp")
def compile_obj(src, out, extra=()):
    cmd = [B.hipcc_path()] + B._flags_for(src, B.legacy_scans()) + list(extra) + ["-c", os.path.join(B.CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise SystemExit(r.stdout + r.stderr)


def main():
    study = sys.argv[1]
    os.makedirs(OBJ, exist_ok=True)
    deps = list(B.HEADERS)
    objs = []
    for s in B.sources():
        if s == study:
            continue
        o = os.path.join(OBJ, s + (".legacy" if B.legacy_scans() else "") + ".o")
        srcs = [os.path.join(B.CSRC, s)] + deps
        if not os.path.exists(o) or any(os.path.getmtime(x) > os.path.getmtime(o) for x in srcs):
            compile_obj(s, o)
        objs.append(o)
    for spec in sys.argv[2:]:
        name, _, flags = spec.partition(":")
        o = os.path.join(OBJ, f"{study}.{name}.o")
        compile_obj(study, o, [f for f in flags.split(",") if f])
        so = os.path.join(OUT, name + ".so")
        r = subprocess.run([B.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + [o, "-o", so, "-ldl"],
                           capture_output=True, text=True)
        if r.returncode:
            raise SystemExit(r.stdout + r.stderr)
        print(so)


if __name__ == "__main__":
    main()
