"""Builds plslam_amd/lib/libplslam_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

Every source is compiled to an object of its own (build/obj/<source>.o, rebuilt when the source, a header or the flags
changed), several at a time, and the objects are linked: a change to one kernel costs one compilation, a build from
nothing about 40 s instead of 80."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
CSRC = os.path.join(_HERE, "csrc")
OUT = os.path.join(_HERE, "lib", "libplslam_hip.so")
# the same library WITH the earlier scan generations: never loaded by the product -- the GPU tests' cross-checks run against it
# in a subprocess (tests/test_gpu_match.py::test_earlier_scan_generations_cross_check_in_a_legacy_build)
LEGACY_OUT = os.path.join(_HERE, "lib", "libplslam_hip_legacy.so")
OBJ_DIR = os.path.join(_ROOT, "build", "obj")
# the product: what AUTO can pick (K1i and its merge kernel, K1f for column-split plans, the popcount kernels) and every
# other row of SURVEY section 8
SOURCES = ["hamming.hip", "hamming_mfma_g.hip", "hamming_mfma_h.hip", "hamming_mfma_i.hip", "lba.hip", "lba_assemble.hip",
           "map2kf.hip", "lbd.hip", "median_desc.hip", "match_grid.hip", "stereo_gates.hip", "pose_gn.hip", "lbd_float.hip",
           "capi.hip"]
# earlier generations of the matrix-core scan, reachable only through the context option "mfma_form" (1 = K1e, 3 = K1g,
# 4 = K1h -- whose scan kernel sits behind the same macro in hamming_mfma_h.hip): cross-checks for the tests and A/B baselines
# for the tools.  OPT-IN: build_hip(legacy=True) / PLSLAM_BUILD_LEGACY_SCANS=1 python -m plslam_amd.build -> LEGACY_OUT.
# Without them (the product) the option values are refused with PLSLAM_ENOTSUP and the test cases that use them skip in the
# main test process.  Only the translation units that see the macro are compiled twice (LEGACY_AWARE): the legacy library
# shares the other twelve objects with the product.
LEGACY_SOURCES = ["hamming_mfma.hip", "hamming_mfma_d.hip"]
LEGACY_AWARE = ("capi.hip", "hamming_mfma_h.hip")
HEADERS = [os.path.join(CSRC, h) for h in ("common.hpp", "gfx950_only.hpp", "mfma_h_common.hpp", "lba_rows_dev.hpp",
                                            "stereo_gates_dev.hpp")] + [os.path.join(_ROOT, "include", "plslam_hip.h")]
# -ffp-contract=off: the fp64 row kernels must execute the reference's operation order
# (no FMA contraction) so that thresholded masks reproduce the CPU restatement bit for bit.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wextra", "-Wno-unused-command-line-argument",
         # every translation unit refuses any device target but gfx950 (the LDS-DMA asm of K1h / K1i, hand-counted vmcnt waits)
         "-include", os.path.join(CSRC, "gfx950_only.hpp")]
# K1h / K1i declare M0 clobbered by their LDS-DMA statement; the pragma form of this does not reach the diagnostic, which is
# raised at code generation -- the flag is given to these translation units ONLY
PER_SOURCE_FLAGS = {"hamming_mfma_h.hip": ["-Wno-inline-asm"], "hamming_mfma_i.hip": ["-Wno-inline-asm"]}


def legacy_scans() -> bool:
    return os.environ.get("PLSLAM_BUILD_LEGACY_SCANS", "0") not in ("", "0")


def sources(legacy=None):
    legacy = legacy_scans() if legacy is None else legacy
    return SOURCES + (LEGACY_SOURCES if legacy else [])


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build libplslam_hip.so)")


def _extra_flags():
    # PLSLAM_HIPCC_EXTRA: extra flags for experiment builds (e.g. -DPLSLAM_GRID_TIMING); never set by the product
    return os.environ.get("PLSLAM_HIPCC_EXTRA", "").split()


def _flags_for(src: str, legacy=None):
    legacy = legacy_scans() if legacy is None else legacy
    f = FLAGS + PER_SOURCE_FLAGS.get(src, []) + _extra_flags() + ["-I" + os.path.join(_ROOT, "include")]
    if legacy and (src in LEGACY_AWARE or src in LEGACY_SOURCES):
        f = f + ["-DPLSLAM_BUILD_LEGACY_SCANS=1"]
    return f


def _obj_for(src: str, legacy=None) -> str:
    tag = hashlib.sha256(" ".join(_flags_for(src, legacy)).encode()).hexdigest()[:10]
    return os.path.join(OBJ_DIR, f"{src}.{tag}.o")


def _stale(path: str, deps) -> bool:
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps)


def _stamp(legacy=None) -> str:
    """What the in-tree library must have been built from: the source list and every source's flags."""
    return hashlib.sha256("|".join(s + " " + " ".join(_flags_for(s, legacy)) for s in sources(legacy)).encode()).hexdigest()


def needs_build(legacy=None) -> bool:
    legacy = legacy_scans() if legacy is None else legacy
    out = LEGACY_OUT if legacy else OUT
    if _stale(out, [os.path.join(CSRC, s) for s in sources(legacy)] + HEADERS):
        return True
    try:
        with open(out + ".stamp") as f:
            return f.read().strip() != _stamp(legacy)
    except OSError:
        return True


def build_hip(force: bool = False, verbose: bool = False, legacy=None) -> str:
    """legacy = None: as PLSLAM_BUILD_LEGACY_SCANS says (default: the product, OUT); True: LEGACY_OUT."""
    legacy = legacy_scans() if legacy is None else legacy
    out = LEGACY_OUT if legacy else OUT
    if not force and not needs_build(legacy):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = hipcc_path()

    def compile_one(src: str):
        obj = _obj_for(src, legacy)
        if not force and not _stale(obj, [os.path.join(CSRC, src)] + HEADERS):
            return None
        cmd = [hipcc] + _flags_for(src, legacy) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            return f"hipcc failed on {src}:\n" + res.stdout + res.stderr
        return None

    workers = max(1, min(8, (os.cpu_count() or 2)))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        errs = [e for e in ex.map(compile_one, sources(legacy)) if e]
    if errs:
        raise RuntimeError("\n".join(errs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj_for(s, legacy) for s in sources(legacy)] + ["-o", out, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stdout + res.stderr)
    with open(out + ".stamp", "w") as f:
        f.write(_stamp(legacy) + "\n")
    return out


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
