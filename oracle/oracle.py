"""ctypes front end of the C oracle + an independent numpy mirror.

TEST INFRASTRUCTURE ONLY (see oracle/plslam_oracle.h for the parity status: matcher
semantics are "parity unpinned" -- OpenCV/stvo-pl are not under /root/reference; the
Hamming distance is pinned to the reference's own in-tree popcount code via oracle/_ref).

Reference anchors: src/mapHandler.cpp:277,424,597,712,3223,3249 (match call sites),
:1358-1540 / :1587-1772 (LBA rows), :605-613 / :720-729 (gates),
src/mapFeatures.cpp:51-93 (median descriptor).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_f64p = C.POINTER(C.c_double)


class FastMatching(C.Structure):
    """plo_fast_matching"""
    _fields_ = [("enabled", C.c_int32), ("grid_cols", C.c_int32), ("grid_rows", C.c_int32), ("ws", C.c_int32),
                ("inv_width", C.c_double), ("inv_height", C.c_double), ("nnr_grid", C.c_double),
                ("line_sim_th", C.c_double)]


class LbdLine(C.Structure):
    """plo_lbd_line"""
    _fields_ = [("num_pixels", C.c_int32), ("sx", C.c_float), ("sy", C.c_float), ("ex", C.c_float), ("ey", C.c_float),
                ("direction", C.c_float)]


class Cam(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("b", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


def _build(target: str = "libplslam_oracle.so") -> None:
    subprocess.run(["make", "-C", _HERE, target], check=True, stdout=subprocess.DEVNULL)


def _load(name: str) -> C.CDLL:
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        _build("native" if "native" in name else name)
    return C.CDLL(path)


def _bind(lib: C.CDLL) -> C.CDLL:
    lib.plo_hamming256.restype = C.c_int
    lib.plo_hamming256_swar.restype = C.c_int
    lib.plo_hamming256_lut.restype = C.c_int
    for f in (lib.plo_hamming256, lib.plo_hamming256_swar, lib.plo_hamming256_lut):
        f.argtypes = [C.c_void_p, C.c_void_p]
    lib.plo_knn2.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.plo_knn2.restype = None
    lib.plo_match_nnr.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_void_p]
    lib.plo_match_nnr.restype = C.c_int32
    lib.plo_match.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_int, C.c_void_p]
    lib.plo_match.restype = C.c_int32
    lib.plo_match_prior.argtypes = lib.plo_match.argtypes
    lib.plo_match_prior.restype = C.c_int32
    lib.plo_match_batched.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
    lib.plo_match_batched.restype = None
    lib.plo_match_batched_mt.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_float, C.c_int, C.c_void_p,
                                                             C.c_void_p, C.c_int]
    lib.plo_match_batched_mt.restype = None
    lib.plo_median_desc.argtypes = [C.c_void_p, C.c_int32]
    lib.plo_median_desc.restype = C.c_int32
    for f in (lib.plo_inverse_se3, lib.plo_expmap_se3, lib.plo_logmap_se3):
        f.argtypes = [C.c_void_p, C.c_void_p]
        f.restype = None
    lib.plo_lba_point_rows.argtypes = [C.POINTER(Cam), C.c_double] + [C.c_void_p] * 5 + [C.c_int32] + \
        [C.c_void_p] * 4
    lib.plo_lba_point_rows.restype = None
    lib.plo_lba_line_rows.argtypes = [C.POINTER(Cam), C.c_double, C.c_int] + [C.c_void_p] * 5 + \
        [C.c_int32] + [C.c_void_p] * 4
    lib.plo_lba_line_rows.restype = None
    for f in (lib.plo_lba_accumulate_points, lib.plo_lba_accumulate_lines):
        f.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 2 + [C.c_int32] + [C.c_void_p] * 7
        f.restype = None
    for f in (lib.plo_map2kf_point_gate, lib.plo_map2kf_line_gate):
        f.argtypes = [C.POINTER(Cam), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                      C.c_double, C.c_void_p]
        f.restype = C.c_int32
    for f in (lib.plo_map2kf_match_points, lib.plo_map2kf_match_lines):
        f.argtypes = [C.POINTER(Cam)] + [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 3 + \
            [C.c_int32, C.c_float, C.c_int, C.c_double, C.c_int32, C.c_void_p]
        f.restype = C.c_int32
    for f in (lib.plo_map_point_visible, lib.plo_map_line_visible):
        f.argtypes = [C.POINTER(Cam), C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
        f.restype = None
    lib.plo_median_desc_batched.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    lib.plo_median_desc_batched.restype = None
    lib.plo_lbd_binarise.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.plo_lbd_binarise.restype = None
    lib.plo_lbd_binary_conversion.argtypes = [C.c_void_p, C.c_void_p]
    lib.plo_lbd_binary_conversion.restype = C.c_uint8
    lib.plo_map2kf_match_points_fast.argtypes = [C.POINTER(Cam)] + [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 3 + \
        [C.c_int32, C.c_float, C.c_int, C.c_double, C.c_int32, C.POINTER(FastMatching), C.c_void_p, C.POINTER(C.c_int32)]
    lib.plo_map2kf_match_points_fast.restype = C.c_int32
    lib.plo_map2kf_match_lines_fast.argtypes = [C.POINTER(Cam)] + [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 4 + \
        [C.c_int32, C.c_float, C.c_int, C.c_double, C.c_int32, C.POINTER(FastMatching), C.c_void_p, C.POINTER(C.c_int32)]
    lib.plo_map2kf_match_lines_fast.restype = C.c_int32
    lib.plo_stereo_point_gate.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_double,
                                          C.c_void_p, C.c_void_p]
    lib.plo_stereo_point_gate.restype = C.c_int32
    lib.plo_stereo_line_gate.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32] + [C.c_double] * 4 + \
        [C.c_void_p, C.c_void_p]
    lib.plo_stereo_line_gate.restype = C.c_int32
    lib.plo_line_segment_overlap_stereo.argtypes = [C.c_double] * 5
    lib.plo_line_segment_overlap_stereo.restype = C.c_double
    lib.plo_kf2kf_match_points.argtypes = [C.POINTER(Cam), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_int32, C.c_float, C.c_int, C.c_int32, C.POINTER(FastMatching), C.c_void_p,
                                           C.POINTER(C.c_int32)]
    lib.plo_kf2kf_match_points.restype = C.c_int32
    lib.plo_kf2kf_match_lines.argtypes = lib.plo_kf2kf_match_points.argtypes
    lib.plo_kf2kf_match_lines.restype = C.c_int32
    lib.plo_pose_gn_accumulate.argtypes = [C.POINTER(Cam), C.c_double] + [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 3 + \
        [C.c_int32] + [C.c_void_p] * 4
    lib.plo_pose_gn_accumulate.restype = None
    lib.plo_lbd_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.plo_lbd_compute.restype = None
    lib.plo_lbd_gauss_tables.argtypes = [C.c_int32, C.c_void_p, C.c_void_p]
    lib.plo_lbd_gauss_tables.restype = None
    lib.plo_match_grid.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32,
                                   C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_double,
                                   C.c_void_p, C.c_double, C.c_int, C.c_void_p]
    lib.plo_match_grid.restype = C.c_int32
    lib.plo_grid_fill_points.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.plo_grid_fill_points.restype = None
    lib.plo_get_line_coords.argtypes = [C.c_double] * 4 + [C.c_void_p, C.c_int32]
    lib.plo_get_line_coords.restype = C.c_int32
    lib.plo_normalize2.argtypes = [C.c_void_p]
    lib.plo_normalize2.restype = None
    return lib


_LIB = None
_NATIVE = None
_REF = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = _bind(_load("libplslam_oracle.so"))
    return _LIB


def native_lib() -> C.CDLL:
    """-march=native build made on the executing host (bench.py cpu_baseline leg only)."""
    global _NATIVE
    if _NATIVE is None:
        try:
            _build("native")
            _NATIVE = _bind(C.CDLL(os.path.join(_HERE, "libplslam_oracle_native.so")))
        except Exception:  # no compiler on this host: the portable build is the baseline
            _NATIVE = lib()
    return _NATIVE


def ref_lib():
    """oracle/_ref/libplslam_ref.so: the reference's OWN code compiled from /root/reference where it lies --
    the popcounts (bitops_custom.hpp:83-96, FORB.cpp:78-101) and the exact multi-index-hashing kNN search
    BinaryDescriptorMatcher::knnMatch (binary_descriptor_matcher.cpp:258-335).  None if never built."""
    global _REF
    if os.environ.get("PLSLAM_ORACLE_NO_REF"):       # lets the suite be run as on a machine that never saw the reference
        return None
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libplslam_ref.so")
        if not os.path.exists(p):
            if os.path.isdir("/root/reference/3rdparty"):
                _build("ref")
            if not os.path.exists(p):
                return None
        r = C.CDLL(p)
        r.ref_ld_match.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        r.ref_ld_match.restype = C.c_int
        r.ref_forb_distance.argtypes = [C.c_void_p, C.c_void_p]
        r.ref_forb_distance.restype = C.c_int
        if hasattr(r, "ref_lba_accumulate"):
            r.ref_lba_accumulate.argtypes = [C.c_int, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_void_p] * 4 + [C.c_int] + \
                [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3
            r.ref_lba_accumulate.restype = C.c_int
        if hasattr(r, "ref_map_visible"):
            r.ref_map_visible.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                          C.c_void_p, C.c_void_p]
            r.ref_map_visible.restype = C.c_int
            r.ref_map2kf_gate.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                          C.c_double, C.c_int, C.c_void_p]
            r.ref_map2kf_gate.restype = C.c_int
        if hasattr(r, "ref_pose_gn_accumulate"):
            r.ref_pose_gn_accumulate.argtypes = [C.c_int, C.c_void_p, C.c_double] + [C.c_void_p] * 4 + [C.c_int] + \
                [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 4
            r.ref_pose_gn_accumulate.restype = C.c_int
        if hasattr(r, "ref_median_desc_point"):
            for f in (r.ref_median_desc_point, r.ref_median_desc_line):
                f.argtypes = [C.c_void_p, C.c_int]
                f.restype = C.c_int
        if hasattr(r, "ref_lbd_compute"):
            r.ref_lbd_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
            r.ref_lbd_compute.restype = C.c_int
            r.ref_lbd_gauss_tables.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
            r.ref_lbd_gauss_tables.restype = C.c_int
            r.ref_lbd_binary_conversion.argtypes = [C.c_void_p, C.c_void_p]
            r.ref_lbd_binary_conversion.restype = C.c_int
        if hasattr(r, "ref_mih_knn"):        # a prebuilt library from before the MIH wrapper lacks it
            r.ref_mih_knn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
            r.ref_mih_knn.restype = C.c_int
        _REF = r
    return _REF


def ref_lba_accumulate(iter_pass, cam, homog_th, nkf, T_map, T_slot, Xw, Lw, pt_lm, pt_kf_map, pt_kf_loc, pt_uv,
                       ls_lm, ls_kf_map, ls_kf_loc, ls_l):
    """The reference's OWN local-BA observation loops (src/mapHandler.cpp:1358-1431 + :1436-1540, or with iter_pass the
    iteration-pass loops :1587-1666 + :1668-1772; with iter_pass == "gba" the first-pass loops of
    levMarquardtOptimizationGBA :2124-2228 + :2233-2355), compiled textually from where they lie (oracle/ref_wrap_lba.cpp)
    -> (H, g, err) as they stand after both loops.  Landmark map index == local index; T_map: poses by key-frame map
    index; T_slot: poses of the nkf optimised slots (what expmap_se3 of X's pose blocks gives in the iteration pass).
    None if unavailable."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_lba_accumulate"):
        return None
    T_map = _c(T_map, np.float64).reshape(-1, 16)
    T_slot = _c(T_slot, np.float64).reshape(-1, 16)
    Xw, Lw = _c(Xw, np.float64).reshape(-1, 3), _c(Lw, np.float64).reshape(-1, 6)
    npt, nls = Xw.shape[0], Lw.shape[0]
    N = 6 * nkf + 3 * npt + 6 * nls
    H, g, err = np.zeros((N, N)), np.zeros(N), np.zeros(1)
    a = [_c(x, np.int32) for x in (pt_lm, pt_kf_map, pt_kf_loc)] + [_c(pt_uv, np.float64)]
    b = [_c(x, np.int32) for x in (ls_lm, ls_kf_map, ls_kf_loc)] + [_c(ls_l, np.float64)]
    c4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy])
    mode = 2 if iter_pass == "gba" else int(bool(iter_pass))
    rc = r.ref_lba_accumulate(mode, c4.ctypes.data, float(homog_th), nkf, npt, nls, T_map.ctypes.data,
                              T_map.shape[0], T_slot.ctypes.data, Xw.ctypes.data, Lw.ctypes.data,
                              *[x.ctypes.data for x in a], a[0].shape[0], *[x.ctypes.data for x in b], b[0].shape[0],
                              H.ctypes.data, g.ctypes.data, err.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"ref_lba_accumulate rc={rc}")
    return H, g, float(err[0])


def ref_lba_lm(cam, homog_th, lambda_lm, lambda_k, max_iters, min_err_change, min_err, nkf, T_map, x_kf, Xw, Lw,
               pt_lm, pt_kf_map, pt_kf_loc, pt_uv, ls_lm, ls_kf_map, ls_kf_loc, ls_l):
    """The reference's OWN Levenberg-Marquardt loop of levMarquardtOptimizationLBA (src/mapHandler.cpp:1334-1812: first pass,
    lambda = lambdaLbaLM * Hmax, the first damped solve, the iterations with their stop tests and lambda schedule), compiled
    textually from where it lies (oracle/ref_wrap_lba_lm.cpp; stvo-pl's SE(3) maps restated, the sparse LDL^T replaced by a dense
    envelope one) -> dict(X (N: the final state, poses as se3 logs), lam, err (one entry per solve, err as the text normalises
    it), iters, err_last, err_prev, lam_last).  None if unavailable."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_lba_lm"):
        return None
    T_map = _c(T_map, np.float64).reshape(-1, 16)
    x_kf = _c(x_kf, np.float64).reshape(-1)
    assert x_kf.shape[0] == 6 * nkf
    Xw, Lw = _c(Xw, np.float64).reshape(-1, 3), _c(Lw, np.float64).reshape(-1, 6)
    npt, nls = Xw.shape[0], Lw.shape[0]
    N = 6 * nkf + 3 * npt + 6 * nls
    a = [_c(x, np.int32) for x in (pt_lm, pt_kf_map, pt_kf_loc)] + [_c(pt_uv, np.float64)]
    b = [_c(x, np.int32) for x in (ls_lm, ls_kf_map, ls_kf_loc)] + [_c(ls_l, np.float64)]
    c4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy])
    cfg = np.array([homog_th, lambda_lm, lambda_k, float(max_iters), min_err_change, min_err], np.float64)
    X, trace, scal = np.zeros(N), np.zeros(2 * (int(max_iters) + 1)), np.zeros(5)
    r.ref_lba_lm.restype = C.c_int
    rc = r.ref_lba_lm(*[C.c_void_p(v) for v in (c4.ctypes.data, cfg.ctypes.data)], int(nkf), int(npt), int(nls),
                      C.c_void_p(T_map.ctypes.data), int(T_map.shape[0]),
                      *[C.c_void_p(v.ctypes.data) for v in (x_kf, Xw, Lw)],
                      *[C.c_void_p(x.ctypes.data) for x in a], int(a[0].shape[0]),
                      *[C.c_void_p(x.ctypes.data) for x in b], int(b[0].shape[0]),
                      *[C.c_void_p(v.ctypes.data) for v in (X, trace, scal)])
    if rc != 0:
        raise RuntimeError(f"ref_lba_lm rc={rc}")
    ns = int(scal[4])
    return dict(X=X, lam=trace[0:2 * ns:2].copy(), err=trace[1:2 * ns:2].copy(), iters=int(scal[0]), err_last=float(scal[1]),
                err_prev=float(scal[2]), lam_last=float(scal[3]))


def _cam6(cam):
    return np.array([cam.fx, cam.fy, cam.cx, cam.cy, float(cam.width), float(cam.height)])


def ref_map_visible(kind, cam, Twf, X, inv_w=1.0, inv_h=1.0):
    """The reference's OWN visibility pre-filter loop of matchMap2KFPoints (src/mapHandler.cpp:545-558) / Lines
    (:647-663), compiled textually -> (vis mask, normalised projections of the selected landmarks).  None if unavailable."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_map_visible"):
        return None
    lines = kind == "lines"
    X = _c(X, np.float64).reshape(-1, 6 if lines else 3)
    T = _c(Twf, np.float64).reshape(16)
    vis = np.zeros(X.shape[0], np.uint8)
    pj = np.zeros((X.shape[0], 4 if lines else 2))
    c6 = _cam6(cam)
    ns = r.ref_map_visible(int(lines), c6.ctypes.data, T.ctypes.data, X.ctypes.data, X.shape[0], float(inv_w), float(inv_h),
                           vis.ctypes.data, pj.ctypes.data)
    if ns < 0:
        raise RuntimeError(f"ref_map_visible rc={ns}")
    return vis, pj[:ns]


def ref_map2kf_gate(kind, cam, Twf, X, m12, obs, max_epip):
    """The reference's OWN gate loop of matchMap2KFPoints (src/mapHandler.cpp:601-629) / Lines (:716-749), compiled
    textually -> (mask of the landmarks that received the observation, final `matches`).  None if unavailable."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_map2kf_gate"):
        return None
    lines = kind == "lines"
    X = _c(X, np.float64).reshape(-1, 6 if lines else 3)
    obs = _c(obs, np.float64).reshape(-1, 3 if lines else 2)
    T = _c(Twf, np.float64).reshape(16)
    m12 = _c(m12, np.int32)
    mask = np.zeros(X.shape[0], np.uint8)
    c6 = _cam6(cam)
    n = r.ref_map2kf_gate(int(lines), c6.ctypes.data, T.ctypes.data, X.ctypes.data, m12.ctypes.data, X.shape[0],
                          obs.ctypes.data, obs.shape[0], float(max_epip), int((m12 >= 0).sum()), mask.ctypes.data)
    if n <= -1000000:
        raise RuntimeError("ref_map2kf_gate failed")
    return mask, int(n)


def ref_pose_gn_accumulate(robust, cam, homog_th, T_inc, P, pl_obs, pt_inlier, sPeP, le_obs, ls_inlier):
    """The reference's OWN pose-only Gauss-Newton loops (computeRelativePoseGN src/mapHandler.cpp:3331-3426, or with
    `robust` the first pair of loops of computeRelativePoseRobustGN :3595-3689), compiled textually from where they
    lie; same arguments and results as pose_gn_accumulate.  None if unavailable."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_pose_gn_accumulate"):
        return None
    T = _c(T_inc, np.float64).reshape(16)
    P, po = _c(P, np.float64).reshape(-1, 3), _c(pl_obs, np.float64).reshape(-1, 2)
    S, lo = _c(sPeP, np.float64).reshape(-1, 6), _c(le_obs, np.float64).reshape(-1, 3)
    pi, li = _c(pt_inlier, np.uint8), _c(ls_inlier, np.uint8)
    H, g, e, n = np.empty((6, 6)), np.empty(6), np.empty(1), np.empty(2, np.int32)
    c4 = np.array([cam.fx, cam.fy, cam.cx, cam.cy])
    rc = r.ref_pose_gn_accumulate(int(bool(robust)), c4.ctypes.data, float(homog_th), T.ctypes.data, P.ctypes.data,
                                  po.ctypes.data, pi.ctypes.data, P.shape[0], S.ctypes.data, lo.ctypes.data, li.ctypes.data,
                                  S.shape[0], H.ctypes.data, g.ctypes.data, e.ctypes.data, n.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"ref_pose_gn_accumulate rc={rc}")
    return H, g, float(e[0]), (int(n[0]), int(n[1]))


def ref_median_desc(descs, kind="point"):
    """The reference's OWN MapPoint / MapLine::updateAverageDescDir (src/mapFeatures.cpp:51-93 / :121-163, compiled
    from where it lies into oracle/_ref): the landmark is built from observation 0 and the others are added one by one;
    returns the index of the observation that ended up as med_desc.  None if unavailable."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_median_desc_point"):
        return None
    d = np.ascontiguousarray(descs, np.uint8).reshape(-1, 32)
    f = r.ref_median_desc_point if kind == "point" else r.ref_median_desc_line
    return int(f(d.ctypes.data, d.shape[0]))


def ref_lbd_compute(dx_img, dy_img, lines, width_of_band=7):
    """The reference's OWN BinaryDescriptor::computeLBD (binary_descriptor_custom.cpp:1026-1372, compiled from where
    it lies into oracle/_ref) on the given gradient images; same arguments as lbd_compute.  None if unavailable."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_lbd_compute"):
        return None
    dx, dy = _c(dx_img, np.int16), _c(dy_img, np.int16)
    ln = np.ascontiguousarray(lines, dtype=LBD_LINE_DTYPE)
    out = np.empty((ln.shape[0], 72), np.float32)
    rc = r.ref_lbd_compute(dx.ctypes.data, dy.ctypes.data, dx.shape[1], dx.shape[0], ln.ctypes.data, ln.shape[0],
                           int(width_of_band), out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"ref_lbd_compute rc={rc}")
    return out


def ref_lbd_gauss_tables(width_of_band=7):
    """gaussCoefL_ / gaussCoefG_ as the reference's BinaryDescriptor constructor builds them (:217-259)."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_lbd_compute"):
        return None
    cl, cg = np.empty(3 * width_of_band), np.empty(9 * width_of_band)
    if r.ref_lbd_gauss_tables(int(width_of_band), cl.ctypes.data, cg.ctypes.data) != 0:
        raise RuntimeError("ref_lbd_gauss_tables")
    return cl, cg


def ref_lbd_binary_conversion(f1, f2):
    """The reference's BinaryDescriptor::binaryConversion (:401-412) on two 8-float groups."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_lbd_compute"):
        return None
    a, b = _c(f1, np.float32), _c(f2, np.float32)
    return int(r.ref_lbd_binary_conversion(a.ctypes.data, b.ctypes.data))


def ref_mih_knn(q, t, k):
    """The reference's in-tree exact kNN (see ref_lib): (idx, dist) of shape (nq, k), or None if unavailable."""
    r = ref_lib()
    if r is None or not hasattr(r, "ref_mih_knn"):
        return None
    q = np.ascontiguousarray(q, np.uint8)
    t = np.ascontiguousarray(t, np.uint8)
    idx = np.empty((len(q), k), np.int32)
    dist = np.empty((len(q), k), np.int32)
    rc = r.ref_mih_knn(q.ctypes.data, len(q), t.ctypes.data, len(t), k, idx.ctypes.data, dist.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"ref_mih_knn rc={rc}")
    return idx, dist


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _desc(a):
    a = _c(a, np.uint8).reshape(-1, 32)
    return a


# ------------------------------------------------------------------------------------------
# C oracle wrappers
# ------------------------------------------------------------------------------------------
def hamming256(a, b, variant="popcnt32") -> int:
    a, b = _c(a, np.uint8), _c(b, np.uint8)
    f = {"popcnt32": lib().plo_hamming256, "swar": lib().plo_hamming256_swar,
         "lut": lib().plo_hamming256_lut}[variant]
    return int(f(_p(a), _p(b)))


def knn2(q, t, L=None):
    L = L or lib()
    q, t = _desc(q), _desc(t)
    idx = np.empty((q.shape[0], 2), np.int32)
    dist = np.empty((q.shape[0], 2), np.int32)
    L.plo_knn2(_p(q), q.shape[0], _p(t), t.shape[0], _p(idx), _p(dist))
    return idx, dist


def match(d1, d2, nnr, mutual=True, L=None):
    L = L or lib()
    d1, d2 = _desc(d1), _desc(d2)
    m12 = np.empty(d1.shape[0], np.int32)
    n = L.plo_match(_p(d1), d1.shape[0], _p(d2), d2.shape[0], float(nnr), int(bool(mutual)), _p(m12))
    return m12, int(n)


def match_prior(d1, d2, nnr, mutual, prior, L=None):
    """match() on a matches_12 that already holds entries (the fall-back after matchGrid): plo_match_prior."""
    L = L or lib()
    d1, d2 = _desc(d1), _desc(d2)
    m12 = np.array(prior, np.int32, copy=True)
    assert m12.shape == (d1.shape[0],)
    n = L.plo_match_prior(_p(d1), d1.shape[0], _p(d2), d2.shape[0], float(nnr), int(bool(mutual)), _p(m12))
    return m12, int(n)


def match_batched(d1, off1, d2, off2, nnr, mutual=True, nthreads=1, L=None):
    L = L or lib()
    d1, d2 = _desc(d1), _desc(d2)
    off1, off2 = _c(off1, np.int32), _c(off2, np.int32)
    B = off1.shape[0] - 1
    m12 = np.empty(int(off1[-1]), np.int32)
    nm = np.empty(B, np.int32)
    if nthreads > 1:
        L.plo_match_batched_mt(_p(d1), _p(off1), _p(d2), _p(off2), B, float(nnr), int(bool(mutual)),
                               _p(m12), _p(nm), int(nthreads))
    else:
        L.plo_match_batched(_p(d1), _p(off1), _p(d2), _p(off2), B, float(nnr), int(bool(mutual)),
                            _p(m12), _p(nm))
    return m12, nm


def median_desc(descs) -> int:
    d = _desc(descs)
    return int(lib().plo_median_desc(_p(d), d.shape[0]))


def median_desc_batched(desc_lists, offsets):
    """updateAverageDescDir (src/mapFeatures.cpp:51-84) for every landmark -> (med_idx, med_desc)."""
    d, off = _desc(desc_lists), _c(offsets, np.int32)
    n_lm = off.shape[0] - 1
    idx = np.empty(n_lm, np.int32)
    md = np.empty((n_lm, 32), np.uint8)
    lib().plo_median_desc_batched(_p(d), _p(off), n_lm, _p(idx), _p(md))
    return idx, md


def np_median_desc(descs) -> int:
    """numpy mirror of median_desc: full matrix, np.sort rows, literal index expression, argmin
    (first minimum)."""
    d = _desc(descs)
    n = d.shape[0]
    if n <= 1:
        return 0
    D = np.sort(np_dist_matrix(d, d), axis=1)
    return int(np.argmin(D[:, int(1 + 0.5 * (n - 1))]))


def inverse_se3(T):
    T = _c(T, np.float64).reshape(4, 4)
    o = np.empty((4, 4))
    lib().plo_inverse_se3(_p(T), _p(o))
    return o


def expmap_se3(x):
    x = _c(x, np.float64).reshape(6)
    o = np.empty((4, 4))
    lib().plo_expmap_se3(_p(x), _p(o))
    return o


def logmap_se3(T):
    T = _c(T, np.float64).reshape(4, 4)
    o = np.empty(6)
    lib().plo_logmap_se3(_p(T), _p(o))
    return o


def make_cam(fx, fy, cx, cy, b=0.0, width=0, height=0) -> Cam:
    return Cam(fx, fy, cx, cy, b, width, height)


def lba_point_rows(cam: Cam, homog_th, T_kf_w, Xw, obs_uv, lm_loc, kf_slot):
    T = _c(T_kf_w, np.float64).reshape(-1, 16)
    Xw = _c(Xw, np.float64).reshape(-1, 3)
    uv = _c(obs_uv, np.float64).reshape(-1, 2)
    lm, kf = _c(lm_loc, np.int32), _c(kf_slot, np.int32)
    n = uv.shape[0]
    Jp, Jl, r, w = np.empty((n, 6)), np.empty((n, 3)), np.empty(n), np.empty(n)
    lib().plo_lba_point_rows(C.byref(cam), float(homog_th), _p(T), _p(Xw), _p(uv), _p(lm), _p(kf), n,
                             _p(Jp), _p(Jl), _p(r), _p(w))
    return Jp, Jl, r, w


def lba_line_rows(cam: Cam, homog_th, T_kf_w, Lw, l_obs, lm_loc, kf_slot, compat_iter_pass=False):
    T = _c(T_kf_w, np.float64).reshape(-1, 16)
    Lw = _c(Lw, np.float64).reshape(-1)
    lo = _c(l_obs, np.float64).reshape(-1, 3)
    lm, kf = _c(lm_loc, np.int32), _c(kf_slot, np.int32)
    n = lo.shape[0]
    Jp, Jl, r, w = np.empty((n, 6)), np.empty((n, 6)), np.empty(n), np.empty(n)
    lib().plo_lba_line_rows(C.byref(cam), float(homog_th), int(bool(compat_iter_pass)), _p(T), _p(Lw),
                            _p(lo), _p(lm), _p(kf), n, _p(Jp), _p(Jl), _p(r), _p(w))
    return Jp, Jl, r, w


def lba_accumulate(kind, nkf, npt, nls, lm_loc, kf_loc, Jp, Jl, r, w, H=None, g=None):
    N = 6 * nkf + 3 * npt + 6 * nls
    H = np.zeros((N, N)) if H is None else H
    g = np.zeros(N) if g is None else g
    err = np.zeros(1)
    lm, kf = _c(lm_loc, np.int32), _c(kf_loc, np.int32)
    Jp, Jl, r, w = (_c(x, np.float64) for x in (Jp, Jl, r, w))
    f = lib().plo_lba_accumulate_points if kind == "points" else lib().plo_lba_accumulate_lines
    f(nkf, npt, nls, _p(lm), _p(kf), r.shape[0], _p(Jp), _p(Jl), _p(r), _p(w), _p(H), _p(g), _p(err))
    return H, g, float(err[0])


def map2kf_point_gate(cam, Twf, Xw, m12, pl, max_epip):
    Twf = _c(Twf, np.float64).reshape(16)
    Xw = _c(Xw, np.float64).reshape(-1, 3)
    m12 = _c(m12, np.int32)
    pl = _c(pl, np.float64).reshape(-1, 2)
    mask = np.empty(m12.shape[0], np.uint8)
    n = lib().plo_map2kf_point_gate(C.byref(cam), _p(Twf), _p(Xw), _p(m12), m12.shape[0], _p(pl),
                                    float(max_epip), _p(mask))
    return mask, int(n)


def map2kf_line_gate(cam, Twf, Lw, m12, le, max_epip):
    Twf = _c(Twf, np.float64).reshape(16)
    Lw = _c(Lw, np.float64).reshape(-1, 6)
    m12 = _c(m12, np.int32)
    le = _c(le, np.float64).reshape(-1, 3)
    mask = np.empty(m12.shape[0], np.uint8)
    n = lib().plo_map2kf_line_gate(C.byref(cam), _p(Twf), _p(Lw), _p(m12), m12.shape[0], _p(le),
                                   float(max_epip), _p(mask))
    return mask, int(n)


def map_point_visible(cam, Twf, Xw):
    Twf = _c(Twf, np.float64).reshape(16)
    Xw = _c(Xw, np.float64).reshape(-1, 3)
    vis = np.empty(Xw.shape[0], np.uint8)
    lib().plo_map_point_visible(C.byref(cam), _p(Twf), _p(Xw), Xw.shape[0], _p(vis))
    return vis


def map_line_visible(cam, Twf, Lw):
    Twf = _c(Twf, np.float64).reshape(16)
    Lw = _c(Lw, np.float64).reshape(-1, 6)
    vis = np.empty(Lw.shape[0], np.uint8)
    lib().plo_map_line_visible(C.byref(cam), _p(Twf), _p(Lw), Lw.shape[0], _p(vis))
    return vis


def map2kf_match(kind, cam, Twf, LM, med_desc, candidate, kf_desc, kf_feat, kf_idx, nnr, mutual, max_epip,
                 min_matches):
    """MapHandler::matchMap2KFPoints / Lines (src/mapHandler.cpp:532-752), BF path."""
    lw, fw = (3, 2) if kind == "points" else (6, 3)
    Twf = _c(Twf, np.float64).reshape(16)
    LM = _c(LM, np.float64).reshape(-1, lw)
    md, cand = _desc(med_desc), _c(candidate, np.uint8)
    kd, kf = _desc(kf_desc), _c(kf_feat, np.float64).reshape(-1, fw)
    ki = _c(kf_idx, np.int32)
    out = np.empty(LM.shape[0], np.int32)
    f = lib().plo_map2kf_match_points if kind == "points" else lib().plo_map2kf_match_lines
    n = f(C.byref(cam), _p(Twf), _p(LM), _p(md), _p(cand), LM.shape[0], _p(kd), _p(kf), _p(ki), kd.shape[0],
          float(nnr), int(bool(mutual)), float(max_epip), int(min_matches), _p(out))
    return out, int(n)


def map2kf_match_fast(kind, cam, Twf, LM, med_desc, candidate, kf_desc, kf_feat, kf_idx, nnr, mutual, max_epip,
                      min_matches, fm, kf_seg=None):
    """The drivers with SlamConfig::fastMatching() (src/mapHandler.cpp:578-598, :681-713) -> (map_to_kf, n, used_match).
    fm: dict(enabled, grid_cols, grid_rows, ws, inv_width, inv_height, nnr_grid, line_sim_th)."""
    lw, fw = (3, 2) if kind == "points" else (6, 3)
    Twf = _c(Twf, np.float64).reshape(16)
    LM = _c(LM, np.float64).reshape(-1, lw)
    md, cand = _desc(med_desc), _c(candidate, np.uint8)
    kd, kf = _desc(kf_desc), _c(kf_feat, np.float64).reshape(-1, fw)
    ki = _c(kf_idx, np.int32)
    out = np.empty(LM.shape[0], np.int32)
    F = FastMatching(int(fm["enabled"]), int(fm["grid_cols"]), int(fm["grid_rows"]), int(fm["ws"]), float(fm["inv_width"]),
                     float(fm["inv_height"]), float(fm["nnr_grid"]), float(fm.get("line_sim_th", 0.75)))
    used = C.c_int32()
    if kind == "points":
        n = lib().plo_map2kf_match_points_fast(C.byref(cam), _p(Twf), _p(LM), _p(md), _p(cand), LM.shape[0], _p(kd), _p(kf),
                                               _p(ki), kd.shape[0], float(nnr), int(bool(mutual)), float(max_epip),
                                               int(min_matches), C.byref(F), _p(out), C.byref(used))
    else:
        sg = _c(kf_seg, np.float64).reshape(-1, 4)
        n = lib().plo_map2kf_match_lines_fast(C.byref(cam), _p(Twf), _p(LM), _p(md), _p(cand), LM.shape[0], _p(kd), _p(kf),
                                              _p(sg), _p(ki), kd.shape[0], float(nnr), int(bool(mutual)), float(max_epip),
                                              int(min_matches), C.byref(F), _p(out), C.byref(used))
    return out, int(n), int(used.value)


def kf2kf_match(kind, cam, DT, X_prev, desc_prev, feat_curr, desc_curr, nnr, mutual, min_matches, fm):
    """MapHandler::matchKF2KFPoints / Lines compute part (src/mapHandler.cpp:246-278, :378-426) -> (matches_12, n, used_match)."""
    xw, fw = (3, 2) if kind == "points" else (6, 4)
    DT = _c(DT, np.float64).reshape(16)
    X = _c(X_prev, np.float64).reshape(-1, xw)
    dp, dc = _desc(desc_prev), _desc(desc_curr)
    fc = _c(feat_curr, np.float64).reshape(-1, fw)
    out = np.empty(X.shape[0], np.int32)
    F = FastMatching(int(fm["enabled"]), int(fm["grid_cols"]), int(fm["grid_rows"]), int(fm["ws"]), float(fm["inv_width"]),
                     float(fm["inv_height"]), float(fm["nnr_grid"]), float(fm.get("line_sim_th", 0.75)))
    used = C.c_int32()
    f = lib().plo_kf2kf_match_points if kind == "points" else lib().plo_kf2kf_match_lines
    n = f(C.byref(cam), _p(DT), _p(X), _p(dp), X.shape[0], _p(fc), _p(dc), fc.shape[0], float(nnr), int(bool(mutual)),
          int(min_matches), C.byref(F), _p(out), C.byref(used))
    return out, int(n), int(used.value)


def pose_gn_accumulate(cam, homog_th, T_inc, P, pl_obs, pt_inlier, sPeP, le_obs, ls_inlier):
    """computeRelativePoseGN iteration body (src/mapHandler.cpp:3324-3424) -> (H[6,6], g[6], e, (N_p, N_l))."""
    T = _c(T_inc, np.float64).reshape(16)
    P, po = _c(P, np.float64).reshape(-1, 3), _c(pl_obs, np.float64).reshape(-1, 2)
    S, lo = _c(sPeP, np.float64).reshape(-1, 6), _c(le_obs, np.float64).reshape(-1, 3)
    pi, li = _c(pt_inlier, np.uint8), _c(ls_inlier, np.uint8)
    H, g, e, n = np.empty((6, 6)), np.empty(6), np.empty(1), np.empty(2, np.int32)
    lib().plo_pose_gn_accumulate(C.byref(cam), float(homog_th), _p(T), _p(P), _p(po), _p(pi), P.shape[0], _p(S), _p(lo),
                                 _p(li), S.shape[0], _p(H), _p(g), _p(e), _p(n))
    return H, g, float(e[0]), (int(n[0]), int(n[1]))


def stereo_point_gate(m12, kp_l, kp_r, max_dist_epip, min_disp):
    """StereoFrame::matchStereoPoints gates ([RECALL]) -> (stereo_12, disp, n)."""
    m12 = _c(m12, np.int32)
    a, b = _c(kp_l, np.float32).reshape(-1, 2), _c(kp_r, np.float32).reshape(-1, 2)
    out, disp = np.empty(m12.shape[0], np.int32), np.empty(m12.shape[0], np.float64)
    n = lib().plo_stereo_point_gate(_p(m12), m12.shape[0], _p(a), _p(b), b.shape[0], float(max_dist_epip), float(min_disp),
                                    _p(out), _p(disp))
    return out, disp, int(n)


def stereo_line_gate(m12, seg_l, seg_r, min_disp, line_horiz_th, stereo_overlap_th, ls_min_disp_ratio):
    """StereoFrame::matchStereoLines gates ([RECALL]) -> (stereo_12, disp_se[n,2], n)."""
    m12 = _c(m12, np.int32)
    a, b = _c(seg_l, np.float32).reshape(-1, 4), _c(seg_r, np.float32).reshape(-1, 4)
    out, disp = np.empty(m12.shape[0], np.int32), np.empty((m12.shape[0], 2), np.float64)
    n = lib().plo_stereo_line_gate(_p(m12), m12.shape[0], _p(a), _p(b), b.shape[0], float(min_disp), float(line_horiz_th),
                                   float(stereo_overlap_th), float(ls_min_disp_ratio), _p(out), _p(disp))
    return out, disp, int(n)


LBD_LINE_DTYPE = np.dtype([("num_pixels", np.int32), ("sx", np.float32), ("sy", np.float32), ("ex", np.float32),
                           ("ey", np.float32), ("direction", np.float32)])


def lbd_compute(dx_img, dy_img, lines, width_of_band=7):
    """BinaryDescriptor::computeLBD (binary_descriptor_custom.cpp:1026-1372) for the lines of one octave.
    dx_img / dy_img: (height, width) int16; lines: structured array LBD_LINE_DTYPE -> (n, 72) float32."""
    dx, dy = _c(dx_img, np.int16), _c(dy_img, np.int16)
    ln = np.ascontiguousarray(lines, dtype=LBD_LINE_DTYPE)
    out = np.empty((ln.shape[0], 72), np.float32)
    lib().plo_lbd_compute(_p(dx), _p(dy), dx.shape[1], dx.shape[0], _p(ln), ln.shape[0], int(width_of_band), _p(out))
    return out


def lbd_gauss_tables(width_of_band=7):
    cl, cg = np.empty(3 * width_of_band), np.empty(9 * width_of_band)
    lib().plo_lbd_gauss_tables(int(width_of_band), _p(cl), _p(cg))
    return cl, cg


def lbd_pairs():
    """combinations[32][2] (binary_descriptor_custom.cpp:74-107) as a (32, 2) int array."""
    arr = (C.c_int * 64).in_dll(lib(), "plo_lbd_pairs")
    return np.array(list(arr), np.int32).reshape(32, 2)


def lbd_binarise(lbd_f32):
    """computeImpl's binary conversion (binary_descriptor_custom.cpp:653-668): n x 72 f32 -> n x 32 u8."""
    f = _c(lbd_f32, np.float32).reshape(-1, 72)
    out = np.empty((f.shape[0], 32), np.uint8)
    lib().plo_lbd_binarise(_p(f), f.shape[0], _p(out))
    return out


def np_lbd_binarise(lbd_f32):
    """numpy mirror of lbd_binarise (different formulation: whole-array compare + packbits)."""
    f = _c(lbd_f32, np.float32).reshape(-1, 9, 8)
    pr = lbd_pairs()
    bits = f[:, pr[:, 0], :] > f[:, pr[:, 1], :]                 # n x 32 x 8, bit i = element i
    return np.packbits(bits, axis=2, bitorder="little").reshape(-1, 32)



# ------------------------------------------------------------------------------------------
# matchGrid (stvo-pl matching.cpp, [RECALL]; call sites src/mapHandler.cpp:271,418,591,706)
# ------------------------------------------------------------------------------------------
def grid_fill_points(xy, cols, rows):
    """grid.at(x, y).push_back(idx) for idx ascending (src/mapHandler.cpp:258-264) -> CSR (cell_start, cell_items)."""
    xy = _c(xy, np.int32).reshape(-1, 2)
    cs = np.empty(cols * rows + 1, np.int32)
    items = np.empty(max(xy.shape[0], 1), np.int32)
    lib().plo_grid_fill_points(_p(xy), xy.shape[0], cols, rows, _p(cs), _p(items))
    return cs, items[:cs[-1]].copy()


def get_line_coords(x1, y1, x2, y2):
    n = lib().plo_get_line_coords(float(x1), float(y1), float(x2), float(y2), None, 0)
    out = np.empty((max(n, 1), 2), np.int32)
    lib().plo_get_line_coords(float(x1), float(y1), float(x2), float(y2), _p(out), n)
    return out[:n]


def grid_fill_lines(seg, cols, rows):
    """Callers' line grid (src/mapHandler.cpp:395-411): every Bresenham cell of segment idx gets idx, idx ascending.
    seg: n x 4 real-valued (x1, y1, x2, y2) in grid units.  -> CSR (cell_start, cell_items)."""
    seg = _c(seg, np.float64).reshape(-1, 4)
    cells = [[] for _ in range(cols * rows)]
    for idx, (x1, y1, x2, y2) in enumerate(seg):
        for x, y in get_line_coords(x1, y1, x2, y2):
            if 0 <= x < cols and 0 <= y < rows:
                cells[int(x) * rows + int(y)].append(idx)
    cs = np.zeros(cols * rows + 1, np.int32)
    cs[1:] = np.cumsum([len(c) for c in cells])
    items = np.array([i for c in cells for i in c], np.int32)
    return cs, items


def normalize2(v):
    v = _c(v, np.float64).reshape(-1, 2).copy()
    for row in v:
        lib().plo_normalize2(_p(row))
    return v


def match_grid(centres, d1, cell_start, cell_items, cols, rows, d2, window, nnr, mutual=True, dir1=None, dir2=None,
               sim_th=0.0):
    d1, d2 = _desc(d1), _desc(d2)
    n1 = d1.shape[0]
    centres = _c(centres, np.int32).reshape(n1, -1, 2) if n1 else np.zeros((0, 1, 2), np.int32)
    cs, items = _c(cell_start, np.int32), _c(cell_items, np.int32)
    w = _c(window, np.int32)
    a = _c(dir1, np.float64) if dir1 is not None else None
    b = _c(dir2, np.float64) if dir2 is not None else None
    m12 = np.empty(n1, np.int32)
    n = lib().plo_match_grid(_p(centres), centres.shape[1], _p(d1), n1, _p(cs), _p(items), cols, rows, _p(d2),
                             d2.shape[0], _p(a) if a is not None else None, _p(b) if b is not None else None,
                             float(sim_th), _p(w), float(nnr), int(bool(mutual)), _p(m12))
    return m12, int(n)


def np_match_grid(centres, d1, cell_start, cell_items, cols, rows, d2, window, nnr, mutual=True, dir1=None,
                  dir2=None, sim_th=0.0):
    """Independent formulation of matchGrid -- the order-free one the device uses.  The sequential
    `if (d < distances[i2]) ... else continue` of upstream makes candidate (i1, i2) take part in row i1's
    best/second-best iff d(i1,i2) is a strict prefix minimum of column i2 in row order, i.e. iff
        i1 == min{ i1' : i2 in C(i1'), d(i1', i2) <= d(i1, i2) };
    matches_21[i2] is the lexicographic minimum (d, i1) of the column."""
    d1, d2 = _desc(d1), _desc(d2)
    n1, n2 = d1.shape[0], d2.shape[0]
    centres = _c(centres, np.int32).reshape(n1, -1, 2) if n1 else np.zeros((0, 1, 2), np.int32)
    cs, items = _c(cell_start, np.int32), _c(cell_items, np.int32)
    w = [int(v) for v in window]
    pairs = set()
    nonempty = np.zeros(n1, bool)          # rows whose candidate SET is not empty (before any per-candidate test)
    for i1 in range(n1):
        for x, y in centres[i1]:
            x, y = int(x), int(y)
            for x_ in range(max(0, x - w[0]), min(cols, x + w[1] + 1)):
                lo, hi = max(0, y - w[2]), min(rows, y + w[3] + 1)
                if lo < hi:
                    for i2 in items[cs[x_ * rows + lo]:cs[x_ * rows + hi]]:
                        nonempty[i1] = True
                        if 0 <= i2 < n2:
                            pairs.add((i1, int(i2)))
    imax = np.iinfo(np.int32).max
    # upstream quirk: a row with candidates of which none counts keeps best_d = best_d2 = INT_MAX, and
    # `INT_MAX < INT_MAX * nnr` holds for nnr > 1: it "matches" best_idx = -1 and is counted
    phantom = float(imax) < float(imax) * float(nnr)
    if not pairs:
        return np.full(n1, -1, np.int32), int(nonempty.sum()) if phantom else 0
    P = np.array(sorted(pairs), np.int64)
    if dir1 is not None and dir2 is not None:
        a, b = _c(dir1, np.float64).reshape(-1, 2), _c(dir2, np.float64).reshape(-1, 2)
        dot = a[P[:, 0], 0] * b[P[:, 1], 0] + a[P[:, 0], 1] * b[P[:, 1], 1]
        P = P[~(np.abs(dot) < sim_th)]
    D = np.bitwise_count(d1[P[:, 0]] ^ d2[P[:, 1]]).sum(axis=1).astype(np.int64)
    live = np.ones(len(P), bool)
    m21 = np.full(n2, -1, np.int64)
    if mutual:
        for i2 in np.unique(P[:, 1]):
            sel = np.nonzero(P[:, 1] == i2)[0]
            i1s, ds = P[sel, 0], D[sel]
            first = np.array([i1s[ds <= d].min() for d in ds])
            live[sel] = first == i1s
            m21[i2] = i1s[np.lexsort((i1s, ds))[0]]
    m12 = np.full(n1, -1, np.int32)
    has_live = np.zeros(n1, bool)
    for i1 in np.unique(P[:, 0]):
        sel = np.nonzero((P[:, 0] == i1) & live)[0]
        if not len(sel):
            continue
        has_live[i1] = True
        order = np.lexsort((P[sel, 1], D[sel]))
        best_d, best_idx = D[sel][order[0]], P[sel, 1][order[0]]
        best_d2 = D[sel][order[1]] if len(sel) > 1 else imax
        if float(best_d) < float(best_d2) * float(nnr):
            m12[i1] = best_idx
    if mutual:
        good = m12 >= 0
        m12 = np.where(good & (m21[np.clip(m12, 0, None)] == np.arange(n1)), m12, -1).astype(np.int32)
    return m12, int((m12 >= 0).sum()) + (int((nonempty & ~has_live).sum()) if phantom else 0)

# ------------------------------------------------------------------------------------------
# independent numpy mirror (different formulation: full distance matrix + stable sort)
# ------------------------------------------------------------------------------------------
def np_dist_matrix(q, t):
    q, t = _desc(q), _desc(t)
    x = q[:, None, :] ^ t[None, :, :]
    return np.bitwise_count(x).sum(axis=2, dtype=np.int32)


def np_knn2(q, t):
    """First two of a STABLE sort by distance == OpenCV batchDistance K=2 insertion order."""
    D = np_dist_matrix(q, t)
    nq, nt = D.shape
    idx = np.full((nq, 2), -1, np.int32)
    dist = np.full((nq, 2), np.iinfo(np.int32).max, np.int32)
    if nt:
        order = np.argsort(D, axis=1, kind="stable")[:, :2]
        k = order.shape[1]
        idx[:, :k] = order
        dist[:, :k] = np.take_along_axis(D, order, axis=1)
    return idx, dist


def np_match_nnr(q, t, nnr):
    idx, dist = np_knn2(q, t)
    nnr32 = np.float32(nnr)
    ok = (idx[:, 1] >= 0) & (dist[:, 0].astype(np.float32) < dist[:, 1].astype(np.float32) * nnr32)
    return np.where(ok, idx[:, 0], -1).astype(np.int32)


def np_match(d1, d2, nnr, mutual=True):
    m12 = np_match_nnr(d1, d2, nnr)
    if mutual:
        m21 = np_match_nnr(d2, d1, nnr)
        good = m12 >= 0
        back = np.where(good, m21[np.clip(m12, 0, max(len(m21) - 1, 0))] if len(m21) else -2, -2)
        m12 = np.where(good & (back == np.arange(len(m12))), m12, -1).astype(np.int32)
    return m12, int((m12 >= 0).sum())
