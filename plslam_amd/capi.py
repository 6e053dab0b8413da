"""ctypes binding of include/plslam_hip.h (libplslam_hip.so).

This module is plumbing: it declares the C-ABI entry points and wraps them for numpy
(host-pointer calls) and raw device pointers (plans).  It contains no algorithm and has NO
CPU fallback: if the HIP library is missing it raises, and without a gfx950 device
``Context()`` raises ``PlslamError(PLSLAM_ENODEV)``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libplslam_hip.so")
# experiment builds of the SAME library (tools/build_exp.py: timing-only variants of one kernel) are loaded through
# this variable by the tools under tools/; the product, the tests and bench.py never set it
if os.environ.get("PLSLAM_HIP_LIB_EXPERIMENT"):
    LIB_PATH = os.path.abspath(os.environ["PLSLAM_HIP_LIB_EXPERIMENT"])

OK, EINVAL, ENODEV, EHIP, ENOMEM, ERANGE, ENOTSUP = 0, -1, -2, -3, -4, -5, -6
SCAN_AUTO, SCAN_LANE_PER_QUERY, SCAN_WAVE_PER_QUERY, SCAN_SYMMETRIC, SCAN_MFMA = 0, 1, 2, 3, 4
MAX_TRAIN_ROWS = 1 << 23

# every symbol include/plslam_hip.h declares (tests check the .so exports all of them)
ABI_VERSION = 5          # include/plslam_hip.h: PLSLAM_ABI_VERSION
ABI_SYMBOLS = (
    "plslam_strerror", "plslam_last_error", "plslam_abi_version",
    "plslam_ctx_create", "plslam_ctx_destroy", "plslam_ctx_set_option", "plslam_ctx_get_option",
    "plslam_ctx_device_info",
    "plslam_knn2_hamming256", "plslam_match", "plslam_match_prior", "plslam_match_batched",
    "plslam_match_plan_create", "plslam_match_plan_run", "plslam_match_plan_run_split", "plslam_match_plan_set_profiling",
    "plslam_match_plan_elapsed", "plslam_match_plan_info", "plslam_match_plan_dump", "plslam_match_plan_key_state", "plslam_match_plan_destroy",
    "plslam_lba_point_rows", "plslam_lba_line_rows", "plslam_lba_point_rows_dev",
    "plslam_lba_line_rows_dev", "plslam_lba_assemble", "plslam_lba_plan_create", "plslam_lba_plan_iterate",
    "plslam_lba_plan_rows", "plslam_lba_plan_destroy", "plslam_lba_plan_iterate_dev", "plslam_lba_plan_device_blocks", "plslam_lba_plan_device_state",
    "plslam_lba_plan_iterate_resident", "plslam_lba_plan_diag_max", "plslam_lba_plan_schur", "plslam_lba_plan_backsub",
    "plslam_lba_plan_set_poses", "plslam_lba_plan_host_state", "plslam_lba_plan_get_landmarks",
    "plslam_lba_plan_iterate_schur", "plslam_lba_plan_apply_step", "plslam_lba_point_rows_dev_n", "plslam_lba_line_rows_dev_n", "plslam_rccl_available",
    "plslam_lba_plan_blocks",
    "plslam_map2kf_point_gate", "plslam_map2kf_line_gate", "plslam_map_point_visible",
    "plslam_map_line_visible", "plslam_map2kf_match_points", "plslam_map2kf_match_lines",
    "plslam_map2kf_match_points_fast", "plslam_map2kf_match_lines_fast", "plslam_map2kf_match_points_dev", "plslam_map2kf_match_lines_dev",
    "plslam_kf2kf_match_points", "plslam_kf2kf_match_lines", "plslam_kf2kf_match_points_dev", "plslam_kf2kf_match_lines_dev",
    "plslam_lbd_binarise", "plslam_lbd_binarise_dev", "plslam_lbd_compute", "plslam_lbd_compute_dev",
    "plslam_median_desc_batched", "plslam_median_desc_batched_dev",
    "plslam_stereo_point_gate", "plslam_stereo_line_gate", "plslam_stereo_point_gate_dev", "plslam_stereo_line_gate_dev",
    "plslam_match_plan_add_stereo_gates", "plslam_match_plan_set_wire16", "plslam_pose_gn_accumulate",
    "plslam_match_pipeline_create", "plslam_match_pipeline_submit", "plslam_match_pipeline_wait",
    "plslam_match_pipeline_destroy", "plslam_pinned_alloc", "plslam_pinned_free",
    "plslam_match_grid", "plslam_grid_plan_create", "plslam_grid_plan_run", "plslam_grid_plan_overflows",
    "plslam_grid_plan_destroy", "plslam_grid_pair_capacity", "plslam_grid_pair_capacity_bound",
    "plslam_gather_match_tables", "plslam_rccl_use", "plslam_match_plan_step_gather", "plslam_match_plan_gather_sync",
)


class Cam(C.Structure):
    """plslam_cam"""
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
                ("b", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class MatchProblem(C.Structure):
    """plslam_match_problem (device pointers)"""
    _fields_ = [("d1", C.c_void_p), ("d2", C.c_void_p), ("n1", C.c_int32), ("n2", C.c_int32),
                ("nnr", C.c_float), ("mutual", C.c_int32), ("matches_12", C.c_void_p),
                ("n_matches", C.c_void_p), ("keep_prior", C.c_int32), ("reserved", C.c_int32)]


class StereoGateProblem(C.Structure):
    """plslam_stereo_gate_problem (device pointers)"""
    _fields_ = [("matches_12", C.c_void_p), ("f_l", C.c_void_p), ("f_r", C.c_void_p), ("n_l", C.c_int32),
                ("n_r", C.c_int32), ("lines", C.c_int32), ("pad", C.c_int32), ("max_dist_epip", C.c_double),
                ("min_disp", C.c_double), ("line_horiz_th", C.c_double), ("stereo_overlap_th", C.c_double),
                ("ls_min_disp_ratio", C.c_double), ("stereo_12", C.c_void_p), ("disp", C.c_void_p),
                ("n_stereo", C.c_void_p)]


class GatherStep(C.Structure):
    """plslam_gather_step (include/plslam_hip.h)"""
    _fields_ = [("comm", C.c_void_p), ("nranks", C.c_int32), ("rank", C.c_int32), ("root", C.c_int32), ("wire_bytes", C.c_int32),
                ("send", C.c_void_p), ("n_entries", C.c_int64), ("recv", C.c_void_p), ("wide", C.c_void_p),
                ("scan_stream", C.c_void_p), ("post_stream", C.c_void_p), ("comm_stream", C.c_void_p)]


class ArenaProblem(C.Structure):
    """plslam_arena_problem"""
    _fields_ = [("d1_off", C.c_int64), ("d2_off", C.c_int64), ("n1", C.c_int32), ("n2", C.c_int32), ("nnr", C.c_float),
                ("mutual", C.c_int32), ("out_off", C.c_int64)]


class GridProblem(C.Structure):
    """plslam_grid_problem (device pointers)"""
    _fields_ = [("d1", C.c_void_p), ("d2", C.c_void_p), ("centres1", C.c_void_p), ("cell_start", C.c_void_p),
                ("cell_items", C.c_void_p), ("dir1", C.c_void_p), ("dir2", C.c_void_p),
                ("n1", C.c_int32), ("n2", C.c_int32), ("n_centres", C.c_int32), ("grid_cols", C.c_int32),
                ("grid_rows", C.c_int32), ("n_items", C.c_int32), ("window", C.c_int32 * 4), ("sim_th", C.c_double), ("nnr", C.c_double),
                ("mutual", C.c_int32), ("pair_capacity", C.c_int32), ("matches_12", C.c_void_p),
                ("n_matches", C.c_void_p)]


class FastMatching(C.Structure):
    """plslam_fast_matching"""
    _fields_ = [("enabled", C.c_int32), ("grid_cols", C.c_int32), ("grid_rows", C.c_int32), ("ws", C.c_int32),
                ("inv_width", C.c_double), ("inv_height", C.c_double), ("nnr_grid", C.c_double),
                ("line_sim_th", C.c_double)]


class PlanInfo(C.Structure):
    """plslam_plan_info"""
    _fields_ = [("distance_evals", C.c_int64), ("directed_evals", C.c_int64),
                ("algorithmic_bytes", C.c_int64), ("n_scans", C.c_int32),
                ("scan_blocks", C.c_int32), ("scan_variant", C.c_int32),
                ("scan_block_threads", C.c_int32)]


# plslam_lbd_line as a numpy record
LBD_LINE_DTYPE = np.dtype([("num_pixels", np.int32), ("sx", np.float32), ("sy", np.float32), ("ex", np.float32),
                           ("ey", np.float32), ("direction", np.float32)])


class PlslamError(RuntimeError):
    def __init__(self, code: int, where: str, detail: str = ""):
        self.code = code
        super().__init__(f"{where}: {code} ({detail})")


_lib = None


def _preload_shared_hip_runtime() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.  A process must run ONE HIP runtime: if
    this library pulled in /opt/rocm's copy first and torch initialised its own afterwards, torch
    would find no GPU.  When torch is installed, load ITS runtime first (without importing torch) so
    that the dynamic loader resolves our DT_NEEDED libamdhip64 to the same copy."""
    import glob
    import importlib.util
    if "torch" in __import__("sys").modules:
        return                                   # torch already brought its runtime in
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libamdhip64.so", "libamdhip64.so.*"):
        for path in sorted(glob.glob(os.path.join(libdir, name))):
            try:
                C.CDLL(path, mode=C.RTLD_GLOBAL)
                return
            except OSError:
                continue


def load() -> C.CDLL:
    """dlopen libplslam_hip.so and declare prototypes.  Fails loudly if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    _preload_shared_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32, f64 = C.c_void_p, C.c_int32, C.c_double
    L.plslam_strerror.restype = C.c_char_p
    L.plslam_strerror.argtypes = [C.c_int]
    L.plslam_last_error.restype = C.c_char_p
    L.plslam_last_error.argtypes = []
    L.plslam_abi_version.restype = C.c_int
    L.plslam_abi_version.argtypes = []
    if L.plslam_abi_version() != ABI_VERSION:
        raise RuntimeError(f"libplslam_hip.so has ABI version {L.plslam_abi_version()}, these bindings were written for {ABI_VERSION} "
                           "(struct layouts differ: rebuild with plslam_amd/build.py)")
    L.plslam_lbd_binarise.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.plslam_lbd_binarise_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.plslam_median_desc_batched.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.plslam_median_desc_batched_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p,
                                                 C.c_void_p, C.c_void_p]
    L.plslam_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.plslam_ctx_destroy.argtypes = [vp]
    L.plslam_ctx_destroy.restype = None
    L.plslam_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.plslam_ctx_get_option.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int)]
    L.plslam_ctx_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32),
                                         C.c_char_p, i32]
    L.plslam_knn2_hamming256.argtypes = [vp, vp, i32, vp, i32, vp, vp]
    L.plslam_match.argtypes = [vp, vp, i32, vp, i32, C.c_float, C.c_int, vp, C.POINTER(i32)]
    L.plslam_match_prior.argtypes = L.plslam_match.argtypes
    L.plslam_match_batched.argtypes = [vp, vp, vp, vp, vp, i32, C.c_float, C.c_int, vp, vp]
    L.plslam_match_plan_create.argtypes = [vp, C.POINTER(MatchProblem), i32, C.POINTER(vp)]
    L.plslam_match_plan_run.argtypes = [vp, vp]
    L.plslam_match_plan_run_split.argtypes = [vp, vp, vp]
    L.plslam_match_plan_set_profiling.argtypes = [vp, C.c_int]
    L.plslam_match_plan_elapsed.argtypes = [vp, C.POINTER(f64), C.POINTER(f64), C.POINTER(C.c_int64)]
    L.plslam_match_plan_info.argtypes = [vp, C.POINTER(PlanInfo)]
    L.plslam_match_plan_dump.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t),
                                         C.POINTER(C.c_size_t)]
    L.plslam_match_plan_key_state.argtypes = [vp, C.POINTER(C.c_int32)]
    L.plslam_match_plan_destroy.argtypes = [vp]
    L.plslam_match_plan_destroy.restype = None
    L.plslam_lba_point_rows.argtypes = [vp, C.POINTER(Cam), f64, vp, i32, vp, i32, vp, vp, vp, i32,
                                        vp, vp, vp, vp]
    L.plslam_lba_line_rows.argtypes = [vp, C.POINTER(Cam), f64, C.c_int, vp, i32, vp, i32, vp, vp, vp,
                                       i32, vp, vp, vp, vp]
    L.plslam_lba_point_rows_dev.argtypes = [vp, C.POINTER(Cam), f64, vp, vp, vp, vp, vp, i32,
                                            vp, vp, vp, vp, vp]
    L.plslam_lba_line_rows_dev.argtypes = [vp, C.POINTER(Cam), f64, C.c_int, vp, vp, vp, vp, vp, i32,
                                           vp, vp, vp, vp, vp]
    L.plslam_lba_assemble.argtypes = [vp, i32, i32, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp,
                                      vp, vp, vp, vp, vp, vp, vp]
    L.plslam_lba_point_rows_dev_n.argtypes = [vp, C.POINTER(Cam), f64, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.plslam_lba_line_rows_dev_n.argtypes = [vp, C.POINTER(Cam), f64, C.c_int, vp, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp]
    L.plslam_lba_plan_create.argtypes = [vp, C.POINTER(Cam), f64, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp, vp, vp,
                                         i32, C.POINTER(vp)]
    L.plslam_lba_plan_iterate.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.plslam_lba_plan_rows.argtypes = [vp] * 9
    L.plslam_lba_plan_iterate_dev.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp]
    L.plslam_lba_plan_device_blocks.argtypes = [vp, vp]
    L.plslam_lba_plan_device_state.argtypes = [vp, vp]
    L.plslam_lba_plan_iterate_resident.argtypes = [vp, C.c_int, vp]
    L.plslam_lba_plan_diag_max.argtypes = [vp, vp]
    L.plslam_lba_plan_schur.argtypes = [vp, C.c_double, vp, vp, vp]
    L.plslam_lba_plan_backsub.argtypes = [vp, vp, C.c_int, vp, vp]
    L.plslam_lba_plan_set_poses.argtypes = [vp, vp]
    L.plslam_lba_plan_get_landmarks.argtypes = [vp, vp, vp]
    L.plslam_lba_plan_iterate_schur.argtypes = [vp, C.c_int, C.c_double, vp, vp, vp, vp]
    L.plslam_lba_plan_apply_step.argtypes = [vp, vp, vp, C.c_int, vp]
    L.plslam_lba_plan_host_state.argtypes = [vp, vp]
    L.plslam_lba_plan_blocks.argtypes = [vp] * 8
    L.plslam_lba_plan_destroy.argtypes = [vp]
    L.plslam_lba_plan_destroy.restype = None
    for f in (L.plslam_map2kf_point_gate, L.plslam_map2kf_line_gate):
        f.argtypes = [vp, C.POINTER(Cam), vp, vp, vp, i32, vp, i32, f64, vp, C.POINTER(i32)]
    for f in (L.plslam_map_point_visible, L.plslam_map_line_visible):
        f.argtypes = [vp, C.POINTER(Cam), vp, vp, i32, vp]
    for f in (L.plslam_map2kf_match_points, L.plslam_map2kf_match_lines):
        f.argtypes = [vp, C.POINTER(Cam), vp, vp, vp, vp, i32, vp, vp, vp, i32, C.c_float, C.c_int, f64, i32, vp,
                      C.POINTER(i32)]
    L.plslam_map2kf_match_points_fast.argtypes = [vp, C.POINTER(Cam), vp, vp, vp, vp, i32, vp, vp, vp, i32, C.c_float,
                                                  C.c_int, f64, i32, C.POINTER(FastMatching), vp, C.POINTER(i32),
                                                  C.POINTER(i32)]
    L.plslam_map2kf_match_lines_fast.argtypes = [vp, C.POINTER(Cam), vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, C.c_float,
                                                 C.c_int, f64, i32, C.POINTER(FastMatching), vp, C.POINTER(i32),
                                                 C.POINTER(i32)]
    L.plslam_map2kf_match_points_dev.argtypes = L.plslam_map2kf_match_points_fast.argtypes
    L.plslam_map2kf_match_lines_dev.argtypes = L.plslam_map2kf_match_lines_fast.argtypes
    for f in (L.plslam_kf2kf_match_points, L.plslam_kf2kf_match_lines, L.plslam_kf2kf_match_points_dev,
              L.plslam_kf2kf_match_lines_dev):
        f.argtypes = [vp, C.POINTER(Cam), vp, vp, vp, i32, vp, vp, i32, C.c_float, C.c_int, i32, C.POINTER(FastMatching), vp,
                      C.POINTER(i32), C.POINTER(i32)]
    L.plslam_pose_gn_accumulate.argtypes = [vp, C.POINTER(Cam), f64, vp, vp, vp, vp, i32, vp, vp, vp, i32, vp, vp, vp, vp]
    L.plslam_lbd_compute.argtypes = [vp, vp, vp, i32, i32, vp, i32, i32, vp]
    L.plslam_lbd_compute_dev.argtypes = [vp, vp, vp, i32, i32, vp, i32, i32, vp, vp]
    L.plslam_stereo_point_gate.argtypes = [vp, vp, i32, vp, vp, i32, f64, f64, vp, vp, C.POINTER(i32)]
    L.plslam_stereo_line_gate.argtypes = [vp, vp, i32, vp, vp, i32, f64, f64, f64, f64, vp, vp, C.POINTER(i32)]
    L.plslam_stereo_point_gate_dev.argtypes = [vp, vp, i32, vp, vp, i32, f64, f64, vp, vp, vp, vp]
    L.plslam_stereo_line_gate_dev.argtypes = [vp, vp, i32, vp, vp, i32, f64, f64, f64, f64, vp, vp, vp, vp]
    L.plslam_match_plan_add_stereo_gates.argtypes = [vp, C.POINTER(StereoGateProblem), i32]
    L.plslam_match_plan_set_wire16.argtypes = [vp, vp, vp, C.c_size_t]
    L.plslam_match_pipeline_create.argtypes = [vp, C.c_size_t, C.POINTER(ArenaProblem), i32, C.c_size_t, i32, C.POINTER(vp)]
    L.plslam_match_pipeline_submit.argtypes = [vp, vp, vp, vp]
    L.plslam_match_pipeline_wait.argtypes = [vp]
    L.plslam_match_pipeline_destroy.argtypes = [vp]
    L.plslam_match_pipeline_destroy.restype = None
    L.plslam_pinned_alloc.argtypes = [C.c_size_t]
    L.plslam_pinned_alloc.restype = vp
    L.plslam_pinned_free.argtypes = [vp]
    L.plslam_pinned_free.restype = None
    L.plslam_match_grid.argtypes = [vp, vp, i32, vp, i32, vp, vp, i32, i32, vp, i32, vp, vp, f64, vp, f64, C.c_int, vp,
                                    C.POINTER(i32)]
    L.plslam_grid_plan_create.argtypes = [vp, C.POINTER(GridProblem), i32, C.POINTER(vp)]
    L.plslam_grid_plan_run.argtypes = [vp, vp]
    L.plslam_grid_plan_overflows.argtypes = [vp, vp, C.POINTER(i32)]
    L.plslam_grid_plan_destroy.argtypes = [vp]
    L.plslam_grid_plan_destroy.restype = None
    L.plslam_grid_pair_capacity.argtypes = [vp, i32, i32, vp, i32, i32, vp, C.c_int]
    L.plslam_grid_pair_capacity.restype = C.c_int64
    L.plslam_grid_pair_capacity_bound.argtypes = [i32, i32, vp, i32, i32, vp, C.c_int]
    L.plslam_grid_pair_capacity_bound.restype = C.c_int64
    L.plslam_gather_match_tables.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp, vp]
    L.plslam_rccl_use.argtypes = [C.c_char_p]
    L.plslam_match_plan_step_gather.argtypes = [vp, C.POINTER(GatherStep)]
    L.plslam_match_plan_gather_sync.argtypes = [vp]
    L.plslam_rccl_available.argtypes = []
    for name in ABI_SYMBOLS:
        f = getattr(L, name)
        if name not in ("plslam_strerror", "plslam_last_error", "plslam_ctx_destroy",
                        "plslam_match_plan_destroy", "plslam_lba_plan_destroy", "plslam_grid_plan_destroy",
                        "plslam_match_pipeline_destroy", "plslam_pinned_alloc", "plslam_pinned_free",
                        "plslam_grid_pair_capacity", "plslam_grid_pair_capacity_bound"):
            f.restype = C.c_int
    _lib = L
    return L


def _check(code: int, where: str) -> None:
    if code != OK:
        L = load()
        raise PlslamError(code, where, f"{L.plslam_strerror(code).decode()}; "
                                       f"{L.plslam_last_error().decode()}")


def _arr(a, dt, shape=None):
    a = np.ascontiguousarray(a, dtype=dt)
    if shape is not None:
        a = a.reshape(shape)
    return a


_from_buffer, _addressof = C.c_char.from_buffer, C.addressof


def _p(a):
    """The address of a contiguous array as an int (every bound function declares void* arguments), None for None.  Through the
    buffer protocol: `a.ctypes.data_as(...)` builds a ctypes helper object per call -- 2 us each, 20 us of a matchGrid call's 58."""
    if a is None:
        return None
    try:
        return _addressof(_from_buffer(a))
    except (TypeError, ValueError, BufferError):          # a read-only or an empty array
        return a.ctypes.data


def grid_pair_capacity(centres, cell_start, cols, rows, window, mutual=True, bound=False) -> int:
    """plslam_grid_problem.pair_capacity (pure host code, no device needed): the documented size, or -- bound=True -- an upper
    bound of it from the grid alone (centres: only their shape is used then)."""
    L = load()
    cen = _arr(centres, np.int32)
    cen = cen.reshape(cen.shape[0], -1, 2) if cen.size else cen.reshape(0, 1, 2)
    cs = _arr(cell_start, np.int32)
    w = _arr(window, np.int32, (4,))
    if bound:
        return int(L.plslam_grid_pair_capacity_bound(cen.shape[0], cen.shape[1], _p(cs), int(cols), int(rows), _p(w),
                                                     int(bool(mutual))))
    return int(L.plslam_grid_pair_capacity(_p(cen), cen.shape[0], cen.shape[1], _p(cs), int(cols), int(rows), _p(w),
                                           int(bool(mutual))))


def make_cam(fx, fy, cx, cy, b=0.0, width=0, height=0) -> Cam:
    return Cam(float(fx), float(fy), float(cx), float(cy), float(b), int(width), int(height))


class Context:
    """plslam_ctx: one HIP device, one stream, scratch pools."""

    def __init__(self, device: int = 0):
        self._L = load()
        h = C.c_void_p()
        _check(self._L.plslam_ctx_create(int(device), C.byref(h)), "plslam_ctx_create")
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.plslam_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    def set_option(self, key: str, value: int) -> None:
        _check(self._L.plslam_ctx_set_option(self._h, key.encode(), int(value)), "plslam_ctx_set_option")

    def get_option(self, key: str) -> int:
        v = C.c_int()
        _check(self._L.plslam_ctx_get_option(self._h, key.encode(), C.byref(v)), "plslam_ctx_get_option")
        return v.value

    def device_info(self) -> dict:
        cu, clk, lds = C.c_int32(), C.c_int32(), C.c_int32()
        name = C.create_string_buffer(256)
        _check(self._L.plslam_ctx_device_info(self._h, C.byref(cu), C.byref(clk), C.byref(lds), name, 256),
               "plslam_ctx_device_info")
        return {"cu_count": cu.value, "clock_khz": clk.value, "lds_bytes": lds.value,
                "name": name.value.decode()}

    # ---- host-pointer calls (numpy in, numpy out) ------------------------------------------
    def knn2(self, q, t):
        q = _arr(q, np.uint8, (-1, 32))
        t = _arr(t, np.uint8, (-1, 32))
        idx = np.empty((q.shape[0], 2), np.int32)
        dist = np.empty((q.shape[0], 2), np.int32)
        _check(self._L.plslam_knn2_hamming256(self._h, _p(q), q.shape[0], _p(t), t.shape[0], _p(idx),
                                              _p(dist)), "plslam_knn2_hamming256")
        return idx, dist

    def match(self, d1, d2, nnr: float, mutual: bool = True):
        """StVO::match(desc1, desc2, nnr, matches_12) -> (matches_12, n_matches)."""
        d1 = _arr(d1, np.uint8, (-1, 32))
        d2 = _arr(d2, np.uint8, (-1, 32))
        m12 = np.empty(d1.shape[0], np.int32)
        n = C.c_int32()
        _check(self._L.plslam_match(self._h, _p(d1), d1.shape[0], _p(d2), d2.shape[0], float(nnr),
                                    int(bool(mutual)), _p(m12), C.byref(n)), "plslam_match")
        return m12, n.value

    def match_prior(self, d1, d2, nnr: float, mutual: bool, prior):
        """StVO::match on a matches_12 that already holds entries (the fall-back after matchGrid) -> (matches_12, n)."""
        d1 = _arr(d1, np.uint8, (-1, 32))
        d2 = _arr(d2, np.uint8, (-1, 32))
        m12 = np.array(prior, np.int32, copy=True)
        if m12.shape != (d1.shape[0],):
            raise ValueError("prior must hold one entry per row of d1")
        n = C.c_int32()
        _check(self._L.plslam_match_prior(self._h, _p(d1), d1.shape[0], _p(d2), d2.shape[0], float(nnr),
                                          int(bool(mutual)), _p(m12), C.byref(n)), "plslam_match_prior")
        return m12, n.value

    def match_grid(self, centres, d1, cell_start, cell_items, cols, rows, d2, window, nnr, mutual=True, dir1=None,
                   dir2=None, sim_th=0.0):
        """StVO::matchGrid (points: one window centre per row; lines: two + directions) -> (matches_12, n)."""
        d1 = _arr(d1, np.uint8, (-1, 32))
        d2 = _arr(d2, np.uint8, (-1, 32))
        n1 = d1.shape[0]
        cen = _arr(centres, np.int32).reshape(n1, -1, 2) if n1 else np.zeros((0, 1, 2), np.int32)
        cs, items = _arr(cell_start, np.int32), _arr(cell_items, np.int32)
        w = _arr(window, np.int32, (4,))
        a = _arr(dir1, np.float64, (-1, 2)) if dir1 is not None else None
        b = _arr(dir2, np.float64, (-1, 2)) if dir2 is not None else None
        m12 = np.empty(n1, np.int32)
        n = C.c_int32()
        _check(self._L.plslam_match_grid(self._h, _p(cen), cen.shape[1], _p(d1), n1, _p(cs), _p(items), int(cols),
                                         int(rows), _p(d2), d2.shape[0], _p(a) if a is not None else None,
                                         _p(b) if b is not None else None, float(sim_th), _p(w), float(nnr),
                                         int(bool(mutual)), _p(m12), C.byref(n)), "plslam_match_grid")
        return m12, n.value

    def pose_gn_accumulate(self, cam, homog_th, T_inc, P, pl_obs, pt_inlier, sPeP, le_obs, ls_inlier):
        """computeRelativePoseGN iteration body -> (H[6,6], g[6], e, (N_p, N_l))."""
        T = _arr(T_inc, np.float64, (16,))
        P, po = _arr(P, np.float64, (-1, 3)), _arr(pl_obs, np.float64, (-1, 2))
        S, lo = _arr(sPeP, np.float64, (-1, 6)), _arr(le_obs, np.float64, (-1, 3))
        pi, li = _arr(pt_inlier, np.uint8), _arr(ls_inlier, np.uint8)
        H, g, e, n = np.empty((6, 6)), np.empty(6), np.empty(1), np.empty(2, np.int32)
        _check(self._L.plslam_pose_gn_accumulate(self._h, C.byref(cam), float(homog_th), _p(T), _p(P), _p(po), _p(pi),
                                                 P.shape[0], _p(S), _p(lo), _p(li), S.shape[0], _p(H), _p(g), _p(e), _p(n)),
               "plslam_pose_gn_accumulate")
        return H, g, float(e[0]), (int(n[0]), int(n[1]))

    def stereo_point_gate(self, m12, kp_l, kp_r, max_dist_epip, min_disp):
        """StereoFrame::matchStereoPoints gates -> (stereo_12, disp, n_stereo)."""
        m12 = _arr(m12, np.int32)
        a, b = _arr(kp_l, np.float32, (-1, 2)), _arr(kp_r, np.float32, (-1, 2))
        out, disp = np.empty(m12.shape[0], np.int32), np.empty(m12.shape[0], np.float64)
        n = C.c_int32()
        _check(self._L.plslam_stereo_point_gate(self._h, _p(m12), m12.shape[0], _p(a), _p(b), b.shape[0],
                                                float(max_dist_epip), float(min_disp), _p(out), _p(disp), C.byref(n)),
               "plslam_stereo_point_gate")
        return out, disp, n.value

    def stereo_line_gate(self, m12, seg_l, seg_r, min_disp, line_horiz_th, stereo_overlap_th, ls_min_disp_ratio):
        """StereoFrame::matchStereoLines gates -> (stereo_12, disp_se[n, 2], n_stereo)."""
        m12 = _arr(m12, np.int32)
        a, b = _arr(seg_l, np.float32, (-1, 4)), _arr(seg_r, np.float32, (-1, 4))
        out, disp = np.empty(m12.shape[0], np.int32), np.empty((m12.shape[0], 2), np.float64)
        n = C.c_int32()
        _check(self._L.plslam_stereo_line_gate(self._h, _p(m12), m12.shape[0], _p(a), _p(b), b.shape[0], float(min_disp),
                                               float(line_horiz_th), float(stereo_overlap_th), float(ls_min_disp_ratio),
                                               _p(out), _p(disp), C.byref(n)), "plslam_stereo_line_gate")
        return out, disp, n.value

    def match_batched(self, d1, off1, d2, off2, nnr: float, mutual: bool = True):
        d1 = _arr(d1, np.uint8, (-1, 32))
        d2 = _arr(d2, np.uint8, (-1, 32))
        off1 = _arr(off1, np.int32)
        off2 = _arr(off2, np.int32)
        B = off1.shape[0] - 1
        m12 = np.empty(int(off1[-1]) if B >= 0 else 0, np.int32)
        nm = np.empty(max(B, 0), np.int32)
        _check(self._L.plslam_match_batched(self._h, _p(d1), _p(off1), _p(d2), _p(off2), B, float(nnr),
                                            int(bool(mutual)), _p(m12), _p(nm)), "plslam_match_batched")
        return m12, nm

    def lba_point_rows(self, cam: Cam, homog_th, T_kf_w, Xw, obs_uv, lm_loc, kf_slot):
        T = _arr(T_kf_w, np.float64, (-1, 16))
        Xw = _arr(Xw, np.float64, (-1, 3))
        uv = _arr(obs_uv, np.float64, (-1, 2))
        lm, kf = _arr(lm_loc, np.int32), _arr(kf_slot, np.int32)
        n = uv.shape[0]
        Jp, Jl, r, w = np.empty((n, 6)), np.empty((n, 3)), np.empty(n), np.empty(n)
        _check(self._L.plslam_lba_point_rows(self._h, C.byref(cam), float(homog_th), _p(T), T.shape[0],
                                             _p(Xw), Xw.shape[0], _p(uv), _p(lm), _p(kf), n, _p(Jp),
                                             _p(Jl), _p(r), _p(w)), "plslam_lba_point_rows")
        return Jp, Jl, r, w

    def lba_line_rows(self, cam: Cam, homog_th, T_kf_w, Lw, l_obs, lm_loc, kf_slot,
                      compat_iter_pass: bool = False):
        T = _arr(T_kf_w, np.float64, (-1, 16))
        Lw = _arr(Lw, np.float64, (-1,))
        lo = _arr(l_obs, np.float64, (-1, 3))
        lm, kf = _arr(lm_loc, np.int32), _arr(kf_slot, np.int32)
        n = lo.shape[0]
        Jp, Jl, r, w = np.empty((n, 6)), np.empty((n, 6)), np.empty(n), np.empty(n)
        _check(self._L.plslam_lba_line_rows(self._h, C.byref(cam), float(homog_th),
                                            int(bool(compat_iter_pass)), _p(T), T.shape[0], _p(Lw),
                                            Lw.shape[0], _p(lo), _p(lm), _p(kf), n, _p(Jp), _p(Jl),
                                            _p(r), _p(w)), "plslam_lba_line_rows")
        return Jp, Jl, r, w

    def lba_assemble(self, nkf, npt, nls, pt_lm, pt_kf_loc, pt_rows, ls_lm, ls_kf_loc, ls_rows):
        """Block-form normal equations from point rows (Jp, Jl, r, w) and line rows."""
        plm, pkf = _arr(pt_lm, np.int32), _arr(pt_kf_loc, np.int32)
        llm, lkf = _arr(ls_lm, np.int32), _arr(ls_kf_loc, np.int32)
        pr = [_arr(x, np.float64) for x in pt_rows]
        lr = [_arr(x, np.float64) for x in ls_rows]
        npo, nlo = plm.shape[0], llm.shape[0]
        N = 6 * nkf + 3 * npt + 6 * nls
        g, Hp = np.empty(N), np.empty((nkf, 6, 6))
        Hpt, Hls = np.empty((npt, 3, 3)), np.empty((nls, 6, 6))
        Wp, Wl = np.empty((npo, 3, 6)), np.empty((nlo, 6, 6))
        err = np.empty(1)
        _check(self._L.plslam_lba_assemble(self._h, nkf, npt, nls, _p(plm), _p(pkf), npo, _p(pr[0]), _p(pr[1]),
                                           _p(pr[2]), _p(pr[3]), _p(llm), _p(lkf), nlo, _p(lr[0]), _p(lr[1]),
                                           _p(lr[2]), _p(lr[3]), _p(g), _p(Hp), _p(Hpt), _p(Hls), _p(Wp), _p(Wl),
                                           _p(err)), "plslam_lba_assemble")
        return dict(g=g, H_pose=Hp, H_pt=Hpt, H_ls=Hls, W_pt=Wp, W_ls=Wl, err=float(err[0]))

    def _gate(self, fn, name, cam, Twf, LM, lw, m12, feat, fw, th):
        Twf = _arr(Twf, np.float64, (16,))
        LM = _arr(LM, np.float64, (-1, lw))
        m12 = _arr(m12, np.int32)
        feat = _arr(feat, np.float64, (-1, fw))
        mask = np.empty(m12.shape[0], np.uint8)
        n = C.c_int32()
        _check(fn(self._h, C.byref(cam), _p(Twf), _p(LM), _p(m12), m12.shape[0], _p(feat), feat.shape[0],
                  float(th), _p(mask), C.byref(n)), name)
        return mask, n.value

    def map2kf_point_gate(self, cam, Twf, Xw, m12, pl, max_epip):
        return self._gate(self._L.plslam_map2kf_point_gate, "plslam_map2kf_point_gate", cam, Twf, Xw, 3,
                          m12, pl, 2, max_epip)

    def map2kf_line_gate(self, cam, Twf, Lw, m12, le, max_epip):
        return self._gate(self._L.plslam_map2kf_line_gate, "plslam_map2kf_line_gate", cam, Twf, Lw, 6,
                          m12, le, 3, max_epip)

    def map_point_visible(self, cam, Twf, Xw):
        Twf = _arr(Twf, np.float64, (16,))
        Xw = _arr(Xw, np.float64, (-1, 3))
        vis = np.empty(Xw.shape[0], np.uint8)
        _check(self._L.plslam_map_point_visible(self._h, C.byref(cam), _p(Twf), _p(Xw), Xw.shape[0], _p(vis)),
               "plslam_map_point_visible")
        return vis

    def map_line_visible(self, cam, Twf, Lw):
        Twf = _arr(Twf, np.float64, (16,))
        Lw = _arr(Lw, np.float64, (-1, 6))
        vis = np.empty(Lw.shape[0], np.uint8)
        _check(self._L.plslam_map_line_visible(self._h, C.byref(cam), _p(Twf), _p(Lw), Lw.shape[0], _p(vis)),
               "plslam_map_line_visible")
        return vis

    def median_desc_batched(self, desc_lists, offsets, want_desc=True):
        """MapPoint/MapLine::updateAverageDescDir (src/mapFeatures.cpp:51-84, :121-157) for all landmarks
        -> (med_idx[n_lm], med_desc[n_lm, 32] or None)."""
        d = _arr(desc_lists, np.uint8, (-1, 32))
        off = _arr(offsets, np.int32)
        n_lm = off.shape[0] - 1
        idx = np.empty(max(n_lm, 0), np.int32)
        md = np.empty((max(n_lm, 0), 32), np.uint8) if want_desc else None
        _check(self._L.plslam_median_desc_batched(self._h, _p(d), _p(off), n_lm, _p(idx),
                                                  _p(md) if want_desc else C.c_void_p(None)),
               "plslam_median_desc_batched")
        return idx, md

    def median_desc_batched_dev(self, d_desc_ptr, d_off_ptr, n_lm, total, d_idx_ptr, d_med_ptr=0, stream=None):
        """Device-pointer form; enqueues on `stream` (None = the context's stream), no sync."""
        _check(self._L.plslam_median_desc_batched_dev(self._h, C.c_void_p(d_desc_ptr), C.c_void_p(d_off_ptr),
                                                      int(n_lm), int(total), C.c_void_p(d_idx_ptr),
                                                      C.c_void_p(d_med_ptr or None), C.c_void_p(stream or 0)),
               "plslam_median_desc_batched_dev")

    def lbd_binarise(self, lbd_f32):
        """BinaryDescriptor::computeImpl's binary conversion (binary_descriptor_custom.cpp:653-668):
        n x 72 float32 LBD -> n x 32 uint8 rows."""
        f = _arr(lbd_f32, np.float32, (-1, 72))
        out = np.empty((f.shape[0], 32), np.uint8)
        _check(self._L.plslam_lbd_binarise(self._h, _p(f), f.shape[0], _p(out)), "plslam_lbd_binarise")
        return out

    def lbd_compute(self, dx_img, dy_img, lines, width_of_band=7):
        """BinaryDescriptor::computeLBD for the lines of one octave: (height, width) int16 gradient images and a
        structured array of plslam_lbd_line -> (n, 72) float32."""
        dx, dy = _arr(dx_img, np.int16), _arr(dy_img, np.int16)
        ln = np.ascontiguousarray(lines, dtype=LBD_LINE_DTYPE)
        out = np.empty((ln.shape[0], 72), np.float32)
        _check(self._L.plslam_lbd_compute(self._h, _p(dx), _p(dy), dx.shape[1], dx.shape[0], _p(ln), ln.shape[0],
                                          int(width_of_band), _p(out)), "plslam_lbd_compute")
        return out

    def lbd_compute_dev(self, d_dx_ptr, d_dy_ptr, width, height, lines, d_lbd_ptr, width_of_band=7, stream=None):
        """Device-pointer form (gradient images and the output on the device; the line records are host data)."""
        ln = np.ascontiguousarray(lines, dtype=LBD_LINE_DTYPE)
        _check(self._L.plslam_lbd_compute_dev(self._h, C.c_void_p(d_dx_ptr), C.c_void_p(d_dy_ptr), int(width), int(height),
                                              _p(ln), ln.shape[0], int(width_of_band), C.c_void_p(d_lbd_ptr),
                                              C.c_void_p(stream or 0)), "plslam_lbd_compute_dev")

    def lbd_binarise_dev(self, d_lbd_ptr, n, d_desc_ptr, stream=None):
        """Device-pointer form; enqueues on `stream` (None = the context's stream), no sync."""
        _check(self._L.plslam_lbd_binarise_dev(self._h, C.c_void_p(d_lbd_ptr), int(n),
                                               C.c_void_p(d_desc_ptr), C.c_void_p(stream or 0)),
               "plslam_lbd_binarise_dev")

    def map2kf_match(self, kind, cam, Twf, LM, med_desc, candidate, kf_desc, kf_feat, kf_idx, nnr, mutual,
                     max_epip, min_matches):
        """MapHandler::matchMap2KFPoints / matchMap2KFLines (compute part) -> (map_to_kf, n_matches)."""
        lw, fw = (3, 2) if kind == "points" else (6, 3)
        Twf = _arr(Twf, np.float64, (16,))
        LM = _arr(LM, np.float64, (-1, lw))
        md, cand = _arr(med_desc, np.uint8, (-1, 32)), _arr(candidate, np.uint8)
        kd, kf = _arr(kf_desc, np.uint8, (-1, 32)), _arr(kf_feat, np.float64, (-1, fw))
        ki = _arr(kf_idx, np.int32)
        out = np.empty(LM.shape[0], np.int32)
        n = C.c_int32()
        fn = self._L.plslam_map2kf_match_points if kind == "points" else self._L.plslam_map2kf_match_lines
        _check(fn(self._h, C.byref(cam), _p(Twf), _p(LM), _p(md), _p(cand), LM.shape[0], _p(kd), _p(kf), _p(ki),
                  kd.shape[0], float(nnr), int(bool(mutual)), float(max_epip), int(min_matches), _p(out),
                  C.byref(n)), "plslam_map2kf_match_" + kind)
        return out, n.value

    def map2kf_match_fast(self, kind, cam, Twf, LM, med_desc, candidate, kf_desc, kf_feat, kf_idx, nnr, mutual,
                          max_epip, min_matches, fm, kf_seg=None):
        """The drivers with SlamConfig::fastMatching() (matchGrid first, StVO::match when it finds too little)
        -> (map_to_kf, n_matches, used_match).  fm: dict with the fields of plslam_fast_matching."""
        lw, fw = (3, 2) if kind == "points" else (6, 3)
        Twf = _arr(Twf, np.float64, (16,))
        LM = _arr(LM, np.float64, (-1, lw))
        md, cand = _arr(med_desc, np.uint8, (-1, 32)), _arr(candidate, np.uint8)
        kd, kf = _arr(kf_desc, np.uint8, (-1, 32)), _arr(kf_feat, np.float64, (-1, fw))
        ki = _arr(kf_idx, np.int32)
        out = np.empty(LM.shape[0], np.int32)
        F = FastMatching(int(fm["enabled"]), int(fm["grid_cols"]), int(fm["grid_rows"]), int(fm["ws"]),
                         float(fm["inv_width"]), float(fm["inv_height"]), float(fm["nnr_grid"]),
                         float(fm.get("line_sim_th", 0.75)))
        n, used = C.c_int32(), C.c_int32()
        if kind == "points":
            _check(self._L.plslam_map2kf_match_points_fast(self._h, C.byref(cam), _p(Twf), _p(LM), _p(md), _p(cand),
                                                           LM.shape[0], _p(kd), _p(kf), _p(ki), kd.shape[0], float(nnr),
                                                           int(bool(mutual)), float(max_epip), int(min_matches),
                                                           C.byref(F), _p(out), C.byref(n), C.byref(used)),
                   "plslam_map2kf_match_points_fast")
        else:
            sg = _arr(kf_seg, np.float64, (-1, 4))
            _check(self._L.plslam_map2kf_match_lines_fast(self._h, C.byref(cam), _p(Twf), _p(LM), _p(md), _p(cand),
                                                          LM.shape[0], _p(kd), _p(kf), _p(sg), _p(ki), kd.shape[0],
                                                          float(nnr), int(bool(mutual)), float(max_epip),
                                                          int(min_matches), C.byref(F), _p(out), C.byref(n),
                                                          C.byref(used)), "plslam_map2kf_match_lines_fast")
        return out, n.value, used.value

    def map2kf_match_dev(self, kind, cam, Twf, d_LM, d_med_desc, d_candidate, n_map, kf_desc, kf_feat, kf_idx, nnr, mutual,
                         max_epip, min_matches, fm, kf_seg=None):
        """map2kf_match_fast with the MAP SIDE on the device: d_LM (n_map x 3 | 6 float64), d_med_desc (n_map x 32 uint8),
        d_candidate (n_map uint8) are device addresses (ints); the keyframe side and the results are host arrays."""
        fw = 2 if kind == "points" else 3
        Twf = _arr(Twf, np.float64, (16,))
        kd, kf = _arr(kf_desc, np.uint8, (-1, 32)), _arr(kf_feat, np.float64, (-1, fw))
        ki = _arr(kf_idx, np.int32)
        out = np.empty(int(n_map), np.int32)
        F = FastMatching(int(fm["enabled"]), int(fm["grid_cols"]), int(fm["grid_rows"]), int(fm["ws"]),
                         float(fm["inv_width"]), float(fm["inv_height"]), float(fm["nnr_grid"]),
                         float(fm.get("line_sim_th", 0.75)))
        n, used = C.c_int32(), C.c_int32()
        if kind == "points":
            _check(self._L.plslam_map2kf_match_points_dev(self._h, C.byref(cam), _p(Twf), int(d_LM), int(d_med_desc),
                                                          int(d_candidate), int(n_map), _p(kd), _p(kf), _p(ki), kd.shape[0],
                                                          float(nnr), int(bool(mutual)), float(max_epip), int(min_matches),
                                                          C.byref(F), _p(out), C.byref(n), C.byref(used)),
                   "plslam_map2kf_match_points_dev")
        else:
            sg = _arr(kf_seg, np.float64, (-1, 4))
            _check(self._L.plslam_map2kf_match_lines_dev(self._h, C.byref(cam), _p(Twf), int(d_LM), int(d_med_desc),
                                                         int(d_candidate), int(n_map), _p(kd), _p(kf), _p(sg), _p(ki),
                                                         kd.shape[0], float(nnr), int(bool(mutual)), float(max_epip),
                                                         int(min_matches), C.byref(F), _p(out), C.byref(n), C.byref(used)),
                   "plslam_map2kf_match_lines_dev")
        return out, n.value, used.value

    def kf2kf_match(self, kind, cam, DT, X_prev, desc_prev, feat_curr, desc_curr, nnr, mutual, min_matches, fm):
        """MapHandler::matchKF2KFPoints / Lines compute part -> (matches_12, n_matches, used_match)."""
        xw, fw = (3, 2) if kind == "points" else (6, 4)
        DT = _arr(DT, np.float64, (16,))
        X = _arr(X_prev, np.float64, (-1, xw))
        dp, dc = _arr(desc_prev, np.uint8, (-1, 32)), _arr(desc_curr, np.uint8, (-1, 32))
        fc = _arr(feat_curr, np.float64, (-1, fw))
        out = np.empty(X.shape[0], np.int32)
        F = FastMatching(int(fm["enabled"]), int(fm["grid_cols"]), int(fm["grid_rows"]), int(fm["ws"]),
                         float(fm["inv_width"]), float(fm["inv_height"]), float(fm["nnr_grid"]),
                         float(fm.get("line_sim_th", 0.75)))
        n, used = C.c_int32(), C.c_int32()
        fn = self._L.plslam_kf2kf_match_points if kind == "points" else self._L.plslam_kf2kf_match_lines
        _check(fn(self._h, C.byref(cam), _p(DT), _p(X), _p(dp), X.shape[0], _p(fc), _p(dc), fc.shape[0], float(nnr),
                  int(bool(mutual)), int(min_matches), C.byref(F), _p(out), C.byref(n), C.byref(used)),
               "plslam_kf2kf_match_" + kind)
        return out, n.value, used.value

    def kf2kf_match_dev(self, kind, cam, DT, d_X_prev, d_desc_prev, n_prev, feat_curr, d_desc_curr, nnr, mutual, min_matches, fm):
        """kf2kf_match with the ROWS on the device: d_X_prev (n_prev x 3 | 6 float64), d_desc_prev (n_prev x 32 uint8) and
        d_desc_curr (len(feat_curr) x 32 uint8) are device addresses (ints); feat_curr and the results are host arrays."""
        fw = 2 if kind == "points" else 4
        DT = _arr(DT, np.float64, (16,))
        fc = _arr(feat_curr, np.float64, (-1, fw))
        out = np.empty(int(n_prev), np.int32)
        F = FastMatching(int(fm["enabled"]), int(fm["grid_cols"]), int(fm["grid_rows"]), int(fm["ws"]),
                         float(fm["inv_width"]), float(fm["inv_height"]), float(fm["nnr_grid"]),
                         float(fm.get("line_sim_th", 0.75)))
        n, used = C.c_int32(), C.c_int32()
        fn = self._L.plslam_kf2kf_match_points_dev if kind == "points" else self._L.plslam_kf2kf_match_lines_dev
        _check(fn(self._h, C.byref(cam), _p(DT), int(d_X_prev), int(d_desc_prev), int(n_prev), _p(fc), int(d_desc_curr),
                  fc.shape[0], float(nnr), int(bool(mutual)), int(min_matches), C.byref(F), _p(out), C.byref(n), C.byref(used)),
               "plslam_kf2kf_match_" + kind + "_dev")
        return out, n.value, used.value

    # ---- device-pointer calls ----------------------------------------------------------------
    def lba_point_rows_dev(self, cam, homog_th, T, Xw, uv, lm, kf, nobs, Jp, Jl, r, w, stream=0, n_pose_slots=0):
        """n_pose_slots: how many 4 x 4 matrices T holds (0 = not stated: the kernels gather them from global memory)."""
        _check(self._L.plslam_lba_point_rows_dev_n(self._h, C.byref(cam), float(homog_th), T, int(n_pose_slots), Xw, uv, lm, kf,
                                                   int(nobs), Jp, Jl, r, w, stream or None),
               "plslam_lba_point_rows_dev_n")

    def lba_line_rows_dev(self, cam, homog_th, compat, T, Lw, lo, lm, kf, nobs, Jp, Jl, r, w, stream=0, n_pose_slots=0):
        _check(self._L.plslam_lba_line_rows_dev_n(self._h, C.byref(cam), float(homog_th), int(bool(compat)), T, int(n_pose_slots),
                                                  Lw, lo, lm, kf, int(nobs), Jp, Jl, r, w, stream or None),
               "plslam_lba_line_rows_dev_n")

    def plan(self, problems) -> "MatchPlan":
        return MatchPlan(self, problems)


class PinnedArray:
    """A numpy view of page-locked host memory (plslam_pinned_alloc): what a host-to-host pipeline should be fed from."""

    def __init__(self, ctx: "Context", shape, dtype):
        self._L = ctx._L
        self.nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self._ptr = self._L.plslam_pinned_alloc(max(self.nbytes, 1))
        if not self._ptr:
            raise PlslamError("plslam_pinned_alloc failed")
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(self.nbytes, 1)).from_address(self._ptr))[:self.nbytes] \
            .view(dtype).reshape(shape)

    def close(self):
        if getattr(self, "_ptr", None):
            self.array = None
            self._L.plslam_pinned_free(self._ptr)
            self._ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatchPipeline:
    """plslam_match_pipeline: a batch shape fixed at creation, `depth` batches in flight (upload / kernels / download on
    three streams).  problems: iterable of (d1_off, n1, d2_off, n2, nnr, mutual, out_off)."""

    def __init__(self, ctx: "Context", arena_bytes: int, problems, out_entries: int, depth: int = 3):
        self._L = ctx._L
        problems = list(problems)
        arr = (ArenaProblem * max(len(problems), 1))()
        for i, (o1, n1, o2, n2, nnr, mutual, oo) in enumerate(problems):
            arr[i] = ArenaProblem(int(o1), int(o2), int(n1), int(n2), float(nnr), int(bool(mutual)), int(oo))
        h = C.c_void_p()
        _check(self._L.plslam_match_pipeline_create(ctx.handle, int(arena_bytes), arr, len(problems), int(out_entries),
                                                    int(depth), C.byref(h)), "plslam_match_pipeline_create")
        self._h, self.nprob, self.arena_bytes, self.out_entries = h, len(problems), int(arena_bytes), int(out_entries)

    def submit(self, arena: np.ndarray, out: np.ndarray, counts: np.ndarray | None = None):
        assert arena.nbytes >= self.arena_bytes and out.dtype == np.int32 and out.size >= self.out_entries
        assert arena.flags.c_contiguous and out.flags.c_contiguous
        _check(self._L.plslam_match_pipeline_submit(self._h, arena.ctypes.data, out.ctypes.data,
                                                    counts.ctypes.data if counts is not None else None),
               "plslam_match_pipeline_submit")

    def wait(self):
        _check(self._L.plslam_match_pipeline_wait(self._h), "plslam_match_pipeline_wait")

    def close(self):
        if getattr(self, "_h", None):
            self._L.plslam_match_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LbaPlan:
    """plslam_lba_plan: observation lists uploaded once, iterate(T_kf_w, Xw, Lw) per LM iteration."""

    def __init__(self, ctx: Context, cam: Cam, homog_th, n_pose_slots, nkf, npt, nls, pt_lm, pt_slot, pt_kf_loc,
                 pt_obs_uv, ls_lm, ls_slot, ls_kf_loc, ls_l_obs):
        self._L = ctx._L
        self._ctx = ctx
        a = [_arr(x, np.int32) for x in (pt_lm, pt_slot, pt_kf_loc)]
        b = [_arr(x, np.int32) for x in (ls_lm, ls_slot, ls_kf_loc)]
        uv, lo = _arr(pt_obs_uv, np.float64, (-1, 2)), _arr(ls_l_obs, np.float64, (-1, 3))
        self.dims = (int(nkf), int(npt), int(nls), uv.shape[0], lo.shape[0], int(n_pose_slots))
        h = C.c_void_p()
        _check(self._L.plslam_lba_plan_create(ctx.handle, C.byref(cam), float(homog_th), int(n_pose_slots), int(nkf),
                                              int(npt), int(nls), _p(a[0]), _p(a[1]), _p(a[2]), _p(uv), uv.shape[0],
                                              _p(b[0]), _p(b[1]), _p(b[2]), _p(lo), lo.shape[0], C.byref(h)),
               "plslam_lba_plan_create")
        self._h = h

    COMPAT_ITER_PASS, COMPAT_GBA = 1, 2

    def iterate_dev(self, T_kf_w, Xw, Lw, compat_flags=0, want_g=True, g_out=None):
        """One iteration with the blocks left on the device -> (err, g or None).  With the arrays of host_state() as T_kf_w /
        Xw / Lw / g_out the call copies nothing on the host."""
        nkf, npt, nls, npo, nlo, nslot = self.dims
        T = _arr(T_kf_w, np.float64, (-1, 16))
        X, Lm = _arr(Xw, np.float64, (-1, 3)), _arr(Lw, np.float64, (-1, 6))
        if want_g and g_out is not None:
            assert g_out.dtype == np.float64 and g_out.flags.c_contiguous and g_out.size == 6 * nkf + 3 * npt + 6 * nls
            g = g_out
        else:
            g = np.empty(6 * nkf + 3 * npt + 6 * nls) if want_g else None
        err = np.empty(1)
        _check(self._L.plslam_lba_plan_iterate_dev(self._h, _p(T), _p(X), _p(Lm), int(compat_flags),
                                                   _p(g) if want_g else None, _p(err)), "plslam_lba_plan_iterate_dev")
        return float(err[0]), g

    def iterate_resident(self, compat_flags=0) -> float:
        """One iteration on the state already on the device (uploaded by an earlier iterate / iterate_dev, updated in place
        by a device-side solver: device_state()) -> err."""
        err = np.empty(1)
        _check(self._L.plslam_lba_plan_iterate_resident(self._h, int(compat_flags), _p(err)), "plslam_lba_plan_iterate_resident")
        return float(err[0])

    def diag_max(self) -> float:
        """max |H(i,i)| over the blocks of the last iteration (the reference's lambda *= Hmax)."""
        h = np.empty(1)
        _check(self._L.plslam_lba_plan_diag_max(self._h, _p(h)), "plslam_lba_plan_diag_max")
        return float(h[0])

    def schur(self, lam: float):
        """The reduced camera system of the last iteration's blocks for the damping `lam` -> (S (6 nkf, 6 nkf), b (6 nkf),
        number of singular landmark blocks)."""
        nkf = self.dims[0]
        S, b, ns = np.empty((6 * nkf, 6 * nkf)), np.empty(6 * nkf), np.zeros(1, np.int32)
        _check(self._L.plslam_lba_plan_schur(self._h, float(lam), _p(S), _p(b), _p(ns)), "plslam_lba_plan_schur")
        return S, b, int(ns[0])

    def backsub(self, dpose, apply=False, want=True):
        """The landmark steps for the pose step `dpose` -> (dX_pt (npt, 3), dX_ls (nls, 6)) or None; apply=True also adds them
        to the resident landmarks."""
        nkf, npt, nls = self.dims[:3]
        dp = _arr(dpose, np.float64, (6 * nkf,))
        dxp, dxl = (np.empty((npt, 3)), np.empty((nls, 6))) if want else (None, None)
        _check(self._L.plslam_lba_plan_backsub(self._h, _p(dp), int(bool(apply)), _p(dxp) if want else None,
                                               _p(dxl) if want else None), "plslam_lba_plan_backsub")
        return (dxp, dxl) if want else None

    def set_poses(self, T_kf_w) -> None:
        T = _arr(T_kf_w, np.float64, (-1, 16))
        assert T.shape[0] == self.dims[5]
        _check(self._L.plslam_lba_plan_set_poses(self._h, _p(T)), "plslam_lba_plan_set_poses")

    def iterate_schur(self, lam: float, compat_flags: int = 0):
        """iterate_resident + schur(lam) behind one synchronisation -> (err, S, b, number of singular landmark blocks)."""
        nkf = self.dims[0]
        err, S, b, ns = np.empty(1), np.empty((6 * nkf, 6 * nkf)), np.empty(6 * nkf), np.zeros(1, np.int32)
        _check(self._L.plslam_lba_plan_iterate_schur(self._h, int(compat_flags), float(lam), _p(err), _p(S), _p(b), _p(ns)),
               "plslam_lba_plan_iterate_schur")
        return float(err[0]), S, b, int(ns[0])

    def apply_step(self, dpose, T_kf_w=None, apply=True) -> float:
        """backsub(dpose, apply) + set_poses(T_kf_w, None: leave them) behind one synchronisation -> sum of squares of the
        landmark steps."""
        nkf = self.dims[0]
        dp = _arr(dpose, np.float64, (6 * nkf,))
        T = None
        if T_kf_w is not None:
            T = _arr(T_kf_w, np.float64, (-1, 16))
            assert T.shape[0] == self.dims[5]
        ss = np.empty(1)
        _check(self._L.plslam_lba_plan_apply_step(self._h, _p(dp), _p(T) if T is not None else None, int(bool(apply)), _p(ss)),
               "plslam_lba_plan_apply_step")
        return float(ss[0])

    def get_landmarks(self):
        """The resident landmarks (after backsub(apply=True)) -> (Xw (npt, 3), Lw (nls, 6)); also refreshes the page-locked images
        of host_state()."""
        nkf, npt, nls, npo, nlo, nslot = self.dims
        X, Lm = np.empty((npt, 3)), np.empty((nls, 6))
        _check(self._L.plslam_lba_plan_get_landmarks(self._h, _p(X), _p(Lm)), "plslam_lba_plan_get_landmarks")
        return X, Lm

    def host_state(self) -> dict:
        """The plan's page-locked images as numpy views (T_kf_w (n_pose_slots, 16), Xw (npt, 3), Lw (nls, 6), g (n,)): a
        solver that keeps its state in them passes them to iterate_dev(..., g_out=g) and nothing is staged.  The views die with
        the plan."""
        class _S(C.Structure):
            _fields_ = [("T_kf_w", C.c_void_p), ("Xw", C.c_void_p), ("Lw", C.c_void_p), ("g", C.c_void_p),
                        ("n_pose_slots", C.c_int32), ("npt", C.c_int32), ("nls", C.c_int32), ("n", C.c_int64)]
        st = _S()
        _check(self._L.plslam_lba_plan_host_state(self._h, C.byref(st)), "plslam_lba_plan_host_state")

        def view(ptr, shape):
            n = int(np.prod(shape))
            if n == 0:
                return np.empty(shape)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(n,)).reshape(shape)
        return dict(T_kf_w=view(st.T_kf_w, (st.n_pose_slots, 16)), Xw=view(st.Xw, (st.npt, 3)), Lw=view(st.Lw, (st.nls, 6)),
                    g=view(st.g, (st.n,)))

    def device_state(self) -> dict:
        """Device pointers (ints) of T_kf_w / Xw / Lw, their row counts and the plan's HIP stream."""
        class _S(C.Structure):
            _fields_ = [("T_kf_w", C.c_void_p), ("Xw", C.c_void_p), ("Lw", C.c_void_p), ("n_pose_slots", C.c_int32),
                        ("npt", C.c_int32), ("nls", C.c_int32), ("stream", C.c_void_p)]
        st = _S()
        _check(self._L.plslam_lba_plan_device_state(self._h, C.byref(st)), "plslam_lba_plan_device_state")
        return {k: getattr(st, k) for k, _ in _S._fields_}

    def blocks(self):
        """Download the blocks of the last iteration (the arrays plslam_lba_plan_device_blocks names)."""
        nkf, npt, nls, npo, nlo, _ = self.dims
        g, Hp = np.empty(6 * nkf + 3 * npt + 6 * nls), np.empty((nkf, 6, 6))
        Hpt, Hls = np.empty((npt, 3, 3)), np.empty((nls, 6, 6))
        Wp, Wl = np.empty((npo, 3, 6)), np.empty((nlo, 6, 6))
        err = np.empty(1)
        _check(self._L.plslam_lba_plan_blocks(self._h, _p(g), _p(Hp), _p(Hpt), _p(Hls), _p(Wp), _p(Wl), _p(err)),
               "plslam_lba_plan_blocks")
        return dict(g=g, H_pose=Hp, H_pt=Hpt, H_ls=Hls, W_pt=Wp, W_ls=Wl, err=float(err[0]))

    def iterate(self, T_kf_w, Xw, Lw, compat_iter_pass=False):
        nkf, npt, nls, npo, nlo, nslot = self.dims
        T = _arr(T_kf_w, np.float64, (-1, 16))
        X, Lm = _arr(Xw, np.float64, (-1, 3)), _arr(Lw, np.float64, (-1, 6))
        assert T.shape[0] == nslot and X.shape[0] == npt and Lm.shape[0] == nls
        N = 6 * nkf + 3 * npt + 6 * nls
        g, Hp = np.empty(N), np.empty((nkf, 6, 6))
        Hpt, Hls = np.empty((npt, 3, 3)), np.empty((nls, 6, 6))
        Wp, Wl = np.empty((npo, 3, 6)), np.empty((nlo, 6, 6))
        err = np.empty(1)
        _check(self._L.plslam_lba_plan_iterate(self._h, _p(T), _p(X), _p(Lm), int(compat_iter_pass), _p(g), _p(Hp),
                                               _p(Hpt), _p(Hls), _p(Wp), _p(Wl), _p(err)), "plslam_lba_plan_iterate")
        return dict(g=g, H_pose=Hp, H_pt=Hpt, H_ls=Hls, W_pt=Wp, W_ls=Wl, err=float(err[0]))

    def rows(self):
        nkf, npt, nls, npo, nlo, _ = self.dims
        pr = [np.empty((npo, 6)), np.empty((npo, 3)), np.empty(npo), np.empty(npo)]
        lr = [np.empty((nlo, 6)), np.empty((nlo, 6)), np.empty(nlo), np.empty(nlo)]
        _check(self._L.plslam_lba_plan_rows(self._h, *[_p(x) for x in pr + lr]), "plslam_lba_plan_rows")
        return pr, lr

    def close(self):
        if getattr(self, "_h", None):
            self._L.plslam_lba_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MatchPlan:
    """plslam_match_plan over device pointers (ints, e.g. torch.Tensor.data_ptr()).

    problems: iterable of (d1_ptr, n1, d2_ptr, n2, nnr, mutual, matches12_ptr, n_matches_ptr|0)
    """

    def __init__(self, ctx: Context, problems):
        self._ctx = ctx
        self._L = ctx._L
        problems = list(problems)
        arr = (MatchProblem * max(len(problems), 1))()
        for i, (d1, n1, d2, n2, nnr, mutual, m12, nm) in enumerate(problems):
            arr[i] = MatchProblem(d1 or None, d2 or None, int(n1), int(n2), float(nnr), int(bool(mutual)),
                                  m12 or None, nm or None)
        h = C.c_void_p()
        _check(self._L.plslam_match_plan_create(ctx.handle, arr, len(problems), C.byref(h)),
               "plslam_match_plan_create")
        self._h = h

    def run(self, stream: int = 0) -> None:
        _check(self._L.plslam_match_plan_run(self._h, stream or None), "plslam_match_plan_run")

    def run_split(self, scan_stream: int, post_stream: int) -> None:
        """The scan on one HIP stream, the stages behind it on another (they overlap the next run's scan)."""
        _check(self._L.plslam_match_plan_run_split(self._h, scan_stream or None, post_stream or None),
               "plslam_match_plan_run_split")

    def make_gather_step(self, comm: int, nranks: int, rank: int, root: int, wire_bytes: int, send: int, n_entries: int, recv: int,
                         wide: int, scan_stream: int, post_stream: int, comm_stream: int = 0):
        """The argument block of step_gather(), built ONCE per (plan, buffer): the step itself is then one foreign call."""
        return GatherStep(comm or None, int(nranks), int(rank), int(root), int(wire_bytes), send or None, int(n_entries), recv or None,
                          wide or None, scan_stream or None, post_stream or None, comm_stream or None)

    def step_gather(self, step) -> None:
        """plslam_match_plan_step_gather: plan run + gather of the finished table to the root (+ widening), all enqueued."""
        rc = self._L.plslam_match_plan_step_gather(self._h, C.byref(step))
        if rc:
            _check(rc, "plslam_match_plan_step_gather")

    def gather_sync(self) -> None:
        _check(self._L.plslam_match_plan_gather_sync(self._h), "plslam_match_plan_gather_sync")

    def add_stereo_gates(self, gates) -> None:
        """The gate stage (StereoFrame::matchStereoPoints / matchStereoLines over the batch).  gates: iterable of dicts
        with the fields of plslam_stereo_gate_problem (device pointers as ints)."""
        gates = list(gates)
        arr = (StereoGateProblem * max(len(gates), 1))()
        for i, g in enumerate(gates):
            arr[i] = StereoGateProblem(g["matches_12"] or None, g["f_l"] or None, g["f_r"] or None, int(g["n_l"]),
                                       int(g["n_r"]), int(bool(g["lines"])), 0, float(g.get("max_dist_epip", 0.0)),
                                       float(g.get("min_disp", 0.0)), float(g.get("line_horiz_th", 0.0)),
                                       float(g.get("stereo_overlap_th", 0.0)), float(g.get("ls_min_disp_ratio", 0.0)),
                                       g["stereo_12"] or None, g["disp"] or None, g.get("n_stereo") or None)
        _check(self._L.plslam_match_plan_add_stereo_gates(self._h, arr, len(gates)), "plslam_match_plan_add_stereo_gates")

    def set_wire16(self, table32: int, table16: int, n_entries: int) -> None:
        """An int16 mirror of the plan's match tables inside [table32, table32 + n_entries) (device pointers as ints), written by
        the finalize kernel beside the int32 entries: the wire format of the N > 1 gather.  table16 = 0 removes it."""
        _check(self._L.plslam_match_plan_set_wire16(self._h, table32 or None, table16 or None, int(n_entries)),
               "plslam_match_plan_set_wire16")

    def set_profiling(self, on: bool) -> None:
        _check(self._L.plslam_match_plan_set_profiling(self._h, int(bool(on))),
               "plslam_match_plan_set_profiling")

    def elapsed(self):
        a, b, n = C.c_double(), C.c_double(), C.c_int64()
        _check(self._L.plslam_match_plan_elapsed(self._h, C.byref(a), C.byref(b), C.byref(n)),
               "plslam_match_plan_elapsed")
        return a.value, b.value, n.value

    def info(self) -> dict:
        i = PlanInfo()
        _check(self._L.plslam_match_plan_info(self._h, C.byref(i)), "plslam_match_plan_info")
        return {k: getattr(i, k) for k, _ in PlanInfo._fields_}

    def key_state(self) -> int:
        """What dump()'s key table holds: a mask of KEYS_ROW_SECOND_INDEX_INEXACT (1), KEYS_COLUMN_SECOND_LAZY (2),
        KEYS_COLUMNS_NOT_IN_MEMORY (4); 0 = every word an exact key."""
        f = C.c_int32()
        _check(self._L.plslam_match_plan_key_state(self._h, C.byref(f)), "plslam_match_plan_key_state")
        return f.value

    def dump(self):
        """Diagnostics: (keys, column_partials) as uint32 arrays, after a device synchronise."""
        kb, pb = C.c_size_t(), C.c_size_t()
        _check(self._L.plslam_match_plan_dump(self._h, None, 0, None, 0, C.byref(kb), C.byref(pb)),
               "plslam_match_plan_dump")
        k = np.empty(kb.value // 4, np.uint32)
        p = np.empty(pb.value // 4, np.uint32)
        _check(self._L.plslam_match_plan_dump(self._h, k.ctypes.data, k.nbytes, p.ctypes.data, p.nbytes,
                                              C.byref(kb), C.byref(pb)), "plslam_match_plan_dump")
        return k, p

    def close(self):
        if getattr(self, "_h", None):
            self._L.plslam_match_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GridPlan:
    """plslam_grid_plan over device pointers: a batch of StVO::matchGrid problems, one kernel launch.

    problems: iterable of dicts with the fields of plslam_grid_problem (pointers as ints; window a 4-sequence).
    """

    def __init__(self, ctx: Context, problems):
        self._ctx = ctx
        self._L = ctx._L
        problems = list(problems)
        arr = (GridProblem * max(len(problems), 1))()
        for i, q in enumerate(problems):
            g = GridProblem()
            for k in ("d1", "d2", "centres1", "cell_start", "cell_items", "dir1", "dir2", "matches_12", "n_matches"):
                setattr(g, k, q.get(k) or None)
            for k in ("n1", "n2", "n_centres", "grid_cols", "grid_rows", "n_items", "pair_capacity"):
                setattr(g, k, int(q[k]))
            g.window = (C.c_int32 * 4)(*[int(v) for v in q["window"]])
            g.sim_th, g.nnr, g.mutual = float(q.get("sim_th", 0.0)), float(q["nnr"]), int(bool(q.get("mutual", True)))
            arr[i] = g
        h = C.c_void_p()
        _check(self._L.plslam_grid_plan_create(ctx.handle, arr, len(problems), C.byref(h)), "plslam_grid_plan_create")
        self._h = h

    def run(self, stream: int = 0) -> None:
        _check(self._L.plslam_grid_plan_run(self._h, stream or None), "plslam_grid_plan_run")

    def overflows(self, stream: int = 0) -> int:
        n = C.c_int32()
        _check(self._L.plslam_grid_plan_overflows(self._h, stream or None, C.byref(n)), "plslam_grid_plan_overflows")
        return n.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.plslam_grid_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _hip_memcpy_dtod(dst: int, src: int, nbytes: int) -> int:
    """Device-to-device copy through the HIP runtime (tests: an in-place update of a plan's device state)."""
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpy.restype = C.c_int
    return hip.hipMemcpy(C.c_void_p(dst), C.c_void_p(src), C.c_size_t(nbytes), 3)      # hipMemcpyDeviceToDevice
