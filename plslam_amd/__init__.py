"""plslam_amd -- MI355X (gfx950) implementation of PL-SLAM's stereo point+line matching front end
(256-bit ORB / LBD brute-force Hamming kNN-2 + ratio + mutual, i.e. StVO::match as called from
src/mapHandler.cpp:277,424,597,712,3223,3249) and of the local-BA residual/Jacobian row build
(src/mapHandler.cpp:1358-1540).

The product is the C-ABI library ``plslam_amd/lib/libplslam_hip.so`` (include/plslam_hip.h,
sources in plslam_amd/csrc).  The Python in this package is plumbing around it (ctypes binding,
device-memory handling through torch, sharding over ranks); the C++ host shim that mirrors
the reference's ``StVO::match`` signature lives in plslam_amd/host.
"""
from .capi import (Cam, Context, MatchPlan, MatchPipeline, PinnedArray, LbaPlan, GridPlan, PlslamError, LIB_PATH, ABI_SYMBOLS, make_cam, load,
                   SCAN_AUTO, SCAN_LANE_PER_QUERY, SCAN_WAVE_PER_QUERY, SCAN_SYMMETRIC, SCAN_MFMA)

__all__ = ["Cam", "Context", "MatchPlan", "MatchPipeline", "PinnedArray", "LbaPlan", "GridPlan", "PlslamError", "LIB_PATH", "ABI_SYMBOLS", "make_cam", "load",
           "SCAN_AUTO", "SCAN_LANE_PER_QUERY", "SCAN_WAVE_PER_QUERY", "SCAN_SYMMETRIC", "SCAN_MFMA"]
