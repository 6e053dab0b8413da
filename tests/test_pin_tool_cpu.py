"""tools/pin_with_opencv.py replays the committed goldens through a real cv::BFMatcher when one is importable.  Here (no
OpenCV) its replay logic is exercised with a stand-in `cv2` whose knnMatch is the oracle's kNN-2: the tool must then find
no difference -- and must find them when the stand-in breaks ties the other way."""
import importlib.util
import os
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pin_with_opencv", os.path.join(root, "tools", "pin_with_opencv.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fake_cv2(oracle, flip_ties=False):
    class DMatch:
        def __init__(self, i, d):
            self.trainIdx, self.distance = int(i), float(d)

    class BF:
        def knnMatch(self, q, t, k=2):
            idx, dist = oracle.knn2(q, t)
            out = []
            for i in range(len(q)):
                row = [DMatch(idx[i, c], dist[i, c]) for c in range(2) if idx[i, c] >= 0]
                if flip_ties and len(row) == 2 and row[0].distance == row[1].distance:
                    row.reverse()
                out.append(row)
            return out
    m = types.SimpleNamespace(NORM_HAMMING=6, __version__="stand-in")
    m.BFMatcher = lambda norm, crossCheck=False: BF()
    return m


def test_replay_logic_reproduces_the_goldens(oracle):
    n, bad = _tool().replay_match_golden(_fake_cv2(oracle))
    assert n >= 5 and bad == []


def test_replay_reports_a_different_tie_order(oracle):
    _, bad = _tool().replay_match_golden(_fake_cv2(oracle, flip_ties=True))
    assert any("tie order" in b[1] for b in bad)


def _export(tmp_path):
    import importlib.util
    spec = importlib.util.spec_from_file_location("export_cases", os.path.join(ROOT, "tools", "pin_stvo", "export_cases.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = str(tmp_path / "cases")
    mod.main(out)
    return mod, out


def test_stvo_harness_export_round_trip(tmp_path):
    """tools/pin_stvo/export_cases.py: every exported array reads back as the golden it came from."""
    mod, out = _export(tmp_path)
    g = np.load(os.path.join(ROOT, "tests", "golden", "grid_golden.npz"))
    assert np.array_equal(mod.read_array(os.path.join(out, "grid_c2_dir2.bin")).view(np.uint64), g["c2_dir2"].view(np.uint64))   # (NaN rows)
    assert np.array_equal(mod.read_array(os.path.join(out, "grid_c0_m1_r75.bin")), g["c0_m1_r75"])
    m = np.load(os.path.join(ROOT, "tests", "golden", "match_golden.npz"))
    assert np.array_equal(mod.read_array(os.path.join(out, "match_planted_q.bin")), m["planted/q"])
    s = np.load(os.path.join(ROOT, "tests", "golden", "stereo_gates_golden.npz"))
    assert np.array_equal(mod.read_array(os.path.join(out, "gate_l0_t0_d.bin")).view(np.uint32), s["l0_t0_disp"].view(np.uint32))
    kinds = [ln.split()[0] for ln in open(os.path.join(out, "manifest.txt"))]
    assert kinds.count("kind=match") >= 12 and kinds.count("kind=grid_lines") == 8 and kinds.count("kind=gate_lines") == 9
    assert kinds.count("kind=se3") == 1


def test_stvo_harness_runs_against_stand_ins(tmp_path):
    """The C++ harness (tools/pin_stvo/pin_stvo.cpp) built against stand-in stvo-pl / OpenCV headers whose StVO:: functions
    forward to the CPU restatement: every replayable case passes, a corrupted expectation is reported.  This checks the
    HARNESS (file format, grid reconstruction, window / ratio plumbing) -- the day a stvo-pl checkout exists the same binary,
    linked against the real sources, checks the restatement."""
    import subprocess
    _, out = _export(tmp_path)
    exe = str(tmp_path / "pin_stvo_selftest")
    standin = os.path.join(ROOT, "tests", "cpp", "stvo_standin")
    cmd = ["g++", "-O1", "-std=c++14", "-DPIN_STANDIN_SE3", "-I" + standin, "-I" + os.path.join(ROOT, "oracle"),
           os.path.join(ROOT, "tools", "pin_stvo", "pin_stvo.cpp"), os.path.join(standin, "standin.cpp"), "-o", exe,
           "-L" + os.path.join(ROOT, "oracle"), "-lplslam_oracle", "-Wl,-rpath," + os.path.join(ROOT, "oracle")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "103 cases replayed, 0 differ; 18 stereo-gate cases listed" in r.stdout
    # ONE table at the end, a row per golden file
    table = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.split() and ln.split()[0].endswith("_golden.npz")}
    assert table["se3_helpers_golden.npz"][:3] == ["5", "5", "0"] and table["stereo_gates_golden.npz"] == ["0", "0", "0", "18"]
    assert table["match_golden.npz"][2] == "0" and table["grid_golden.npz"][2] == "0"
    assert "PASS se3 inverse_se3" in r.stdout and "PASS se3 robustWeightCauchy" in r.stdout
    # corrupt one expectation: the harness must say which table differs
    p = os.path.join(out, "grid_c0_m1_r75.bin")
    raw = bytearray(open(p, "rb").read())
    raw[-4:] = (12345).to_bytes(4, "little")
    open(p, "wb").write(bytes(raw))
    r = subprocess.run([exe, out], capture_output=True, text=True)
    assert r.returncode == 1 and "FAIL grid_points c0 nnr 0.75 mutual 1" in r.stdout and "1 differ" in r.stdout
    # ... and a helper that follows another convention (here: one corrupted element of the expected inverse poses)
    p = os.path.join(out, "se3_inverse.bin")
    raw = bytearray(open(p, "rb").read())
    raw[-8 * 16:-8 * 15] = np.float64(0.25).tobytes()
    open(p, "wb").write(bytes(raw))
    r = subprocess.run([exe, out], capture_output=True, text=True)
    assert r.returncode == 1 and "FAIL se3 inverse_se3" in r.stdout and "PASS se3 expmap_se3" in r.stdout


def test_se3_helper_golden_is_the_oracle(oracle):
    """tests/golden/se3_helpers_golden.npz (make_se3_golden.py) against the oracle it came from, bit for bit, and against an
    independent numpy statement of the same helpers (Rodrigues, [R^T, -R^T t], the pinhole formula) to 1e-12."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "se3_helpers_golden.npz"))
    for x, T, Ti, lg in zip(g["twists"], g["expmap"], g["inverse"], g["logmap"]):
        assert np.array_equal(oracle.expmap_se3(x), T) and np.array_equal(oracle.inverse_se3(T), Ti)
        assert np.array_equal(oracle.logmap_se3(T), lg)
        R, t = T[:3, :3], T[:3, 3]
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1) < 1e-12
        ref = np.eye(4); ref[:3, :3] = R.T; ref[:3, 3] = -R.T @ t
        assert np.allclose(Ti, ref, rtol=0, atol=1e-12 * max(1.0, np.abs(t).max()))
        th = np.linalg.norm(x[3:])
        if th >= 1e-6:                                   # Rodrigues
            K = np.array([[0, -x[5], x[4]], [x[5], 0, -x[3]], [-x[4], x[3], 0]]) / th
            assert np.allclose(R, np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K, atol=1e-12)
        else:
            assert np.array_equal(R, np.eye(3)) and np.array_equal(t, x[:3])
        if 1e-6 <= th < 3.0:                             # the log of the exp is the twist (away from pi)
            assert np.allclose(lg, x, atol=1e-8)
        elif th < 1e-6:                                  # below the small-angle switch the rotation is dropped
            assert np.array_equal(lg[:3], x[:3]) and not lg[3:].any()
    fx, fy, cx, cy = g["cam"]
    with np.errstate(divide="ignore", invalid="ignore"):
        uv = np.stack([cx + fx * g["points"][:, 0] / g["points"][:, 2], cy + fy * g["points"][:, 1] / g["points"][:, 2]], 1)
    assert np.array_equal(uv.view(np.uint64), g["projection"].view(np.uint64))
    with np.errstate(over="ignore"):
        assert np.array_equal(1.0 / (1.0 + g["cauchy_r"] ** 2), g["cauchy_w"]) and g["cauchy_w"][5] == 0.0
