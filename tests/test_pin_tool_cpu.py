"""tools/pin_with_opencv.py replays the committed goldens through a real cv::BFMatcher when one is importable.  Here (no
OpenCV) its replay logic is exercised with a stand-in `cv2` whose knnMatch is the oracle's kNN-2: the tool must then find
no difference -- and must find them when the stand-in breaks ties the other way."""
import importlib.util
import os
import types

import numpy as np


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
