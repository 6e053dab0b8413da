"""oracle/ref_extract_lba.py -- TEST INFRASTRUCTURE ONLY (run by oracle/Makefile, writes only into oracle/_ref/).

Cuts the four observation loops of MapHandler::levMarquardtOptimizationLBA out of the reference source WHERE IT LIES
(/root/reference/src/mapHandler.cpp; nothing is copied into the repository: oracle/_ref/ is git-ignored) so that
oracle/ref_wrap_lba.cpp can compile them textually against the dense-matrix stand-in oracle/ref_shim/mini_dense.hpp:
    lba_pt_first.inc   first pass,     point observations  (the `for (... pt_it ...)` loop that starts near :1358)
    lba_ls_first.inc   first pass,     line observations                                            (near :1436)
    lba_pt_iter.inc    iteration pass, point observations                                           (near :1587)
    lba_ls_iter.inc    iteration pass, line observations                                            (near :1668)
    gba_pt_first.inc, gba_ls_first.inc   the first-pass loops of levMarquardtOptimizationGBA        (near :2124, :2233)
and the point / line loops of the pose-only Gauss-Newton iterations (K17's checker):
    gn_pt.inc, gn_ls.inc           MapHandler::computeRelativePoseGN        (near :3330, :3370)
    gnr_pt.inc, gnr_ls.inc         MapHandler::computeRelativePoseRobustGN  (first pair of loops, near :3594, :3634)
and the whole LM body of levMarquardtOptimizationLBA (everything between its opening brace and its write-back section):
    lba_lm_body.inc                (near :1334-1812)
and the visibility pre-filter / geometric gate loops of the map <-> key-frame matchers:
    m2kf_pt_vis.inc, m2kf_pt_gate.inc   MapHandler::matchMap2KFPoints  (near :545, :602)
    m2kf_ls_vis.inc, m2kf_ls_gate.inc   MapHandler::matchMap2KFLines   (near :646, :715)
The loops are found by their headers inside those functions and closed by brace matching, not by line number.
usage: ref_extract_lba.py <reference root> <output dir>
"""
import os
import sys


def loop_at(lines, start):
    depth, seen = 0, False
    for k in range(start, len(lines)):
        for ch in lines[k]:
            if ch == "{":
                depth += 1
                seen = True
            elif ch == "}":
                depth -= 1
        if seen and depth == 0:
            return k
    raise SystemExit("unbalanced braces")


def main(ref, out):
    src = open(os.path.join(ref, "src", "mapHandler.cpp"), encoding="utf-8", errors="replace").read().split("\n")
    f0 = next(i for i, l in enumerate(src) if "MapHandler::levMarquardtOptimizationLBA" in l)
    f1 = next(i for i, l in enumerate(src) if i > f0 and "MapHandler::globalBundleAdjustment" in l)
    found = {"pt": [], "ls": []}
    for i in range(f0, f1):
        for key in found:
            if "vector<Vector6i>::iterator %s_it" % key in src[i] and src[i].lstrip().startswith("for"):
                found[key].append(i)
    if len(found["pt"]) != 2 or len(found["ls"]) != 2:
        raise SystemExit("expected two point and two line loops, found %r" % found)
    os.makedirs(out, exist_ok=True)
    for key in ("pt", "ls"):
        for which, start in zip(("first", "iter"), found[key]):
            end = loop_at(src, start)
            name = "lba_%s_%s.inc" % (key, which)
            with open(os.path.join(out, name), "w") as fh:
                fh.write("// generated from src/mapHandler.cpp:%d-%d by oracle/ref_extract_lba.py -- not part of the repository\n"
                         % (start + 1, end + 1))
                fh.write("\n".join(src[start:end + 1]) + "\n")
            print("[ref_extract_lba] %s = src/mapHandler.cpp:%d-%d" % (name, start + 1, end + 1))
    # the WHOLE body of levMarquardtOptimizationLBA up to its write-back section (:1334-1812: variables, first pass, lambda,
    # first solve and update, the LM iterations with their lambda schedule and stop tests) for oracle/ref_wrap_lba_lm.cpp
    b0 = next(i for i in range(f0, f1) if src[i].strip() == "{") + 1
    b1 = next(i for i in range(b0, f1) if "vo_status != VO_INSERTING_KF" in src[i])
    with open(os.path.join(out, "lba_lm_body.inc"), "w") as fh:
        fh.write("// generated from src/mapHandler.cpp:%d-%d by oracle/ref_extract_lba.py -- not part of the repository\n" % (b0 + 1, b1))
        fh.write("\n".join(src[b0:b1]) + "\n")
    print("[ref_extract_lba] lba_lm_body.inc = src/mapHandler.cpp:%d-%d" % (b0 + 1, b1))
    f2 = next(i for i, l in enumerate(src) if "MapHandler::levMarquardtOptimizationGBA" in l)
    f3 = next(i for i, l in enumerate(src) if i > f2 and "MapHandler::removeBadMapLandmarks" in l)
    for key in ("pt", "ls"):
        start = next(i for i in range(f2, f3) if "vector<Vector6i>::iterator %s_it" % key in src[i] and src[i].lstrip().startswith("for"))
        end = loop_at(src, start)
        name = "gba_%s_first.inc" % key
        with open(os.path.join(out, name), "w") as fh:
            fh.write("// generated from src/mapHandler.cpp:%d-%d by oracle/ref_extract_lba.py -- not part of the repository\n"
                     % (start + 1, end + 1))
            fh.write("\n".join(src[start:end + 1]) + "\n")
        print("[ref_extract_lba] %s = src/mapHandler.cpp:%d-%d" % (name, start + 1, end + 1))
    for fn, nxt, tag, vis_hdr in (("MapHandler::matchMap2KFPoints", "MapHandler::matchMap2KFLines", "m2kf_pt", "for (MapPoint* pt : map_points)"),
                                  ("MapHandler::matchMap2KFLines", "MapHandler::lookForCommonMatches", "m2kf_ls", "for (MapLine* ls : map_lines)")):
        g0 = next(i for i, l in enumerate(src) if fn + "(" in l.replace(" ", ""))
        g1 = next(i for i, l in enumerate(src) if i > g0 and nxt in l)
        for key, hdr in (("vis", vis_hdr), ("gate", "for (int i1 = 0; i1 < matches_12.size(); ++i1)")):
            start = next(i for i in range(g0, g1) if hdr in src[i])
            end = loop_at(src, start)
            name = "%s_%s.inc" % (tag, key)
            with open(os.path.join(out, name), "w") as fh:
                fh.write("// generated from src/mapHandler.cpp:%d-%d by oracle/ref_extract_lba.py -- not part of the repository\n"
                         % (start + 1, end + 1))
                fh.write("\n".join(src[start:end + 1]) + "\n")
            print("[ref_extract_lba] %s = src/mapHandler.cpp:%d-%d" % (name, start + 1, end + 1))
    for fn, nxt, tag in (("MapHandler::computeRelativePoseGN", "MapHandler::computeRelativePoseRobustGN", "gn"),
                         ("MapHandler::computeRelativePoseRobustGN", "MapHandler::loopClosureOptimizationEssGraphG2O", "gnr")):
        g0 = next(i for i, l in enumerate(src) if fn + "(" in l.replace(" ", ""))
        g1 = next(i for i, l in enumerate(src) if i > g0 and nxt in l)
        for key, hdr in (("pt", "vector<PointFeature*>::iterator pt_it"), ("ls", "vector<LineFeature*>::iterator ls_it")):
            start = next(i for i in range(g0, g1) if hdr in src[i] and src[i].lstrip().startswith("for"))
            end = loop_at(src, start)
            name = "%s_%s.inc" % (tag, key)
            with open(os.path.join(out, name), "w") as fh:
                fh.write("// generated from src/mapHandler.cpp:%d-%d by oracle/ref_extract_lba.py -- not part of the repository\n"
                         % (start + 1, end + 1))
                fh.write("\n".join(src[start:end + 1]) + "\n")
            print("[ref_extract_lba] %s = src/mapHandler.cpp:%d-%d" % (name, start + 1, end + 1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
