"""oracle/ref_extract_lba.py -- TEST INFRASTRUCTURE ONLY (run by oracle/Makefile, writes only into oracle/_ref/).

Cuts the four observation loops of MapHandler::levMarquardtOptimizationLBA out of the reference source WHERE IT LIES
(/root/reference/src/mapHandler.cpp; nothing is copied into the repository: oracle/_ref/ is git-ignored) so that
oracle/ref_wrap_lba.cpp can compile them textually against the dense-matrix stand-in oracle/ref_shim/mini_dense.hpp:
    lba_pt_first.inc   first pass,     point observations  (the `for (... pt_it ...)` loop that starts near :1358)
    lba_ls_first.inc   first pass,     line observations                                            (near :1436)
    lba_pt_iter.inc    iteration pass, point observations                                           (near :1587)
    lba_ls_iter.inc    iteration pass, line observations                                            (near :1668)
The loops are found by their headers inside that function and closed by brace matching, not by line number.
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


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
