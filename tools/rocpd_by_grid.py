#!/usr/bin/env python3
"""Per-(kernel, grid size) averages of a rocprofv3 rocpd database: duration and every collected counter.  The plain summary
(tools/rocpd_summary.py) averages over all launches of a kernel; this one keeps launches of different sizes apart (e.g. the
128- and the 4096-problem launches of tools/grid_scaling.py).   usage: rocpd_by_grid.py <run_results.db> [kernel substring]"""
import sqlite3
import sys


def main(path, sub=""):
    c = sqlite3.connect(path)
    views = [r[0] for r in c.execute("select name from sqlite_master where type in ('view','table')")]
    kcols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("# kernels columns:", kcols)
    name = "name" if "name" in kcols else "kernel_name"
    gcol = next((x for x in ("grid_size_x", "grid_x", "grid_size", "workgroup_count_x") if x in kcols), None)
    if gcol:
        for n, g, cnt, avg in c.execute(f"select {name}, {gcol}, count(*), avg(end-start) from kernels where {name} like ? "
                                        f"group by {name}, {gcol} order by 1, 2", (f"%{sub}%",)):
            print(f"{n[:60]:60s} grid {g:>9} calls {cnt:4d} avg_us {avg / 1e3:10.2f}")
    if "counters_collection" in views:
        ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        print("# counters_collection columns:", ccols)
        kn = "kernel_name" if "kernel_name" in ccols else name
        gc = next((x for x in ("grid_size_x", "grid_x", "grid_size") if x in ccols), None)
        if gc:
            for n, g, cn, cnt, avg in c.execute(f"select {kn}, {gc}, counter_name, count(*), avg(value) from counters_collection "
                                                f"where {kn} like ? group by {kn}, {gc}, counter_name order by 1, 2, 3", (f"%{sub}%",)):
                print(f"{n[:40]:40s} grid {g:>9} {cn:28s} {cnt:4d} {avg:18.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
