#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd database (the *_results.db written by ROCm 7.2's rocprofv3) into
the per-kernel table that `--stats` prints: calls, total / average / min / max duration, and -- if
the run collected PMC counters -- the per-dispatch average of every counter.
   python tools/rocpd_summary.py gpurun_out/prof_kt/bench_results.db > profiles/r1_kernel_trace.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     f"from kernels group by {name_col} order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 kernel summary of {path}")
    print(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for n, cnt, s, a, mn, mx in rows:
        print(f"{n[:70]:70s} {cnt:6d} {s / 1e6:10.3f} {a / 1e3:10.2f} {mn / 1e3:10.2f} {mx / 1e3:10.2f} {100.0 * s / tot:6.2f}")
    try:
        ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    except sqlite3.Error:
        ccols = []
    if ccols:
        kn = "kernel_name" if "kernel_name" in ccols else name_col
        try:
            pm = c.execute(f"select {kn}, counter_name, count(*), avg(value), sum(value) from counters_collection "
                           f"group by {kn}, counter_name order by 1, 2").fetchall()
        except sqlite3.Error as e:
            pm = []
            print("# (no PMC rows:", e, ")")
        if pm:
            print("\n# PMC counters: per-dispatch average (summed over XCDs/SEs by rocprofv3), and dispatch count")
            print(f"{'kernel':60s} {'counter':24s} {'dispatches':>10s} {'avg_per_dispatch':>20s}")
            for n, cn, cnt, avg, _ in pm:
                print(f"{n[:60]:60s} {cn:24s} {cnt:10d} {avg:20.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
