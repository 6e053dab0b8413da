#!/usr/bin/env python3
"""Prints the last N kernel dispatches and memory copies of a rocprofv3 rocpd database in start order (ns relative)."""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
kc = [r[1] for r in c.execute("pragma table_info(kernels)")]
nm = "name" if "name" in kc else "kernel_name"
for name, s, e in c.execute(f"select {nm}, start, end from kernels"):
    rows.append((s, e, "K " + name[:40]))
try:
    mc = [r[1] for r in c.execute("pragma table_info(memory_copies)")]
    ncol = "name" if "name" in mc else mc[1]
    for name, s, e, sz in c.execute(f"select {ncol}, start, end, size from memory_copies"):
        rows.append((s, e, f"C {name} {sz}"))
except sqlite3.Error as ex:
    print("# no memory_copies table:", ex)
rows.sort()
rows = rows[-n:]
t0 = rows[0][0]
for s, e, what in rows:
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f} us  {what}")
