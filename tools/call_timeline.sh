#!/bin/bash
# The host-API / copy / kernel timeline of the LAST calls of a command (rocprofv3 --hip-trace --kernel-trace --memory-copy-trace; no
# counters): tools/call_timeline.sh <tag> <command...>  -> gpurun_out/<tag>_timeline.txt (everything after the 4th-last
# hipStreamSynchronize: the last three calls of a one-synchronisation-per-call loop)
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd); out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/tl_$tag
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d $out/tl_$tag -o run -- "$@" > $out/${tag}_timeline_log.txt 2>&1
db=$(find $out/tl_$tag -name "*.db" | head -1)
python - "$db" > $out/${tag}_timeline.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
rows = []
def cols(t): return [r[1] for r in c.execute(f"pragma table_info({t})")]
for t in tabs:
    cs = cols(t)
    if t == "kernels" and "start" in cs:
        nm = "name" if "name" in cs else "kernel_name"
        rows += [(s, e, "KERNEL " + n[:70]) for n, s, e in c.execute(f"select {nm}, start, end from {t}")]
    if t == "memory_copies" and "start" in cs:
        nm = "name" if "name" in cs else cs[1]
        rows += [(s, e, f"COPY {n} {sz}") for n, s, e, sz in c.execute(f"select {nm}, start, end, size from {t}")]
    if t in ("regions", "regions_and_samples") and "start" in cs and "name" in cs:
        try:
            rows += [(s, e, "API " + n[:60]) for n, s, e in c.execute(f"select name, start, end from {t}")]
        except sqlite3.Error as ex:
            print("#", t, ex)
rows = sorted(set(rows))
rows = [r for r in rows if not any(k in r[2] for k in ("hipGetDevice", "hipGetLastError", "CallConfiguration"))]
syncs = [i for i, r in enumerate(rows) if "hipStreamSynchronize" in r[2]]
# up to the last synchronisation that ends a call (what follows is the process tearing down)
hi = syncs[-1] + 1 if syncs else len(rows)
lo = syncs[-4] + 1 if len(syncs) >= 4 else 0
t0 = rows[lo][0]
for s, e, w in rows[lo:hi]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {w}")
PY
rm -rf $out/tl_$tag
tail -40 $out/${tag}_timeline.txt
