python tools/grid_call_latency.py both 300 0 16 64 1024
python tools/grid_call_latency.py both 300 0 16 64 1024
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for zc in 0 1024; do
  rm -rf $out/hiptl_$zc
  rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d $out/hiptl_$zc -o run -- python $root/tools/grid_call_latency.py lines 30 $zc > $out/r6_k_hiptrace_log_$zc.txt 2>&1
  db=$(find $out/hiptl_$zc -name "*.db" | head -1)
  python - "$db" > $out/r6_k_call_timeline_lines_zc$zc.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print("# tables/views:", [t for t in tabs if not t.startswith('rocpd_info')][:40])
rows = []
def cols(t): return [r[1] for r in c.execute(f"pragma table_info({t})")]
for t in tabs:
    cs = cols(t)
    if t in ("kernels",) and "start" in cs:
        nm = "name" if "name" in cs else "kernel_name"
        rows += [(s, e, "KERNEL " + n[:60]) for n, s, e in c.execute(f"select {nm}, start, end from {t}")]
    if t in ("memory_copies",) and "start" in cs:
        nm = "name" if "name" in cs else cs[1]
        rows += [(s, e, f"COPY {n} {sz}") for n, s, e, sz in c.execute(f"select {nm}, start, end, size from {t}")]
    if t in ("regions", "regions_and_samples", "hip_api", "api") and "start" in cs and "name" in cs:
        try:
            rows += [(s, e, "API " + n[:60]) for n, s, e in c.execute(f"select name, start, end from {t}")]
        except sqlite3.Error as ex:
            print("#", t, ex)
rows = sorted(set(rows))
# the last three calls: everything after the third-last hipStreamSynchronize
syncs = [i for i, r in enumerate(rows) if "hipStreamSynchronize" in r[2]]
lo = syncs[-4] + 1 if len(syncs) >= 4 else 0
t0 = rows[lo][0]
for s, e, w in rows[lo:]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  {w}")
PY
  rm -rf $out/hiptl_$zc
done
tail -45 $out/r6_k_call_timeline_lines_zc0.txt
