# kernel timelines of the bench's timed loop for two configurations (which kernel runs when, relative to the scans)
root=$(pwd); out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for cfg in "stage_high v168_48" "scan_high v144_18" "scan_high v168_48"; do
  set -- $cfg
  export PLSLAM_STREAM_PRIO=$1 PLSLAM_HIP_LIB_EXPERIMENT=$root/build/exp/$2.so
  rm -rf $out/tl_$1_$2
  rocprofv3 --kernel-trace -d $out/tl_$1_$2 -o run -- python $root/bench.py --no-cpu-baseline --no-secondary --steps 12 --warmup 3 --full-json /tmp/x.json > $out/r6_g_log_$1_$2.txt 2>&1
  db=$(find $out/tl_$1_$2 -name "*.db" | head -1)
  echo "== $1 $2"; python $root/tools/rocpd_timeline.py $db 400 | grep -v "^#" > $out/r6_g_timeline_$1_$2.txt
  rm -rf $out/tl_$1_$2
  grep -c . $out/r6_g_timeline_$1_$2.txt
done
