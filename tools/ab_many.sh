#!/bin/bash
# A/B of several experiment builds on ONE box over the default bench workload: tools/ab_many.sh <rounds> name1 name2 ...
# (names of build/exp/<name>.so; "tree" = the library of the tree)
rounds=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq $rounds); do
  for v in "$@"; do
    if [ "$v" = tree ]; then unset PLSLAM_HIP_LIB_EXPERIMENT; else export PLSLAM_HIP_LIB_EXPERIMENT=$root/build/exp/$v.so; fi
    python $root/bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 --full-json /tmp/ab_full.json >/dev/null 2>&1
    python -c "
import json; d=json.load(open('/tmp/ab_full.json')); k=d['kernel_ms']
print('%-10s %.0f pairs/s  step %.3f ms (median %.3f)  scan %.3f  post %.3f  scan in step %.3f  post in step %.3f' % ('$v', d['value'], d['ms_per_step'], d['ms_per_step_distribution']['median'], k['scan'], k['post_scan_stages'], k['scan_in_timed_region'], k['post_scan_stages_in_timed_region']))"
  done
done
