#!/bin/bash
# A/B of two builds of the library on ONE box (boxes differ by a few per cent): tools/ab_lib.sh <other.so> [rounds] [bench args...]
# alternates the tree's library with <other.so> (loaded through PLSLAM_HIP_LIB_EXPERIMENT) over the default bench workload.
other=$1; rounds=${2:-3}; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
one() {
  python $root/bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 --full-json /tmp/ab_full.json "$@" >/dev/null 2>&1; python -c "
import json,sys; d=json.load(open('/tmp/ab_full.json')); k=d['kernel_ms']
print('%-6s %.0f pairs/s  step %.3f ms (median %.3f)  scan %.3f  post %.3f  scan in step %.3f' % ('$tag', d['value'], d['ms_per_step'], d['ms_per_step_distribution']['median'], k['scan'], k['post_scan_stages'], k['scan_in_timed_region']))"
}
for r in $(seq $rounds); do
  tag=tree; one "$@"
  tag=other; PLSLAM_HIP_LIB_EXPERIMENT=$other one "$@"
done
