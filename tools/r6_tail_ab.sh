#!/bin/bash
# VERDICT r5 #3's experiment on ONE box: the product library against the experiment build (-DPLSLAM_MI_TAIL=1: K1i's last workgroup
# of a problem merges its column partials; agent-scope release per workgroup, acquire on the last one) with the option off and on
# (`--opt scan_tail=1`: no merge launch).  bench.py verifies every table against the oracle in all three.
root=$(cd "$(dirname "$0")/.." && pwd)
one() {  # tag, extra bench args...
  tag=$1; shift
  python $root/bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 --full-json /tmp/tail_full.json "$@" > /tmp/tail_line.json 2>/tmp/tail_err.txt || { echo "$tag FAILED: $(tail -2 /tmp/tail_err.txt)"; return; }
  python -c "
import json; d=json.load(open('/tmp/tail_full.json')); k=d['kernel_ms']
print('%-22s %.0f pairs/s  step %.3f ms (median %.3f)  scan %.3f  post %.3f  scan in step %.3f  verified: %s' % ('$tag', d['value'], d['ms_per_step'], d['ms_per_step_distribution']['median'], k['scan'], k['post_scan_stages'], k['scan_in_timed_region'], str(d.get('verified'))[:60]))"
}
for r in 1 2 3; do
  unset PLSLAM_HIP_LIB_EXPERIMENT; one product
  export PLSLAM_HIP_LIB_EXPERIMENT=$root/build/exp/${TAILLIB:-tail}.so
  one tail_build_option_off
  one tail_build_scan_tail --opt scan_tail=1
done
