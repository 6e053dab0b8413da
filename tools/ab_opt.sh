#!/bin/bash
# A/B of a context option on one box: tools/ab_opt.sh "<bench args A>" "<bench args B>" [rounds]
root=$(cd "$(dirname "$0")/.." && pwd)
a=$1; b=$2; rounds=${3:-3}
one() {
  python $root/bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']
print('%-14s %.0f pairs/s  step %.3f ms (median %.3f)  scan %.3f  post %.3f  scan in step %.3f' % ('$1', d['value'], d['ms_per_step'], d['ms_per_step_distribution']['median'], k['scan'], k['post_scan_stages'], k['scan_in_timed_region']))"
}
for r in $(seq $rounds); do one A "$a"; one B "$b"; done
