#!/bin/bash
# the cap on the post-scan stages' workgroups (option post_workgroups) over the default bench workload, on one box
root=$(cd "$(dirname "$0")/.." && pwd)
for n in "$@"; do
  python $root/bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 --post-wgs $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']
print('post_wgs %-6s %.0f pairs/s  step %.3f ms (median %.3f)  scan %.3f  post %.3f  scan in step %.3f post in step %.3f' % ('$n', d['value'], d['ms_per_step'], d['ms_per_step_distribution']['median'], k['scan'], k['post_scan_stages'], k['scan_in_timed_region'], k['post_scan_stages_in_timed_region']))"
done
