#!/bin/bash
# bench.py's default workload under different context options on one box: tools/opt_sweep.sh "" "--opt direct21=1" "--post-wgs 768" ...
root=$(cd "$(dirname "$0")/.." && pwd)
for a in "$@"; do
  python $root/bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 $a 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']
print('%-28s %.0f pairs/s  step %.3f ms (median %.3f)  scan %.3f  post %.3f  scan in step %.3f post in step %.3f' % ('[$a]', d['value'], d['ms_per_step'], d['ms_per_step_distribution']['median'], k['scan'], k['post_scan_stages'], k['scan_in_timed_region'], k['post_scan_stages_in_timed_region']))"
done
