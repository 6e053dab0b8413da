#!/bin/bash
# usage: ab_multi.sh rounds lib1 lib2 ...   ("tree" = the tree's library)
rounds=$1; shift
root=$GRAFT_REPO_ROOT
one() {
  python $root/bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); k=d['kernel_ms']
print('%-8s %.0f pairs/s  step %.3f ms (median %.3f)  scan %.3f  post %.3f  scan in step %.3f' % ('$tag', d['value'], d['ms_per_step'], d['ms_per_step_distribution']['median'], k['scan'], k['post_scan_stages'], k['scan_in_timed_region']))"
}
for r in $(seq $rounds); do
  for lib in "$@"; do
    tag=$lib
    if [ "$lib" = tree ]; then one; else PLSLAM_HIP_LIB_EXPERIMENT=$root/build/exp/$lib.so one; fi
  done
done
