#!/bin/bash
# timing experiments on K3 (k_point_rows): builds of build/exp named on the command line (their results are NOT valid rows)
cd $GRAFT_REPO_ROOT
for x in tree "$@"; do
  if [ $x = tree ]; then unset PLSLAM_HIP_LIB_EXPERIMENT; else export PLSLAM_HIP_LIB_EXPERIMENT=$PWD/build/exp/$x.so; fi
  echo "$x: $(python tools/lba_stream.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v['ms_per_launch_events'],4), round(v['GBps_moved'])) for k,v in d.items()})")"
done
