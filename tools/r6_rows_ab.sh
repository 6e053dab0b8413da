#!/bin/bash
# K3 / K4 streaming launches (tools/lba_stream.py), alternating on ONE box: builds of build/exp named on the command line against the
# tree's library; first the tree with the pose count not stated (PLSLAM_STREAM_SLOTS=0: the matrices gathered from global memory)
cd $GRAFT_REPO_ROOT
one() { python tools/lba_stream.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:(round(v['ms_per_launch_events'],4), round(v['GBps_moved'])) for k,v in d.items()})"; }
for r in 1 2; do
  unset PLSLAM_HIP_LIB_EXPERIMENT
  echo "tree, n_pose_slots 0: $(PLSLAM_STREAM_SLOTS=0 one)"
  echo "tree: $(one)"
  for x in "$@"; do echo "$x: $(PLSLAM_HIP_LIB_EXPERIMENT=$PWD/build/exp/$x.so one)"; done
done
