#!/bin/bash
# timing experiments on k_lba_blocks (builds of build/exp named on the command line; their results are NOT valid)
cd $GRAFT_REPO_ROOT
for x in tree "$@"; do
  if [ $x = tree ]; then unset PLSLAM_HIP_LIB_EXPERIMENT; else export PLSLAM_HIP_LIB_EXPERIMENT=$PWD/build/exp/$x.so; fi
  bash tools/kt.sh $x python $PWD/tools/lba_iter_trace.py 60 schur two > /dev/null 2>&1
  echo "$x: $(grep 'k_lba_blocks' gpurun_out/${x}_kt.txt | awk '{print $(NF-4), $(NF-3)}')"
done
