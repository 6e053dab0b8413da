#!/bin/bash
# timing experiments on the Schur partials' launch: builds of build/exp named on the command line (results may NOT be valid numbers)
cd $GRAFT_REPO_ROOT
for x in tree "$@"; do
  if [ $x = tree ]; then unset PLSLAM_HIP_LIB_EXPERIMENT; else export PLSLAM_HIP_LIB_EXPERIMENT=$PWD/build/exp/$x.so; fi
  bash tools/kt.sh $x python $PWD/tools/lba_iter_trace.py 60 schur two > /dev/null 2>&1
  echo "$x: $(grep k_schur_partials_lba gpurun_out/${x}_kt.txt | awk '{print $(NF-4), $(NF-3)}') $(grep 'us per iteration' gpurun_out/${x}_kt.log | sed 's/.*: //')"
done
