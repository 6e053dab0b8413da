#!/bin/bash
# `mfma_form 6` -- two DIRECTED passes per mutual problem, each keeping only lane-local v_min3_f32 minima of the unpacked
# accumulators (no pack, no row cells, no group pushes, no row finish) -- priced by its UPPER BOUND against K1i on one box:
# the experiment build PLSLAM_MI_X=128 (hamming_mfma_i.hip) IS that loop with every other piece of a real kernel left out
# (no second best, no index: its results are wrong), run over the C2 batch as non-mutual problems = ONE direction; a mutual
# batch costs two such launches.  usage: tools/form6_bound.sh [rounds]   (build first: tools/build_exp.py hamming_mfma_i.hip
# base: form6ub:-DPLSLAM_MI_X=128)
root=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq ${1:-3}); do
  python $root/tools/scan_time.py 4 4096 1 5
  PLSLAM_HIP_LIB_EXPERIMENT=$root/build/exp/base.so python $root/tools/scan_time.py 4 4096 0 5
  PLSLAM_HIP_LIB_EXPERIMENT=$root/build/exp/form6ub.so python $root/tools/scan_time.py 4 4096 0 5
done
