for r in 1 2; do
for prio in stage_high scan_high equal; do
  export PLSLAM_STREAM_PRIO=$prio
  echo "== $prio"
  bash tools/ab_many.sh 1 v168_48 v144_18 v144_28 v152_26
done
done
