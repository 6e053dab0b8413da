for r in 1 2 3; do
  for v in base two four r5; do PLSLAM_HIP_LIB_EXPERIMENT=build/exp/$v.so python tools/scan_time.py 4 4096 1 5 2>/dev/null; done
done
PLSLAM_HIP_LIB_EXPERIMENT=build/exp/prof.so python tools/k1i_profile.py 2>/dev/null
PLSLAM_HIP_LIB_EXPERIMENT=build/exp/prof5.so python tools/k1i_profile.py 2>/dev/null
