set -x
./build/pk_min3_f16_check > gpurun_out/r6_c_pk_min3_opsel_check.txt 2>&1
tail -8 gpurun_out/r6_c_pk_min3_opsel_check.txt
timeout 900 python -m pytest tests/test_gpu_match.py -x -q 2>&1 | tail -5 > gpurun_out/r6_c_match_tests.txt; cat gpurun_out/r6_c_match_tests.txt
for r in 1 2 3; do
  python tools/scan_time.py 4 4096 1 5 2>/dev/null
  PLSLAM_HIP_LIB_EXPERIMENT=build/exp/r5.so python tools/scan_time.py 4 4096 1 5 2>/dev/null
  python tools/scan_time.py 4 4096 0 5 2>/dev/null
  PLSLAM_HIP_LIB_EXPERIMENT=build/exp/r5.so python tools/scan_time.py 4 4096 0 5 2>/dev/null
done > gpurun_out/r6_c_scan_ab_nopack.txt 2>&1
cat gpurun_out/r6_c_scan_ab_nopack.txt
bash tools/ab_lib.sh build/exp/r5.so 2 > gpurun_out/r6_c_bench_ab_nopack.txt 2>&1; cat gpurun_out/r6_c_bench_ab_nopack.txt
