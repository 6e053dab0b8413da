R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_match.py tests/test_stereo_gates.py tests/test_gpu_bench_script.py -m gpu -x -q > $O/final_pytest_match.txt 2>&1; grep -E "passed|failed|error" $O/final_pytest_match.txt | tail -2
bash tools/profile_round.sh > $O/profile_round.log 2>&1
python tools/make_pmc_traffic.py r2_p 1500 200 4096 k_scan_sym_mfma_g > /dev/null 2>&1
python bench.py > $O/r2_p_bench_n1.json 2> $O/r2_p_bench_n1.err
cp profiles/pmc_traffic.json $O/pmc_traffic.json
python -c "
import json; d=json.loads(open('$O/r2_p_bench_n1.json').readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['traffic'])"
head -5 $O/r2_p_kernel_trace_stats_serial.txt | tail -3
