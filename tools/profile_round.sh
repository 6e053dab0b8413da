set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
python $R/bench.py > $O/r2_p_bench_n1.json 2> $O/r2_p_bench_n1.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt_s; rocprofv3 --kernel-trace --stats -d $O/kt_s -o run -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap > $O/r2_p_bench_n1_serial.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find $O/kt_s -name "*.db" | head -1) > $O/r2_p_kernel_trace_stats_serial.txt; rm -rf $O/kt_s
rm -rf $O/kt_o; rocprofv3 --kernel-trace --stats -d $O/kt_o -o run -- python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/kt_o -name "*.db" | head -1) > $O/r2_p_kernel_trace_stats_overlapped.txt; rm -rf $O/kt_o
bash $R/tools/pmc_passes.sh r2_p python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-secondary
head -5 $O/r2_p_kernel_trace_stats_serial.txt
