#!/bin/bash
# Round-4 set c (one gpurun call): the GPU test suite, then tools/profile_round4.sh (kernel traces serial / overlapped, four counter
# passes of the scan, pmc_traffic entry, driver / LBA traces, default bench line) with TAG=r4_c, plus: the C3 plan's trace (two
# launches), the FETCH_SIZE calibration for 8-byte gathers (tools/fetch_gather_calib.hip) and one FETCH_SIZE pass of the bench
# command per setting of the finalize kernel's block-table order (option post_xcd = 0 / 1; 2 is the default of the main passes),
# and the default bench command twice with post_xcd 0 and 2.   Files -> gpurun_out/${TAG}_*.
set -x
export TAG=${TAG:-r4_d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R && python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/${TAG}_pytest_gpu.txt; cat $O/${TAG}_pytest_gpu.txt
bash $R/tools/profile_round4.sh
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt_c3; rocprofv3 --kernel-trace --stats -d $O/kt_c3 -o run -- python $R/tools/c3_time.py > $O/${TAG}_c3.stdout 2>/dev/null
python $R/tools/rocpd_summary.py $(find $O/kt_c3 -name "*.db" | head -1) > $O/${TAG}_c3_plan_trace.txt; rm -rf $O/kt_c3; tail -1 $O/${TAG}_c3.stdout >> $O/${TAG}_c3_plan_trace.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_gather_calib.hip -o /tmp/fetch_gather_calib 2>/dev/null
rm -rf $O/pmc_calib; rocprofv3 --pmc FETCH_SIZE -d $O/pmc_calib -o run -- /tmp/fetch_gather_calib > $O/${TAG}_fetch_gather_calib.stdout 2>/dev/null
python $R/tools/rocpd_summary.py $(find $O/pmc_calib -name "*.db" | head -1) > $O/${TAG}_fetch_gather_calib.txt; rm -rf $O/pmc_calib
grep -v "^W2\|^E2\|^I2" $O/${TAG}_fetch_gather_calib.stdout | tail -8 >> $O/${TAG}_fetch_gather_calib.txt; cat $O/${TAG}_fetch_gather_calib.txt
for x in 0 1; do
rm -rf $O/pmc_xcd; rocprofv3 --pmc FETCH_SIZE -d $O/pmc_xcd -o run -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-secondary --batches 1 --opt post_xcd=$x > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find $O/pmc_xcd -name "*.db" | head -1) > $O/${TAG}_pmc_fetch_post_xcd$x.txt; rm -rf $O/pmc_xcd; grep -n "finalize" $O/${TAG}_pmc_fetch_post_xcd$x.txt
done
cd $R
for x in 0 2 0 2; do python bench.py --no-cpu-baseline --no-secondary --steps 30 --opt post_xcd=$x 2>/dev/null | tail -1 | python -c "
import json,sys; b=json.loads(sys.stdin.readline()); print('post_xcd=$x', round(b['value']), round(b['ms_per_step'],4), 'scan', round(b['kernel_ms']['scan'],4), 'stages', round(b['kernel_ms']['post_scan_stages'],4))"
done > $O/${TAG}_post_xcd_ab.txt; cat $O/${TAG}_post_xcd_ab.txt
