#!/bin/bash
# Round-3 profile set (one gpurun call): bench line, kernel traces (serial / overlapped steps), SQ + FETCH/WRITE counter passes of
# the scan, the PMC entry bench.py reports, the same for the LBA row kernels' streaming launches.  Files -> gpurun_out/${TAG}_*.
set -x
TAG=${TAG:-r3_w}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
kt() {  # tag, command...
  tag=$1; shift
  rm -rf $O/kt_$tag; rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o run -- "$@" > $O/${TAG}_${tag}.stdout 2>/dev/null
  python $R/tools/rocpd_summary.py $(find $O/kt_$tag -name "*.db" | head -1) > $O/${TAG}_kernel_trace_stats_$tag.txt; rm -rf $O/kt_$tag
}
kt serial python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --batches 1
kt overlapped python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary
bash $R/tools/pmc_passes.sh ${TAG} python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-secondary --batches 1
cd $R && python tools/make_pmc_traffic.py ${TAG} 1500 200 4096 k_scan_sym_mfma_h 2.0 > $O/${TAG}_pmc_entry.json 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cd /tmp
kt lba python $R/tools/lba_stream.py
PMC_SQ_ONLY= bash $R/tools/pmc_passes.sh ${TAG}_lba python $R/tools/lba_stream.py
cd $R && python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
python -c "
import json; d=json.loads(open('$O/${TAG}_bench_n1.json').readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['ms_per_step_distribution'])"
head -6 $O/${TAG}_kernel_trace_stats_serial.txt
grep -h "k_point_rows\|k_line_rows" $O/${TAG}_kernel_trace_stats_lba.txt $O/${TAG}_lba_pmc_fetch.txt $O/${TAG}_lba_pmc_write.txt | cut -c1-40,60-140
