#!/bin/bash
# LBA row kernels' streaming launches under rocprofv3 (kernel trace + FETCH/WRITE passes) and the default bench line.
# Files -> gpurun_out/${TAG}_*   (the scan's passes: tools/profile_round3.sh)
TAG=${TAG:-r3_w}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt_lba; rocprofv3 --kernel-trace --stats -d $O/kt_lba -o run -- python $R/tools/lba_stream.py > $O/${TAG}_lba_stream.json 2>/dev/null
python $R/tools/rocpd_summary.py $(find $O/kt_lba -name "*.db" | head -1) > $O/${TAG}_kernel_trace_stats_lba.txt; rm -rf $O/kt_lba
PMC_SQ_ONLY= bash $R/tools/pmc_passes.sh ${TAG}_lba python $R/tools/lba_stream.py
python $R/tools/lba_stream.py 64 256 > $O/${TAG}_lba_stream_0p4GB.json 2>/dev/null
cd $R && python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
cat $O/${TAG}_lba_stream.json $O/${TAG}_lba_stream_0p4GB.json
grep -h "k_point_rows\|k_line_rows" $O/${TAG}_kernel_trace_stats_lba.txt $O/${TAG}_lba_pmc_fetch.txt $O/${TAG}_lba_pmc_write.txt | cut -c1-40,60-140
python -c "
import json; d=json.loads(open('$O/${TAG}_bench_n1.json').readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['traffic']); c=d['secondary']['c3']; print(c['lba_point_rows_streaming']); print(c['lba_line_rows_streaming'])"
tail -2 $O/${TAG}_bench_n1.err
