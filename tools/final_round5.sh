#!/bin/bash
# The last evidence set of round 5 (one gpurun call) on the FINAL tree: the whole GPU suite, the kernel trace of one LM iteration
# with the Schur step, the forced one-rank-group bench line, the default bench line.  The scan's sources are those of set r5_c
# (profiles/pmc_traffic.json's hash): its traces and counters stand.  Files -> gpurun_out/${TAG}_*.
set -x
TAG=${TAG:-r5_f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/${TAG}_tests.txt; cat $O/${TAG}_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/${TAG}_smoke.txt; cat $O/${TAG}_smoke.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt_schur; rocprofv3 --kernel-trace --stats -d $O/kt_schur -o run -- python $R/tools/lba_iter_trace.py 60 schur > $O/${TAG}_schur.stdout 2>/dev/null
python $R/tools/rocpd_summary.py $(find $O/kt_schur -name "*.db" | head -1) > $O/${TAG}_lba_schur_iteration_trace.txt; rm -rf $O/kt_schur
cat $O/${TAG}_schur.stdout >> $O/${TAG}_lba_schur_iteration_trace.txt; cat $O/${TAG}_lba_schur_iteration_trace.txt
cd $R
python bench.py --gpus 1 --force-dist --no-cpu-baseline > $O/${TAG}_bench_forced_dist_1rank.json 2> $O/${TAG}_bench_forced_dist_1rank.err
python bench.py > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
python -c "
import json
d=json.loads(open('$O/${TAG}_bench_n1.json').readline()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline'].get('frac_profiles'), d['roofline']['traffic'])
l=d['secondary']['lba_plan_iterate_dev']; print({k: l[k] for k in ('us_median','err_only_us_median','state_resident_us_median','state_in_page_locked_images_us_median')}, l['schur_step']['schur_us_median'], l['schur_step']['lm_iteration_blocks_resident_us_median'])
f=json.loads(open('$O/${TAG}_bench_forced_dist_1rank.json').readline()); print(f['value'], f['ms_per_step'], f['secondary'].get('config4_strong', {}).get('vs_plain_step'))"
