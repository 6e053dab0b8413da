#!/bin/bash
# Round-6 profile set (round 5's script, retagged; the bench line is compact now: the full record is read from its side file) (one gpurun call): kernel traces of the default bench command (serial / overlapped steps), the four SQ +
# FETCH/WRITE counter passes of the scan, the PMC entry bench.py reports (with the trace's average duration: roofline.frac_profiles),
# the driver / LBA-iteration call traces ON THESE SOURCES, the LBA row kernels' streaming launches, then the default bench line.
# Files -> gpurun_out/${TAG}_*; copy what is to be judged into profiles/.
set -x
TAG=${TAG:-r6_p}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
kt() {  # tag, command...
  tag=$1; shift
  rm -rf $O/kt_$tag; rocprofv3 --kernel-trace --stats -d $O/kt_$tag -o run -- "$@" > $O/${TAG}_${tag}.stdout 2>/dev/null
  python $R/tools/rocpd_summary.py $(find $O/kt_$tag -name "*.db" | head -1) > $O/${TAG}_kernel_trace_stats_$tag.txt; rm -rf $O/kt_$tag
}
if [ -z "$ONLY_DRIVERS" ]; then   # (ONLY_DRIVERS=1: the scan's sources are those of the last full set -- its traces and counters stand)
kt serial python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --batches 1
kt overlapped python $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-secondary
bash $R/tools/pmc_passes.sh ${TAG} python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-secondary --batches 1
cd $R && python tools/make_pmc_traffic.py ${TAG} 1500 200 4096 k_scan_sym_mfma_i 2.0 > $O/${TAG}_pmc_entry.json 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
fi
cd /tmp
if [ -z "$SKIP_DRIVERS" ]; then
for w in map2kf_points map2kf_lines kf2kf_points kf2kf_lines; do
  kt driver_$w python $R/tools/driver_trace.py $w 1 60
  mv $O/${TAG}_kernel_trace_stats_driver_$w.txt $O/${TAG}_driver_trace_$w.txt; head -3 $O/${TAG}_driver_$w.stdout >> $O/${TAG}_driver_trace_$w.txt
done
kt driver_map2kf_points_brute_force python $R/tools/driver_trace.py map2kf_points 0 60
mv $O/${TAG}_kernel_trace_stats_driver_map2kf_points_brute_force.txt $O/${TAG}_driver_trace_map2kf_points_brute_force.txt; head -3 $O/${TAG}_driver_map2kf_points_brute_force.stdout >> $O/${TAG}_driver_trace_map2kf_points_brute_force.txt
if [ -z "$ONLY_DRIVERS" ]; then
kt lba_iterate python $R/tools/lba_iter_trace.py 60
mv $O/${TAG}_kernel_trace_stats_lba_iterate.txt $O/${TAG}_lba_iterate_trace.txt; cat $O/${TAG}_lba_iterate.stdout >> $O/${TAG}_lba_iterate_trace.txt
kt lba python $R/tools/lba_stream.py
PMC_SQ_ONLY= bash $R/tools/pmc_passes.sh ${TAG}_lba python $R/tools/lba_stream.py
fi
fi
cd $R && python bench.py --full-json $O/${TAG}_bench_n1_full.json > $O/${TAG}_bench_n1.json 2> $O/${TAG}_bench_n1.err
python -c "
import json; l=json.loads(open('$O/${TAG}_bench_n1.json').readline()); d=json.load(open('$O/${TAG}_bench_n1_full.json'))
print(len(json.dumps(l)), 'bytes;', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline'].get('frac_profiles'), d['roofline']['traffic'], d['ms_per_step_distribution'])"
[ -z "$ONLY_DRIVERS" ] && head -6 $O/${TAG}_kernel_trace_stats_serial.txt
