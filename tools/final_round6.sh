#!/bin/bash
# The last evidence set of round 6 (one gpurun call) on the FINAL tree: the whole GPU suite, smoke(), the native LM-iteration timing,
# the forced one-rank-group bench line.  (The default bench line, traces and counters of this tree: tools/profile_round6.sh, TAG r6_w.)
set -x
TAG=${TAG:-r6_z}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd $R
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/${TAG}_tests.txt; cat $O/${TAG}_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > $O/${TAG}_smoke.txt; cat $O/${TAG}_smoke.txt
cp $O/lm_iteration_native.json $O/${TAG}_lm_iteration_native.json
python bench.py --gpus 1 --force-dist --no-cpu-baseline --full-json $O/${TAG}_bench_forced_dist_1rank_full.json > $O/${TAG}_bench_forced_dist_1rank.json 2> $O/${TAG}_bench_forced_dist_1rank.err
python -c "
import json
f=json.loads(open('$O/${TAG}_bench_forced_dist_1rank.json').readline()); print(len(json.dumps(f)), f['value'], f['ms_per_step'], f.get('config4_strong'))"
