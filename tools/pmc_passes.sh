#!/bin/bash
# Collects the SQ / TCC counter passes of one command (rocprofv3 --pmc, one pass per counter set, no tracing options)
# and summarises each rocpd database with tools/rocpd_summary.py.
# usage: tools/pmc_passes.sh <tag> <command...>        -> gpurun_out/<tag>_pmc_{a,b,c,fetch,write}.txt
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name counters...
  name=$1; shift
  rm -rf $out/${tag}_pmc_$name
  rocprofv3 --pmc "$@" -d $out/${tag}_pmc_$name -o run -- "${CMD[@]}" > $out/${tag}_pmc_$name.log 2>&1
  db=$(find $out/${tag}_pmc_$name -name "*.db" | head -1)
  python $root/tools/rocpd_summary.py "$db" > $out/${tag}_pmc_$name.txt 2>&1
  rm -rf $out/${tag}_pmc_$name
}
CMD=("$@")
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
run b SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU
if [ -z "$PMC_SQ_ONLY" ]; then
run fetch FETCH_SIZE
run write WRITE_SIZE GRBM_GUI_ACTIVE
fi
