#!/bin/bash
# SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS per launch of the scan kernel for several experiment builds
# usage: tools/pmc_valu_exp.sh <form> lib1.so lib2.so ...   -> gpurun_out/pmc_valu_exp.txt
form=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $out/pmc_valu_exp.txt
for lib in "$@"; do
  rm -rf $out/pve
  if [ "$lib" != shipped ]; then export PLSLAM_HIP_LIB_EXPERIMENT=$root/$lib; else unset PLSLAM_HIP_LIB_EXPERIMENT; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY -d $out/pve -o run -- python $root/tools/scan_time.py 4 4096 1 $form > $out/pve.log 2>&1
  db=$(find $out/pve -name "*.db" | head -1)
  echo "== $lib" >> $out/pmc_valu_exp.txt
  python $root/tools/rocpd_summary.py "$db" 2>/dev/null | grep "k_scan" | cut -c1-24,60-140 >> $out/pmc_valu_exp.txt
done
rm -rf $out/pve
cat $out/pmc_valu_exp.txt
