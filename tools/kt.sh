#!/bin/bash
# kernel-trace summary of one command: tools/kt.sh <tag> <command...>  -> gpurun_out/<tag>_kt.txt
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf $out/${tag}_kt
rocprofv3 --kernel-trace --stats -d $out/${tag}_kt -o run -- "$@" > $out/${tag}_kt.log 2>&1
python $root/tools/rocpd_summary.py $(find $out/${tag}_kt -name "*.db" | head -1) > $out/${tag}_kt.txt 2>&1
rm -rf $out/${tag}_kt
head -12 $out/${tag}_kt.txt
