# Counters of the lone-problem matchGrid kernels (k_grid_records + k_match_grid<2, 1024>) under the keyframe<->keyframe driver:
# where do the cycles of a 10 us kernel go -- instruction issue, waits, instruction fetch?   gpurun -- bash tools/grid_lone_pmc.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf $O/gl_$name; rocprofv3 --pmc "$@" -d $O/gl_$name -o run -- python $R/tools/driver_trace.py kf2kf_points 1 100 > $O/gl_$name.log 2>&1; python $R/tools/rocpd_by_grid.py $(find $O/gl_$name -name "*.db" | head -1) "grid" > $O/grid_lone_pmc_$name.txt 2>&1; rm -rf $O/gl_$name; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
run b SQ_IFETCH SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU
rocprofv3 --list-avail 2>/dev/null | grep -i "icache\|ifetch\|INST_FETCH" | head -20 > $O/grid_lone_pmc_avail.txt
cat $O/grid_lone_pmc_a.txt | grep -v "^#" | head -40; cat $O/grid_lone_pmc_b.txt | grep -v "^#" | head -40; cat $O/grid_lone_pmc_avail.txt
