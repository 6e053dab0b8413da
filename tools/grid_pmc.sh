R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rm -rf $O/gp_$name; rocprofv3 --pmc "$@" -d $O/gp_$name -o run -- python $R/tools/grid_scaling.py > $O/gp_$name.log 2>&1; python $R/tools/rocpd_by_grid.py $(find $O/gp_$name -name "*.db" | head -1) "k_match_grid<2, 1024" > $O/grid_pmc_$name.txt 2>&1; rm -rf $O/gp_$name; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU
run b SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC
head -60 $O/grid_pmc_a.txt
