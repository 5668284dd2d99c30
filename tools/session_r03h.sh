R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O
cd $R
python3 tools/exp_cfg5_overlap.py 32 0 16 0 > $O/overlap_bb.txt 2>&1
KFILTER=k_sepx bash tools/pmc_cmd.sh r03h_sepx "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" -- tools/perf_probe.py cubic > $O/sepx_pmc.txt 2>&1
grep -v amdgpu $O/overlap_bb.txt; grep "k_sepx<3" $O/sepx_pmc.txt
