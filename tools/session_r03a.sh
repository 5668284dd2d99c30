# round 3, GPU session a: overlap experiment, encoder phases, cfg3 PMC
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
python3 tools/exp_cfg5_overlap.py 32 0 16 8 4 > $O/overlap.txt 2>&1
python3 tools/probe_enc.py > $O/enc.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_packers.py -x -q -m gpu > $O/pytest_packers.txt 2>&1
EZHIP_CFG5_SUB=8 bash tools/prof_cmd.sh r03a_trace8 tools/probe_cfg5.py 32 > /dev/null 2>&1
KFILTER=k_pts,k_polar,k_wind bash tools/pmc_cmd.sh r03a_cfg3 "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" -- tools/probe_cfg3.py > $O/cfg3_pmc.txt 2>&1
bash tools/prof_cmd.sh r03a_cfg3trace tools/probe_cfg3.py > /dev/null 2>&1
tail -5 $O/overlap.txt $O/enc.txt $O/pytest_packers.txt
