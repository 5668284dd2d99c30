cd $GRAFT_REPO_ROOT
bash tools/pmc_cmd.sh r03k_pmc "k_sepx|k_armn_enc1|k_bb_bounds" tools/probe_cfg5_fused.py 8
timeout 900 python3 -m pytest tests/test_gpu_packers.py -x -q -m gpu -k "cfg5 or fused" 2>&1 | tail -3
EZHIP_CFG5_FUSED=1 timeout 900 python3 -m pytest tests/test_gpu_packers.py -x -q -m gpu -k "cfg5 or fused" 2>&1 | tail -3
