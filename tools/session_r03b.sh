R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gpu_extrema.py -x -q -m gpu > $O/pytest_extrema.txt 2>&1
timeout 900 python3 -m pytest tests/test_gpu_packers.py -x -q -m gpu > $O/pytest_packers.txt 2>&1
python3 tools/exp_cfg5_overlap.py 32 0 > $O/bb_on.txt 2>&1
EZHIP_CFG5_NO_BB=1 python3 tools/exp_cfg5_overlap.py 32 0 > $O/bb_off.txt 2>&1
EZHIP_BB_FORCE_ALL=1 python3 tools/exp_cfg5_overlap.py 8 0 > $O/bb_forceall.txt 2>&1
bash tools/prof_cmd.sh r03b_trace tools/probe_cfg5.py 32 > /dev/null 2>&1
tail -4 $O/pytest_extrema.txt $O/pytest_packers.txt; tail -2 $O/bb_on.txt $O/bb_off.txt $O/bb_forceall.txt; head -12 $R/gpurun_out/r03b_trace/summary.txt
