R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O
cd $R
( time timeout 2400 python3 -m pytest tests -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1
python3 tools/exp_cfg5_overlap.py 32 0 > $O/bb_on.txt 2>&1
bash tools/prof_cmd.sh r03c_trace tools/probe_cfg5.py 32 > /dev/null 2>&1
tail -n 8 $O/pytest_gpu.txt; tail -n 1 $O/bb_on.txt; grep "k_bb\|k_sepx\|k_armn" $R/gpurun_out/r03c_trace/summary.txt
