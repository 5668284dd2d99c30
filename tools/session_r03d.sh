R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O
cd $R
python3 tools/probe_cfg3.py > $O/cfg3.txt 2>&1
bash tools/prof_cmd.sh r03d_cfg3trace tools/probe_cfg3.py > /dev/null 2>&1
KFILTER=k_pts,k_polar bash tools/pmc_cmd.sh r03d_cfg3 "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" -- tools/probe_cfg3.py > $O/cfg3_pmc.txt 2>&1
python3 tools/exp_cfg5_overlap.py 32 0 > $O/bb_on.txt 2>&1
bash tools/prof_cmd.sh r03d_trace tools/probe_cfg5.py 32 > /dev/null 2>&1
timeout 900 python3 -m pytest tests/test_gpu_extrema.py tests/test_gpu_interp.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -n 1 $O/cfg3.txt; head -4 $R/gpurun_out/r03d_cfg3trace/summary.txt; grep k_pts2 $O/cfg3_pmc.txt; tail -n 1 $O/bb_on.txt; grep "k_bb" $R/gpurun_out/r03d_trace/summary.txt; tail -n 3 $O/pytest.txt
