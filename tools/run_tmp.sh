python -m pytest tests/test_gpu_interp.py -x -q -m gpu 2>&1 | tail -2
bash tools/prof_cmd.sh r5sp tools/probe_cfg3.py > /dev/null 2>&1; grep "cfg3 c_ez" gpurun_out/r5sp/plain.txt; grep "special\|k_uvt<" gpurun_out/r5sp/summary.txt
bash tools/prof_cmd.sh r5sp2 tools/probe_cfg3_scalar.py > /dev/null 2>&1; cat gpurun_out/r5sp2/plain.txt | tail -2; grep "special\|k_st<" gpurun_out/r5sp2/summary.txt
