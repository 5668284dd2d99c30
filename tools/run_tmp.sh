python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r05_tests.txt
bash tools/prof_round.sh r05 > gpurun_out/prof_round_r05.log 2>&1
cat gpurun_out/r05_tests.txt; tail -2 gpurun_out/prof_round_r05.log
