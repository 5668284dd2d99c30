python -m pytest tests/test_gpu_packers.py tests/test_gpu_extrema.py tests/test_known_answers.py -x -q -m gpu 2>&1 | tail -3
python tools/fuzz_armn.py 60 5 2>&1 | tail -2
python tools/fuzz_armn_wide.py 100 3 2>&1 | tail -2
