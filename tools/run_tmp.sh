set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_packers.py -x -q -m gpu -k "ragged_minimum" 2>&1 | tail -30
python tools/fuzz_armn.py 5 80 2>&1 | tail -3
python tools/fuzz_armn_wide.py 7 12 2>&1 | tail -3
