cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_interp.py -x -q -m gpu -k "staged" 2>&1 | tail -3
python tools/fuzz_vs_ref2.py 150 701 2>&1 | tail -1 | cut -c1-160
python tools/fuzz_vs_ref.py 100 702 2>&1 | tail -1 | cut -c1-160
