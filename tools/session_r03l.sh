cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_packers.py -x -q -m gpu -k "one_launch or cfg5 or fused" 2>&1 | tail -25
