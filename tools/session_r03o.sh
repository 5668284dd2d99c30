cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_lambert.py -x -q 2>&1 | tail -25
