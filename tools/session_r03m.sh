cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
