cd $GRAFT_REPO_ROOT
EZHIP_TRACE_FIRST=1 python tools/probe_cfg2_first.py 2>&1 | tail -18
python tools/probe_cfg3_first.py 2>&1 | tail -3
python -m pytest tests/test_gpu_interp.py -x -q -m gpu -k "wind or uvint or cfg3" 2>&1 | tail -2
