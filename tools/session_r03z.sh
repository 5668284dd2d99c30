cd $GRAFT_REPO_ROOT
python3 tools/sweep_lone.py 0 5 0 5 0 2>&1 | grep -v amdgpu.ids
timeout 1500 python3 -m pytest tests/test_gpu_interp.py -x -q -m gpu 2>&1 | tail -2
python3 tools/probe_host_abi3.py 2>&1 | grep -v amdgpu.ids | head -2
