cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_interp.py -x -q -m gpu -k "uvint or cfg3 or wind or batch_equals" 2>&1 | tail -3
python tools/probe_cfg3_pair_batch.py 2 8 2>&1 | grep -v amdgpu
EZHIP_UVT_SPECIAL_LAUNCH=1 python tools/probe_cfg3_pair_batch.py 2 8 2>&1 | grep -v amdgpu
python tools/sweep_cfg3.py '' EZHIP_UVT_SPECIAL_LAUNCH=1 2>&1 | grep -v amdgpu
python tools/fuzz_vs_ref2.py 120 601 2>&1 | tail -1
