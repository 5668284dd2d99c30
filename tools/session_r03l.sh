cd $GRAFT_REPO_ROOT
timeout 600 python3 tools/probe_host_abi3.py 2>&1 | grep -v amdgpu.ids
EZHIP_HOST_UPLOADER=1 EZHIP_BOUNCE_MB=4 timeout 600 python3 tools/probe_host_abi3.py 2>&1 | grep -v amdgpu.ids
EZHIP_HOST_UPLOADER=1 timeout 1200 python3 -m pytest tests/test_gpu_interp.py tests/test_gpu_vs_reference_build.py -x -q -m gpu 2>&1 | tail -4
timeout 1200 python3 -m pytest tests/test_gpu_interp.py tests/test_gpu_packers.py -x -q -m gpu 2>&1 | tail -4
