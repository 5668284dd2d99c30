cd $GRAFT_REPO_ROOT
for v in "EZHIP_NO_SPEC_CACHE=1" "EZHIP_NOP=1" "EZHIP_NO_SPEC_CACHE=1" "EZHIP_NOP=1"; do echo "[$v] $(env $v python3 tools/probe_cfg3.py 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"; done
timeout 1500 python3 -m pytest tests/test_gpu_interp.py tests/test_gpu_vs_reference_build.py tests/test_lambert.py -x -q -m gpu 2>&1 | tail -3
timeout 600 python3 tools/fuzz_vs_ref2.py 1500 12103 2>&1 | tail -n 1
timeout 600 python3 tools/fuzz_vs_ref4.py 400 12105 2>&1 | tail -n 1
