cd $GRAFT_REPO_ROOT
echo "== packer tests (scan form default)"; timeout 900 python3 -m pytest tests/test_gpu_packers.py -x -q -m gpu 2>&1 | tail -4
for n in 1e-5 5e-4; do timeout 200 python3 tools/probe_decode1.py $n 2>&1 | grep -v amdgpu.ids | tail -4; done
echo "== armn32"; timeout 200 python3 tools/probe_a32.py 2>&1 | grep -v amdgpu.ids | tail -8
