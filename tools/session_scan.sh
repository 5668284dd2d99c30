cd $GRAFT_REPO_ROOT
EZHIP_DEC_SCAN_MIN_NTX=64 timeout 600 python3 tools/fuzz_armn.py 9405 1200 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python3 tools/fuzz_armn.py 9406 600 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python3 -m pytest tests/test_gpu_packers.py -x -q -m gpu 2>&1 | tail -3
for n in 1e-5 5e-4; do timeout 200 python3 tools/probe_decode1.py $n 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "== batch"; timeout 300 python3 tools/probe_decode_batch.py 2>&1 | grep -v amdgpu.ids | tail -2
