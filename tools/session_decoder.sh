# the checks the parallel form of armn_compress UNCOMPRESS (k_dsc_*) went through: gpurun -- 'bash tools/session_decoder.sh'
cd $GRAFT_REPO_ROOT
echo "== packer tests"; timeout 900 python3 -m pytest tests/test_gpu_packers.py -x -q -m gpu 2>&1 | tail -3
echo "== random shapes, the form pushed onto rows of >= 64 tiles (the form and its fallback both occur)"
EZHIP_DEC_SCAN_MIN_NTX=64 timeout 600 python3 tools/fuzz_armn.py 9405 1200 2>&1 | grep -v amdgpu.ids | tail -2
echo "== random shapes, default eligibility"; timeout 600 python3 tools/fuzz_armn.py 9406 600 2>&1 | grep -v amdgpu.ids | tail -2
echo "== one full-size stream at two compression ratios: default, serial chain kernel, state words"
for n in 1e-5 5e-4; do timeout 200 python3 tools/probe_decode1.py $n 2>&1 | grep -v amdgpu.ids | tail -4; done
echo "== a batch of 32"; timeout 300 python3 tools/probe_decode_batch.py 2>&1 | grep -v amdgpu.ids | tail -2
echo "== kernel trace"; timeout 250 bash tools/prof_list.sh dec_scan "armn_dec|dsc_" tools/probe_decode1.py 5e-4 | tail -26
