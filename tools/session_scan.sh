cd $GRAFT_REPO_ROOT
for n in 1e-5 5e-4; do timeout 200 python3 tools/probe_decode1.py $n 2>&1 | grep -v amdgpu.ids | tail -3; done
timeout 250 bash tools/prof_list.sh dec_scan "dsc_dtab|dsc_emit|dec_spec" tools/probe_decode1.py 5e-4 | tail -3
EZHIP_DEC_SCAN_MIN_NTX=64 timeout 600 python3 tools/fuzz_armn.py 9407 1000 2>&1 | grep -v amdgpu.ids | tail -2
echo "== batch"; timeout 300 python3 tools/probe_decode_batch.py 2>&1 | grep -v amdgpu.ids | tail -1
