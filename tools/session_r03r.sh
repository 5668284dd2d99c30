cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_packers.py -x -q -m gpu -k "uncompress or decode or unzip or read_side or fst or batch" 2>&1 | tail -3
timeout 900 python3 tools/fuzz_armn.py 3000 14107 2>&1 | tail -n 1
python3 tools/probe_decode1.py 1e-5 2>&1 | tail -1
python3 tools/probe_decode1.py 1e-4 2>&1 | tail -1
python3 tools/probe_decode1.py 5e-4 2>&1 | tail -1
python3 tools/probe_decode_batch.py 32 5e-4 2>&1 | tail -1
