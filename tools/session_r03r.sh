cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_packers.py -x -q -m gpu -k "uncompress or decode or unzip or read_side or fst or batch" 2>&1 | tail -3
for v in "EZHIP_DEC_ONE_STREAM=1" "EZHIP_NOP=1" "EZHIP_DEC_ONE_STREAM=1" "EZHIP_NOP=1"; do echo "[$v] $(env $v python3 tools/probe_decode_batch.py 32 5e-4 2>&1 | tail -1)"; done
echo "[16] $(python3 tools/probe_decode_batch.py 16 5e-4 2>&1 | tail -1)"; echo "[16 one stream] $(EZHIP_DEC_ONE_STREAM=1 python3 tools/probe_decode_batch.py 16 5e-4 2>&1 | tail -1)"
