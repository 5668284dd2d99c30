cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_packers.py -x -q -m gpu -k "uncompress or armn" 2>&1 | tail -2
python tools/probe_decode_batch.py 1 2>&1 | tail -1
python tools/probe_decode_batch.py 32 2>&1 | tail -1
python tools/fuzz_armn_wide.py 21 40 2>&1 | tail -1
python tools/fuzz_armn32.py 40 22 2>&1 | tail -1
bash tools/prof_cmd.sh r5dec4 tools/probe_decode_batch.py 1 > /dev/null 2>&1
grep "k_dsc_emit\|k_dsc_dtab\|k_armn_dec_spec" gpurun_out/r5dec4/summary.txt | cut -c1-150
