cd $GRAFT_REPO_ROOT
python tools/probe_decode_batch.py 32 2>&1 | tail -1
python tools/probe_decode_batch.py 1 2>&1 | tail -1
python -m pytest tests/test_gpu_packers.py -x -q -m gpu -k "uncompress or armn" 2>&1 | tail -2
python tools/fuzz_armn.py 11 60 2>&1 | tail -1
python tools/fuzz_armn32.py 5 2>&1 | tail -1
bash tools/prof_cmd.sh r5dec3 tools/probe_decode_batch.py > /dev/null 2>&1
head -6 gpurun_out/r5dec3/summary.txt | cut -c1-140
