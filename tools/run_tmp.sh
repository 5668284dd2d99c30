cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_packers.py -x -q -m gpu 2>&1 | tail -2
python tools/probe_decode_batch.py 1 2>&1 | tail -1
python tools/probe_decode_batch.py 32 2>&1 | tail -1
python tools/probe_decode_minimum.py 1e-5 7201 3601 2>&1 | grep "SCAN=1\|SCAN=0\|tokens right"
python tools/probe_a32_dev.py 7200 3601 positive 2>&1 | tail -1
python tools/probe_a32_dev.py 7201 3601 mixed 2>&1 | tail -1
python tools/fuzz_armn_wide.py 31 60 2>&1 | tail -1
python tools/fuzz_armn.py 32 60 2>&1 | tail -1
python tools/fuzz_armn32.py 80 33 2>&1 | tail -1
bash tools/prof_cmd.sh r5dec5 tools/probe_decode_batch.py 1 > /dev/null 2>&1
head -14 gpurun_out/r5dec5/summary.txt | cut -c1-150
