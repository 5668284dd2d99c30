python -m pytest tests/test_gpu_packers.py tests/test_known_answers.py -x -q -m gpu 2>&1 | tail -2
python tools/fuzz_armn.py 60 5 2>&1 | tail -1
python tools/fuzz_armn_wide.py 100 3 2>&1 | tail -1
export EZHIP_LIBRARY=$PWD/devlibs/librmn_ez_hip_dev.so
bash tools/pmc_enc_phases.sh 2>&1 | tail -6
python tools/probe_enc_batch.py 0 32 3 2>/dev/null
unset EZHIP_LIBRARY
bash tools/prof_cmd.sh r5enc3 tools/probe_enc_batch.py 0 > /dev/null 2>&1; head -3 gpurun_out/r5enc3/summary.txt
