set -x
mkdir -p gpurun_out/r5k
python tools/fuzz_vs_ref2.py 600 7 2>/dev/null | tail -2 > gpurun_out/r5k/fuzz7.txt
python -m pytest tests/test_gpu_interp.py tests/test_gpu_vs_reference_build.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r5k/tests.txt
python tools/sweep_cfg3.py "" > gpurun_out/r5k/sweep_new.txt 2>&1
EZHIP_LIBRARY=$PWD/devlibs/base.so python tools/sweep_cfg3.py "" > gpurun_out/r5k/sweep_base.txt 2>&1
bash tools/prof_cmd.sh r5k/prof tools/probe_cfg3.py > /dev/null 2>&1
cat gpurun_out/r5k/fuzz7.txt gpurun_out/r5k/tests.txt gpurun_out/r5k/sweep_*.txt; head -4 gpurun_out/r5k/prof/summary.txt
