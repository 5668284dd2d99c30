set -x
mkdir -p gpurun_out/r5a
python -m pytest tests/test_gpu_interp.py -x -q -m gpu -k "uvt or cfg3 or wind or uvint or pts2" 2>&1 | tail -15 > gpurun_out/r5a/tests.txt
EZHIP_LIBRARY=$PWD/devlibs/base.so python tools/sweep_cfg3.py "" > gpurun_out/r5a/sweep_base.txt 2>&1
python tools/sweep_cfg3.py "" "EZHIP_UVT_XCD=1" > gpurun_out/r5a/sweep_new.txt 2>&1
EZHIP_LIBRARY=$PWD/devlibs/base.so python tools/sweep_cfg3.py "" > gpurun_out/r5a/sweep_base2.txt 2>&1
bash tools/prof_cmd.sh r5a/prof tools/probe_cfg3.py > /dev/null 2>&1
cat gpurun_out/r5a/tests.txt gpurun_out/r5a/sweep_*.txt; head -12 gpurun_out/r5a/prof/summary.txt
