set -x
mkdir -p gpurun_out/r5b
python -m pytest tests/test_gpu_interp.py -x -q -m gpu -k "uvt or cfg3 or wind or uvint or pts2" 2>&1 | tail -15 > gpurun_out/r5b/tests.txt
python tools/sweep_cfg3.py "" "EZHIP_NO_UVP=1" "EZHIP_UVP_BPC=2" > gpurun_out/r5b/sweep_new.txt 2>&1
EZHIP_LIBRARY=$PWD/devlibs/librmn_ez_hip_dev.so python tools/sweep_cfg3.py "" "EZHIP_UVT_DEBUG=1" "EZHIP_UVT_DEBUG=2" "EZHIP_UVT_DEBUG=4" "EZHIP_UVT_DEBUG=6" "EZHIP_UVT_DEBUG=7" "EZHIP_UVT_DEBUG=15" "EZHIP_UVT_DEBUG=47" "EZHIP_POLAR_WIND_SIDE=1" > gpurun_out/r5b/sweep_dbg.txt 2>&1
bash tools/prof_cmd.sh r5b/prof tools/probe_cfg3.py > /dev/null 2>&1
EZHIP_POLAR_WIND_SIDE=1 bash tools/prof_cmd.sh r5b/prof_side tools/probe_cfg3.py > /dev/null 2>&1
cat gpurun_out/r5b/tests.txt gpurun_out/r5b/sweep_*.txt; head -8 gpurun_out/r5b/prof/summary.txt; head -8 gpurun_out/r5b/prof_side/summary.txt
