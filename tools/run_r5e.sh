set -x
mkdir -p gpurun_out/r5e
python -m pytest tests/test_gpu_interp.py -x -q -m gpu -k "uvt or cfg3 or wind or uvint or pts2" 2>&1 | tail -15 > gpurun_out/r5e/tests.txt
python tools/sweep_cfg3.py "" "EZHIP_NO_UVP=1" > gpurun_out/r5e/sweep_new.txt 2>&1
export EZHIP_LIBRARY=$PWD/devlibs/librmn_ez_hip_dev.so
python tools/sweep_cfg3.py "" "EZHIP_UVP_PPT=4" "EZHIP_UVT_DEBUG=64" "EZHIP_UVT_DEBUG=64,EZHIP_UVP_PPT=4" "EZHIP_UVT_DEBUG=65" "EZHIP_UVT_DEBUG=66" "EZHIP_UVT_DEBUG=68" "EZHIP_UVT_DEBUG=71" "EZHIP_UVT_DEBUG=79" "EZHIP_UVT_DEBUG=64,EZHIP_UVP_BPC=1" > gpurun_out/r5e/sweep_dbg.txt 2>&1
EZHIP_UVT_CAP=1792 python tools/sweep_cfg3.py "" "EZHIP_UVP_PPT=4" "EZHIP_UVT_DEBUG=64" "EZHIP_UVT_DEBUG=64,EZHIP_UVP_PPT=4" > gpurun_out/r5e/sweep_cap1792.txt 2>&1
EZHIP_UVT_CAP=1280 python tools/sweep_cfg3.py "" "EZHIP_UVP_PPT=4" "EZHIP_UVT_DEBUG=64" "EZHIP_UVT_DEBUG=64,EZHIP_UVP_PPT=4" > gpurun_out/r5e/sweep_cap1280.txt 2>&1
unset EZHIP_LIBRARY
bash tools/prof_cmd.sh r5e/prof tools/probe_cfg3.py > /dev/null 2>&1
cat gpurun_out/r5e/tests.txt gpurun_out/r5e/sweep_*.txt; head -6 gpurun_out/r5e/prof/summary.txt
