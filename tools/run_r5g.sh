mkdir -p gpurun_out/r5g
bash tools/pmc_sq.sh r5g/uvp "k_uvp" tools/probe_cfg3.py > /dev/null 2>&1
EZHIP_NO_UVP=1 bash tools/pmc_sq.sh r5g/uvt "k_uvt" tools/probe_cfg3.py > /dev/null 2>&1
cat gpurun_out/r5g/uvp/sq.txt gpurun_out/r5g/uvt/sq.txt
