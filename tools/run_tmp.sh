python tools/probe_cfg3_first.py 2>&1 | tail -1
EZHIP_LOCATE_THREADS=32 python tools/probe_cfg3_first.py 2>&1 | tail -1
EZHIP_LOCATE_THREADS=64 python tools/probe_cfg3_first.py 2>&1 | tail -1
python -m pytest tests/test_gpu_interp.py -x -q -m gpu -k "cfg3" 2>&1 | tail -2
